# destination split (hj_shard_histogram_dev + hj_shard_scatter_dev) kernel times at 2^28 tuples for G = 2, 4, 8
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_split
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in ${GS:-2 8}; do
cat > /tmp/split_$G.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np, htm_hashjoin_amd as hj
n = 1 << ${LOG2N:-28}
R_ = hj.generate_data("${DIST:-uniform}", n, n, 16)
with hj.HashJoinContext(0) as c:
    d_in = c.dev_alloc(n * 8); d_out = c.dev_alloc(n * 4 + 64); d_cnt = c.dev_alloc($G * 8)
    c.copy_h2d(d_in, R_)
    for _ in range(4):
        c.shard_histogram(d_in, n, $G, d_cnt)
        c.shard_scatter(d_in, n, $G, d_cnt, d_out)
    cnt = np.empty($G, dtype=np.uint64); c.copy_d2h(cnt, d_cnt); print($G, cnt.tolist())
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/g$G -- python3 /tmp/split_$G.py > $OUT/g$G.log 2>&1
cd $R; python3 tools/summarize_prof.py stats $OUT/g$G $OUT/g$G.csv; rm -rf $OUT/g$G; echo "G=$G"; head -4 $OUT/g$G.csv; cd /tmp
done
