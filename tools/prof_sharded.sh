# kernel stats of the N>1 code path rehearsed on one GPU (world = 1 under RCCL): HJ_BENCH_FORCE_SHARDED=1
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_sharded
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export HJ_BENCH_FORCE_SHARDED=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --log2n ${LOG2N:-28} --steps 5 --warmup 1 > $OUT/bench.json 2> $OUT/bench.err
cd $R
python3 tools/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.csv
rm -rf $OUT/stats
head -20 $OUT/kernel_stats.csv; cut -c1-300 $OUT/bench.json
