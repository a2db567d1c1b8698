set -e
mkdir -p gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/uniform -- python3 $R/bench.py --log2n 27 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 --dist uniform > $R/gpurun_out/prof/uniform.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof/lshuf -- python3 $R/bench.py --log2n 27 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 --dist local_shuffle --shuffle-range 1024 > $R/gpurun_out/prof/lshuf.log 2>&1
cd $R/gpurun_out/prof
find . -name "*kernel_stats.csv" | while read f; do echo "== $f"; cut -d, -f1-8 "$f" | head -12; done
