# Round profile: kernel-trace stats of the bench's main leg (`bench.py --no-extra --no-cpu-baseline`: only the metric's
# workload, so that the per-kernel averages can be held against the HIP-event times of the JSON line) + FETCH_SIZE /
# WRITE_SIZE passes of the same leg.
set -e
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_r01
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r01/stats -- python3 $R/bench.py --no-extra --no-cpu-baseline > $R/gpurun_out/prof_r01/bench_under_rocprof.json 2> $R/gpurun_out/prof_r01/bench_under_rocprof.err
echo "stats done" ; tail -c 600 $R/gpurun_out/prof_r01/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_r01/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $R/gpurun_out/prof_r01/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_r01/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $R/gpurun_out/prof_r01/write.log 2>&1
echo "write done"
cd $R
python3 tools/summarize_prof.py stats gpurun_out/prof_r01/stats gpurun_out/prof_r01/kernel_stats.csv
python3 tools/summarize_prof.py pmc gpurun_out/prof_r01/fetch gpurun_out/prof_r01/pmc_fetch.json
python3 tools/summarize_prof.py pmc gpurun_out/prof_r01/write gpurun_out/prof_r01/pmc_write.json
rm -rf gpurun_out/prof_r01/stats gpurun_out/prof_r01/fetch gpurun_out/prof_r01/write
cat gpurun_out/prof_r01/kernel_stats.csv
