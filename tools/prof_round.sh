# Round profile (tools/prof_round.sh rNN): kernel-trace stats of the bench's main leg (`bench.py --no-extra
# --no-cpu-baseline`: only the metric's workload, so that the per-kernel averages can be held against the HIP-event
# times of the JSON line) + separate FETCH_SIZE / WRITE_SIZE passes of the same leg (never combined with tracing).
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_$TAG
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python3 $R/bench.py --no-extra --no-cpu-baseline > $P/bench_under_rocprof.json 2> $P/bench_under_rocprof.err
echo "stats done" ; tail -c 300 $P/bench_under_rocprof.json
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $P/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline > $P/write.log 2>&1
echo "write done"
cd $R
python3 tools/summarize_prof.py stats $P/stats $P/kernel_stats.csv
python3 tools/summarize_prof.py pmc $P/fetch $P/pmc_fetch.json
python3 tools/summarize_prof.py pmc $P/write $P/pmc_write.json
python3 tools/summarize_prof.py traffic $P/pmc_fetch.json $P/pmc_write.json $P/pmc_traffic.json
rm -rf $P/stats $P/fetch $P/write
cat $P/kernel_stats.csv; cat $P/pmc_traffic.json
