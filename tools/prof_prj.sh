# PRJ per-kernel times at 2^30 (config 3): rocprofv3 kernel stats of the reference CLI on the library.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_prj
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $R/htm-hashjoin_amd/bin/main --algo prj \
    --rSize ${PRJ_RSIZE:-1073741824} --dataDistr local_shuffle --shuffleRange 1024 --repeat 4 > $OUT/main.log 2> $OUT/main.err
cd $R
python3 tools/summarize_prof.py stats $OUT/stats $OUT/kernel_stats.csv
rm -rf $OUT/stats
cat $OUT/kernel_stats.csv; tail -2 $OUT/main.log
