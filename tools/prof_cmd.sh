# Kernel-trace stats of any command (tools/prof_cmd.sh TAG program args...): per-kernel summary to
# gpurun_out/prof_TAG/kernel_stats.csv. The program itself follows `--` (no shell, env or launcher in between).
set -e
TAG=$1; shift
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/prof_$TAG
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- "$@" > $P/cmd.out 2> $P/cmd.err
cd $R
python3 tools/summarize_prof.py stats $P/stats $P/kernel_stats.csv
rm -rf $P/stats
cut -c1-160 $P/kernel_stats.csv
