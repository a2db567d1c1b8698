# Timeline of ONE build + probe step (tools/step_timeline.sh LOG2N [VARIANT]): rocprofv3 --kernel-trace of
# `python3 tools/time_build.py`, then every dispatch of the last step with its start offset, duration and the gap to
# the one before -- what a small relation's step is made of (launch tail). Output: gpurun_out/timeline_LOG2N.txt
set -e
L=$1; V=${2:-0}
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/timeline_$L
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $P/trace -- python3 $R/tools/time_build.py --log2n $L --variant $V --reps 5 > $P/cmd.out 2> $P/cmd.err
cd $R
python3 tools/step_timeline.py $P/trace > $R/gpurun_out/timeline_$L.txt
rm -rf $P/trace
cat $R/gpurun_out/timeline_$L.txt
