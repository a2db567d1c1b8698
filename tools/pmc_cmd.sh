# One PMC pass of any command (tools/pmc_cmd.sh TAG "COUNTER [COUNTER...]" program args...): per-kernel averages to
# gpurun_out/pmc_TAG/pmc.json (tools/summarize_prof.py pmc). Counters only: never together with tracing.
set -e
TAG=$1; CTR=$2; shift; shift
R=$GRAFT_REPO_ROOT
P=$R/gpurun_out/pmc_$TAG
mkdir -p $P
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc $CTR --output-format csv -d $P/raw -- "$@" > $P/cmd.out 2> $P/cmd.err
cd $R
python3 tools/summarize_prof.py pmc $P/raw $P/pmc.json
rm -rf $P/raw
python3 -c "
import json,sys
d=json.load(open('$P/pmc.json'))
for k,v in d.items(): print(k[:70], v)
"
