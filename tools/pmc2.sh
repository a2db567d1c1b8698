set -e
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for d in "uniform 16" "local_shuffle 1024"; do
  set -- $d
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"; do
    tag=$(echo $grp | cut -d' ' -f1)
    rm -rf $R/gpurun_out/pmc/$1_$tag
    rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc/$1_$tag -- python3 $R/bench.py --log2n 27 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 --dist $1 --shuffle-range $2 > $R/gpurun_out/pmc/$1_$tag.log 2>&1
  done
done
cd $R/gpurun_out/pmc
python3 - <<'PY'
import csv,glob,collections
for f in sorted(glob.glob('*/*/*counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:30]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[(k,r['Counter_Name'])]+=1
    print(f)
    for k in acc:
        if 'build_own' in k:
            print('  ',k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
cd $R
ABLS="1" bash tools/abl2.sh
