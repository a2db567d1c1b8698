# SQ counters of the build kernels (k_build_wave / k_build_own), 2^27: tools/pmc_build.sh <variant> "<dist:W> ..." [log2n]
# Two --pmc passes (8 SQ slots each), nothing else traced. Summary -> gpurun_out/pmc_build_v<variant>/summary.txt
R=$GRAFT_REPO_ROOT
V=$1; DISTS=${2:-"uniform:16 sorted:16"}; L=${3:-27}
OUT=$R/gpurun_out/pmc_build_v$V
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in $DISTS; do
  i=0
  for grp in "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rm -rf $OUT/${d}_g$i
    timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/${d}_g$i -- python3 $R/tools/time_build.py --log2n $L --variant $V --dists $d --reps 2 > $OUT/${d}_g$i.log 2>&1 || { tail -5 $OUT/${d}_g$i.log; }
  done
done
cd $OUT
python3 - <<'PY' | tee summary.txt
import csv,glob,collections
for f in sorted(glob.glob('*_g*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[(k,r['Counter_Name'])]+=1
    print(f.split('/')[0])
    for k in acc:
        if 'k_build_' in k:
            print('  ',k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
rm -rf $OUT/*_g1 $OUT/*_g2
