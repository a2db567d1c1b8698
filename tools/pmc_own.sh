# SQ counters of k_build_own (LDS side), 2^27, uniform and local_shuffle
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_own
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in "uniform 16" "local_shuffle 1024"; do
  set -- $d
  i=0
  for grp in "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rm -rf $OUT/$1_g$i
    rocprofv3 --pmc $grp --output-format csv -d $OUT/$1_g$i -- python3 $R/bench.py --log2n 27 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --dist $1 --shuffle-range $2 > $OUT/$1_g$i.log 2>&1 || { tail -5 $OUT/$1_g$i.log; }
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
        if 'build_own' in k or 'probe' in k:
            print('  ',k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
rm -rf $OUT/*_g1 $OUT/*_g2
