# SQ / TCC counters of the PRJ kernels (one rocprofv3 --pmc pass per group), 2^28 tuples per relation.
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_prj
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf $OUT/g$i
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- $R/htm-hashjoin_amd/bin/main --algo prj --rSize ${PRJ_RSIZE:-268435456} \
      --dataDistr local_shuffle --shuffleRange 1024 --repeat 2 > $OUT/g$i.log 2>&1
done
cd $OUT
python3 - <<'PY' | tee summary.txt
import csv,glob,collections
for f in sorted(glob.glob('g*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:60]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[(k,r['Counter_Name'])]+=1
    print(f.split('/')[0])
    for k in acc:
        if 'scatter' in k or 'hist' in k or 'join' in k:
            print('  ',k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
rm -rf $OUT/g1 $OUT/g2 $OUT/g3 $OUT/g4
