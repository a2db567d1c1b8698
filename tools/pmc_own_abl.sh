# LDS cycles of k_build_own by ablation (HJ_OWN_ABLATE: 1 no insert, 9 +no need/claim, 17 no insert+no retire, 25 none), sorted 2^27
set -e
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_own_abl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for a in ${ABLS:-0 1 9 17 25}; do
  rm -rf $OUT/a$a
  HJ_OWN_ABLATE=$a rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $OUT/a$a -- python3 $R/bench.py --log2n 27 --steps 2 --warmup 1 --no-extra --no-cpu-baseline --dist ${DIST:-local_shuffle} --shuffle-range ${WIN:-1024} --build-variant 2 > $OUT/a$a.log 2>&1 || tail -3 $OUT/a$a.log
done
cd $OUT
python3 - <<'PY' | tee summary.txt
import csv,glob,collections
for f in sorted(glob.glob('a*/**/*counter_collection.csv', recursive=True)):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][:40]
        acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
        cnt[(k,r['Counter_Name'])]+=1
    for k in acc:
        if 'build_own' in k:
            print(f.split('/')[0], k, {c: round(v/cnt[(k,c)]) for c,v in acc[k].items()})
PY
rm -rf $OUT/a[0-9]*/
