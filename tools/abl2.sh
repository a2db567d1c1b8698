mkdir -p gpurun_out
rm -f gpurun_out/abl2.log
for cfg in "sorted 16" "local_shuffle 16" "local_shuffle 1024" "uniform 16"; do
  set -- $cfg
  for a in ${ABLS:-0}; do
  echo "dist=$1 W=$2 ABLATE=$a" >> gpurun_out/abl2.log
  HJ_OWN_ABLATE=$a timeout -k 10 120 python bench.py --log2n ${LOG2N:-27} --steps 5 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 --dist $1 --shuffle-range $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print({'phaseA': round(d['roofline']['launch_us']), **{k: (round(v) if isinstance(v,(int,float)) else v) for k,v in d['roofline']['other_kernels'].items() if k.endswith('_us') or 'group' in k}}, d['result']['buildDeferred'], d['ms_per_step'], d['value'])" >> gpurun_out/abl2.log 2>&1 || exit 1
  done
done
cat gpurun_out/abl2.log
