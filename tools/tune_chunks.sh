mkdir -p gpurun_out; rm -f gpurun_out/tune.log
for c in 512 768 1024 2048 4096; do
 for cfg in "local_shuffle 1024" "uniform 16"; do
  set -- $cfg
  echo "chunks=$c dist=$1" >> gpurun_out/tune.log
  HJ_OWN_CHUNKS=$c timeout -k 10 120 python bench.py --log2n ${LOG2N:-27} --steps 5 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 --dist $1 --shuffle-range $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['roofline']['launch_us']), d['result']['buildDeferred'], d['ms_per_step'], d['value'])" >> gpurun_out/tune.log 2>&1 || exit 1
 done
done
cat gpurun_out/tune.log
