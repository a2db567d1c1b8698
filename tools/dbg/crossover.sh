mkdir -p gpurun_out; rm -f gpurun_out/cross.log
for e in 10 11 12 13 14 16; do
  W=$((1<<e))
  for v in 2 1; do
    echo "W=2^$e variant=$v" >> gpurun_out/cross.log
    timeout -k 10 120 python bench.py --log2n 27 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --build-variant $v --dist local_shuffle --shuffle-range $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(r['launch_us']), r['other_kernels'], d['result']['buildDeferred'], round(d['ms_per_step'],3))" >> gpurun_out/cross.log 2>&1 || exit 1
  done
done
cat gpurun_out/cross.log
