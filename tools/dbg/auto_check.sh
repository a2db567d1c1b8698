rm -f gpurun_out/auto.log
for e in 9 10 11 12 13; do
  W=$((1<<e))
  echo "W=2^$e auto" >> gpurun_out/auto.log
  timeout -k 10 120 python bench.py --log2n 27 --steps 3 --warmup 1 --no-extra --no-cpu-baseline --dist local_shuffle --shuffle-range $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['result']['buildVariant'], d['result']['buildDeferred'], round(d['ms_per_step'],3))" >> gpurun_out/auto.log 2>&1 || exit 1
done
cat gpurun_out/auto.log
