mkdir -p gpurun_out
for a in 0 1 2 3 7; do
  echo "ABLATE=$a" >> gpurun_out/abl.log
  HJ_OWN_ABLATE=$a timeout -k 10 120 python bench.py --log2n 27 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['roofline']['kernel_us'], d['result']['buildDeferred'])" >> gpurun_out/abl.log 2>&1 || exit 1
done
cat gpurun_out/abl.log
