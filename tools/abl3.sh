mkdir -p gpurun_out; rm -f gpurun_out/abl3.log
for a in ${ABLS:-0 1 5 7}; do
  echo "sorted ABLATE=$a" >> gpurun_out/abl3.log
  HJ_OWN_ABLATE=$a timeout -k 10 120 python bench.py --log2n 27 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 --dist sorted 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['roofline']['launch_us']))" >> gpurun_out/abl3.log 2>&1 || exit 1
done
cat gpurun_out/abl3.log
