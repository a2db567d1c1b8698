rm -f gpurun_out/look.log
for d in 1 0; do
  cp tools/dbg/lib_look_$d.so htm-hashjoin_amd/lib/libhtmjoin_hip.so
  for cfg in "uniform 16" "local_shuffle 1024"; do
    set -- $cfg
    echo "look=$d dist=$1" >> gpurun_out/look.log
    timeout -k 10 120 python bench.py --log2n 27 --steps 5 --warmup 1 --no-extra --no-cpu-baseline --build-variant 2 --dist $1 --shuffle-range $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['roofline']['launch_us']), d['result']['conflicts'], d['ms_per_step'])" >> gpurun_out/look.log 2>&1 || exit 1
  done
done
cat gpurun_out/look.log
