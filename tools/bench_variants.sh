#!/bin/bash
# Development tool (GPU box): the bench's main leg (whole step, 20 timed steps) once per tools/variants/*.so named.
names=$1; shift
lib=htm-hashjoin_amd/lib/libhtmjoin_hip.so
cp $lib /tmp/libhtmjoin_hip.product.so
for v in $names; do
  cp tools/variants/$v.so $lib
  timeout -k 10 400 python bench.py --no-extra --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e3,1), 'Gt/s  build', round(d['roofline']['launch_us']), 'probe', round(d['roofline']['other_kernels']['k_probe_us']))"
done
cp /tmp/libhtmjoin_hip.product.so $lib
