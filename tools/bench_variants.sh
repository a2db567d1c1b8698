#!/bin/bash
# Development tool (GPU box): the bench's main leg (whole step, 20 timed steps) once per tools/variants/*.so named
# ("product" = the product library). Variants are loaded by path (HJ_DEV_LIB_VARIANT); the product library is never touched.
names=$1; shift
for v in $names; do
  if [ "$v" == "product" ]; then unset HJ_DEV_LIB_VARIANT; else export HJ_DEV_LIB_VARIANT=tools/variants/$v.so; fi
  timeout -k 10 400 python bench.py --no-extra --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3), 'ms/step', round(d['value']/1e3,1), 'Gt/s  build', round(d['roofline']['launch_us']), 'probe', round(d['roofline']['other_kernels']['k_probe_us']))"
done
