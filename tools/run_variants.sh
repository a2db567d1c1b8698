#!/bin/bash
# Development tool (GPU box): times tools/variants/*.so one after the other through tools/time_build.py.
#   tools/run_variants.sh "base abl1" --log2n 27 --variant 3 --dists uniform:16,sorted:16
# The variant is loaded by path (HJ_DEV_LIB_VARIANT, htm-hashjoin_amd/_lib.py); the product library is never touched.
# "product" names the product library itself.
names=$1; shift
for v in $names; do
  if [ "$v" == "product" ]; then unset HJ_DEV_LIB_VARIANT; else export HJ_DEV_LIB_VARIANT=tools/variants/$v.so; fi
  timeout -k 10 300 python tools/time_build.py --tag $v "$@" 2>/dev/null || echo "{\"tag\": \"$v\", \"error\": $?}"
done
