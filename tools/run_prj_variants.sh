#!/bin/bash
# Development tool (GPU box): tools/time_prj.py once per tools/variants/*.so named ("product" = the product library).
# Variants are loaded by path (HJ_DEV_LIB_VARIANT, htm-hashjoin_amd/_lib.py); the product library is never touched.
names=$1; shift
for v in $names; do
  if [ "$v" == "product" ]; then unset HJ_DEV_LIB_VARIANT; else export HJ_DEV_LIB_VARIANT=tools/variants/$v.so; fi
  timeout -k 10 300 python tools/time_prj.py --tag $v "$@" 2>/dev/null || echo "{\"tag\": \"$v\", \"error\": $?}"
done
