#!/bin/bash
# Development tool (GPU box): tools/time_prj.py once per tools/variants/*.so named.
names=$1; shift
lib=htm-hashjoin_amd/lib/libhtmjoin_hip.so
cp $lib /tmp/libhtmjoin_hip.product.so
for v in $names; do
  cp tools/variants/$v.so $lib
  timeout -k 10 300 python tools/time_prj.py --tag $v "$@" || echo "{\"tag\": \"$v\", \"error\": $?}"
done
cp /tmp/libhtmjoin_hip.product.so $lib
