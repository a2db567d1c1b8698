#!/bin/bash
# Development tool (GPU box): times tools/variants/*.so one after the other through tools/time_build.py.
#   tools/run_variants.sh "base abl1" --log2n 27 --variant 3 --dists uniform:16,sorted:16
names=$1; shift
lib=htm-hashjoin_amd/lib/libhtmjoin_hip.so
cp $lib /tmp/libhtmjoin_hip.product.so
for v in $names; do
  cp tools/variants/$v.so $lib
  timeout -k 10 300 python tools/time_build.py --tag $v "$@" || echo "{\"tag\": \"$v\", \"error\": $?}"
done
cp /tmp/libhtmjoin_hip.product.so $lib
