#!/bin/bash
# Development tool: builds a copy of libhtmjoin_hip.so with extra -D flags on ONE kernel file into tools/variants/NAME.so
#   tools/mk_variant.sh NAME hj_build_wave "-DHJ_WV_PER=4"
# The product build (make -C htm-hashjoin_amd/csrc) never defines any of these.
set -e
name=$1; file=$2; flags=$3
here=$(cd "$(dirname "$0")/.." && pwd)
src=$here/htm-hashjoin_amd/csrc
out=$here/tools/variants
mkdir -p $out/obj_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-value $flags -c $src/$file.hip -o $out/obj_$name/$file.o
objs=""
for o in hj_kernels hj_build_own hj_build_wave hj_htm hj_prj hj_api hj_datagen; do
  if [ "$o" == "$file" ]; then objs="$objs $out/obj_$name/$file.o"; else objs="$objs $src/$o.o"; fi
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/$name.so $objs -lpthread
rm -rf $out/obj_$name
echo built $out/$name.so
