#!/bin/bash
# tools/build_variant.sh <name> <source.hip> [extra hipcc flags...] -- a second build of libfastga_amd.so with ONE kernel
# file compiled differently (kernel A/B experiments on the GPU box: FGA_LIBRARY=fastga_amd/variants/lib_<name>.so).
# The variant travels with gpurun (fastga_amd/variants/ is git-ignored through *.so, not gpurun-ignored).
set -e
name=$1; src=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/build/obj
mkdir -p $root/fastga_amd/variants $obj
make -s -C $root/fastga_amd/csrc -j8 >/dev/null
base=$(basename $src .hip)
# the per-file flags of fastga_amd/csrc/Makefile
extra=""; [ $base = fga_extend ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp -fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -Wno-unused-value \
  -I$root/include -I$root/fastga_amd/csrc $extra "$@" -c $root/fastga_amd/csrc/$src -o $obj/variant_$name.o
objs=$(ls $obj/*.o | grep -v "/variant_" | grep -v "/$base.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/fastga_amd/variants/lib_$name.so $objs $obj/variant_$name.o -lz -lpthread
echo built fastga_amd/variants/lib_$name.so
