#!/bin/bash
# Build a kernel-variant library for A/B timing:  tools/build_variant.sh NAME file.hip "-DFLAG=1 ..."
# -> otvm_amd/variants/libotvm_NAME.so (use with OTVM_HIP_LIB=...; the directory travels to the GPU box, csrc/build/ does not)
set -e
cd "$(dirname "$0")/.."
python otvm_amd/csrc/build.py > /dev/null
name=$1; src=$2; flags=$3
d=otvm_amd/variants; mkdir -p $d
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $flags -c otvm_amd/csrc/$src -o $d/${base}_$name.o
objs=$(ls otvm_amd/csrc/build/*.o | grep -v "/${base}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $d/libotvm_$name.so $objs $d/${base}_$name.o
echo $d/libotvm_$name.so
