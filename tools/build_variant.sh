#!/bin/bash
# An experimental build of the library next to the product one: tools/build_variant.sh NAME FILE.hip "-DFLAG ..."
# -> gci_amd/csrc/libgci_hip_NAME.so (FILE.hip recompiled with the flags, every other object of the product build reused).
# Run a tool with GCI_LIB_PATH=.../libgci_hip_NAME.so to use it (A/B within ONE gpurun call: boxes differ by 20 %).
set -e
name=$1; file=$2; flags=$3
cd "$(dirname "$0")/../gci_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off $flags -c $file -o build/${file}.${name}.o
objs=""
for o in build/*.hip.o build/*.cpp.o; do
  case $o in build/${file}.o) objs="$objs build/${file}.${name}.o";; *) objs="$objs $o";; esac
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgci_hip_${name}.so $objs -lz -lpthread
echo "built libgci_hip_${name}.so"
