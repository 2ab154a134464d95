#!/bin/bash
# Alternative builds of k_inflate_wave.hip (each a libgci_hip.so of its own under gci_amd/csrc/build/variants/, cross-compiled in the
# build container: they travel with the snapshot), for tools/hwtests/run_iw_ab.sh on the GPU box.
#   build_iw_variants.sh name1="-DIW_CP_PREFETCH=0" name2="-DIW_CP_RES_UNITS=2" ...      ("head=" + a git ref builds that ref's file)
set -e
cd "$(dirname "$0")/../../gci_amd/csrc"
mkdir -p build/variants
objs=$(ls build/*.hip.o build/*.cpp.o | grep -v k_inflate_wave)
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  src=k_inflate_wave.hip
  if [[ "$flags" == ref:* ]]; then git show "${flags#ref:}:gci_amd/csrc/k_inflate_wave.hip" > build/variants/kiw_$name.hip; src=build/variants/kiw_$name.hip; flags=""; fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -ffp-contract=off -I. -I../../include $flags -c $src -o build/variants/kiw_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/libgci_hip_$name.so $objs build/variants/kiw_$name.o -lz -lpthread
  echo "built build/variants/libgci_hip_$name.so ($flags)"
done
