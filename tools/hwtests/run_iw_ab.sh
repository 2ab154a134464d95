#!/bin/bash
# On the GPU box: every build of build_iw_variants.sh through tools/hwtests/inflate_product.py under rocprofv3 (kernel averages), twice
# round robin so that a drifting box shows; then the product build with the in-kernel phase counters (GCI_IW_PROF=1).
#   run_iw_ab.sh tag [scale]
tag=${1:-iw_ab}; scale=${2:-0.5}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out/$tag
cd /tmp; export TMPDIR=/tmp
for round in 1 2; do
  for lib in $root/gci_amd/csrc/build/variants/libgci_hip_*.so; do
    name=$(basename $lib .so); name=${name#libgci_hip_}
    rm -rf /tmp/prof_$name
    GCI_LIB_PATH=$lib CHECK_CRC=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -o iw -- python $root/tools/hwtests/inflate_product.py $scale 4 > /tmp/log_$name.txt 2>&1
    f=$(find /tmp/prof_$name -name "*kernel_stats.csv" | head -1)
    echo "round $round $name: $(grep -E 'k_inflate_symbols|k_inflate_copy' $f | awk -F'",' '{n=split($1,a,"("); gsub(/"/,"",a[1]); split($2,b,","); printf "%s avg %.3f ms; ", a[1], b[3]/1e6}') $(grep -o 'calls .* GB/s out; equal to the stream: [A-Za-z]*' /tmp/log_$name.txt | tail -1)"
  done
done 2>&1 | tee $root/gpurun_out/$tag/ab.txt
GCI_IW_PROF=1 CHECK_CRC=0 timeout 200 python $root/tools/hwtests/inflate_product.py $scale 2 2>&1 | grep -E "iw prof|mode" | tail -3 | tee $root/gpurun_out/$tag/prof.txt
