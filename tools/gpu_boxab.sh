#!/bin/bash
# does the store policy of k_tile_build explain the two kinds of boxes (3.4 ms / 4.1 ms)?  product (track non-temporal, text plain),
# all plain, all non-temporal -- on whatever box this call gets
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/boxab
tag=$(date +%H%M%S)
{
  cat /sys/module/amdgpu/version 2>/dev/null; uname -r
  rocm-smi --showfwinfo 2>/dev/null | grep -E "GPU\[0\]" | grep -E "MEC|SMC|SDMA|VBIOS|MC |RLC:" | head -8
  for v in product plain allnt product; do
    if [ $v = product ]; then unset GCI_LIB_PATH; else export GCI_LIB_PATH=$PWD/gci_amd/csrc/libgci_hip_$v.so; fi
    timeout 600 python bench.py --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'ms_per_step', round(d['ms_per_step'],3), 'k_tile_build', round(d['roofline']['avg_launch_ms'],3), 'K1', d['kernel_us_per_launch']['k_bam_filter'], 'parity', d.get('parity_vs_oracle_full_size'))"
  done
} > gpurun_out/boxab/$tag.txt 2>&1
cat gpurun_out/boxab/$tag.txt
