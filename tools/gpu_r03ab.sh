#!/bin/bash
# round 3, call ab: the bucket join with every entry's name line requested at kernel start, A/B on one box
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r03ab
run() { # name lib
  GCI_LIB_PATH=$2 timeout 900 python bench.py --no-e2e --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_us_per_launch']
print('%-10s step %.3f ms  partition %.1f  join %.1f  build %.1f  parity %s' % ('$1', d['ms_per_step'], k.get('k_part1+k_part2 (radix partition of the join)', 0), k.get('k_join_part', 0), k.get('k_tile_build', 0), d.get('parity_vs_oracle_full_size')))"
}
L=$PWD/gci_amd/csrc
{ run product $L/libgci_hip.so; run pf $L/libgci_hip_pf.so; run product $L/libgci_hip.so; run pf $L/libgci_hip_pf.so; } | tee gpurun_out/r03ab/join_prefetch.txt
GCI_LIB_PATH=$L/libgci_hip_pf.so timeout 900 python -m pytest tests/test_gpu_seams.py -x -q -m gpu -k "join" 2>&1 | tail -3
