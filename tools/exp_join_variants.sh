#!/bin/bash
# A/B of the bucket join's geometry on one box: threads per bucket workgroup, entries per thread, slots per table
cd /root/repo
run() { # name lib slots
  GCI_LIB_PATH=$2 GCI_JOIN_SLOTS=$3 timeout 900 python bench.py --no-e2e --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d['kernel_us_per_launch']
print('%-22s step %.3f ms  partition %.1f  join %.1f  build %.1f  parity %s' % ('$1', d['ms_per_step'], k.get('k_part1+k_part2 (radix partition of the join)', 0), k.get('k_join_part', 0), k.get('k_tile_build', 0), d.get('parity_vs_oracle_full_size')))"
}
L=$PWD/gci_amd/csrc
run product-s1024 $L/libgci_hip.so 1024
run product-s512 $L/libgci_hip.so 512
run j256e2-s512 $L/libgci_hip_j256e2.so 512
run j256e2-s1024 $L/libgci_hip_j256e2.so 1024
run j256e1-s512 $L/libgci_hip_j256e1.so 512
run j512e1-s1024 $L/libgci_hip_j512e1.so 1024
timeout 900 python tools/exp_tile_place.py
