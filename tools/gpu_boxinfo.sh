#!/bin/bash
# which box is this: partition modes, clocks, and the tile build's time on it
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/boxinfo
tag=$(date +%H%M%S)
{
  for f in /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/available_memory_partition; do echo "$f: $(cat $f 2>/dev/null)"; done
  rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -v "^$" | head -20
  rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk" | head -8
  rocm-smi --showpower --showtemp 2>&1 | grep -E "Power|Temp" | head -6
  timeout 600 python bench.py --steps 10 --warmup 2 --no-e2e --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step', round(d['ms_per_step'],3), 'k_tile_build', d['roofline']['avg_launch_ms'], 'kernels', d['kernel_us_per_launch'])"
  rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk" | head -4
} > gpurun_out/boxinfo/$tag.txt 2>&1
cat gpurun_out/boxinfo/$tag.txt
