#!/bin/bash
# round 3, call q: PAF by byte range (single-GPU stages + the two-rank command line), run-ahead uploads
set -x
mkdir -p gpurun_out/r03q
cd /root/repo
export TMPDIR=/tmp
(cat /sys/fs/cgroup/memory.max; cat /sys/fs/cgroup/memory.high; grep -E "MemTotal|MemAvailable" /proc/meminfo; nproc; cat /sys/fs/cgroup/cpu.max) > gpurun_out/r03q/box.txt 2>&1
cat gpurun_out/r03q/box.txt
timeout 1200 python -m pytest tests/test_gpu_paf.py tests/test_gpu_e2e.py tests/test_gpu_inflate.py tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r03q/pytest.txt
cat gpurun_out/r03q/pytest.txt
