#!/bin/bash
# round 3, call x2: what costs 18 ms when bam_join_input returns; the bench contract test with its new leg
set -x
mkdir -p gpurun_out/r03x
cd /root/repo
export TMPDIR=/tmp
timeout 600 python tools/exp_cli_teardown.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03x/teardown.txt
timeout 900 python -m pytest tests/test_gpu_bench.py -x -q -m gpu 2>&1 | tail -4
