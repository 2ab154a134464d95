#!/bin/bash
# round 3, call j: lean common path of the paged filter (A/B against the full path only, and occupancy variants); the whole GPU suite
set -x
mkdir -p gpurun_out/r03j
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py -x -q -m gpu -k "pages or bam_filter" 2>&1 | tail -15 > gpurun_out/r03j/pytest_k1.txt
cat gpurun_out/r03j/pytest_k1.txt
timeout 1200 python tools/exp_k1_pages.py 0.3 product nolean w5 w4 > gpurun_out/r03j/k1_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03j/k1_variants.txt
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 > gpurun_out/r03j/pytest_all.txt
cat gpurun_out/r03j/pytest_all.txt
