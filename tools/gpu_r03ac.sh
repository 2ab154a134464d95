#!/bin/bash
# round 3, call ac: lanes per record in the lean pass again, now with the busy waves rotating with the page (all four SIMDs busy)
set -x
mkdir -p gpurun_out/r03ac
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python tools/exp_k1_pages.py 0.3 product lpr2 lpr1 lpr2w6 product lpr2 > gpurun_out/r03ac/k1_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03ac/k1_variants.txt
GCI_LIB_PATH=$PWD/gci_amd/csrc/libgci_hip_lpr2.so timeout 900 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py -x -q -m gpu -k "pages or filter or bam" 2>&1 | tail -3
