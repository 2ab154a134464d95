#!/bin/bash
# round 3, call w: the inflate kernel with more members per CU (sorted symbols out of LDS, more waves per SIMD), A/B on one box;
# the PAF tests with the parser changes
set -x
mkdir -p gpurun_out/r03w
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python tools/exp_inflate_variants.py 1.0 product sg4 sg5 sg6 sg7 sg4@16 sg5@16 sg4@4 sg5@4 sg6@4 product > gpurun_out/r03w/inflate_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03w/inflate_variants.txt
GCI_LIB_PATH=$PWD/gci_amd/csrc/libgci_hip_sg5.so timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -4
