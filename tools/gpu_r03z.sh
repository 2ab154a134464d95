#!/bin/bash
# round 3, call z: the inflate kernel with the a primary table for the distance codes against the limits only; zlib byte-equality tests
set -x
mkdir -p gpurun_out/r03z
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python tools/exp_inflate_variants.py 1.0 product nodt dt5 dt7 product nodt > gpurun_out/r03z/inflate_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03z/inflate_variants.txt
