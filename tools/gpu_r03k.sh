#!/bin/bash
# round 3, call k: how fast the lean path alone would be (occupancy, page size); then the bench line with its new legs
set -x
mkdir -p gpurun_out/r03k
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python tools/exp_k1_pages.py 0.3 product product@16384 product@32768 lo5 lo6 lo8 lo6@16384 lo8@16384 lo8@12288 > gpurun_out/r03k/k1_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03k/k1_variants.txt
timeout 2400 python bench.py > gpurun_out/r03k/bench.json 2> gpurun_out/r03k/bench.err
tail -c 6000 gpurun_out/r03k/bench.json; tail -5 gpurun_out/r03k/bench.err
