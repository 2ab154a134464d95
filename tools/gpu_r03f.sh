#!/bin/bash
# round 3, call f: tight aux walk + multilinear name hash in the paged filter, fold payload in LDS, the sharded join on two ranks
set -x
mkdir -p gpurun_out/r03f
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py tests/test_gpu_genome.py tests/test_gpu_paf.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03f/pytest.txt
cat gpurun_out/r03f/pytest.txt
timeout 1200 python tools/exp_k1_pages.py 0.3 product noaux nohash nodecide none > gpurun_out/r03f/k1_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03f/k1_variants.txt
timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err
tail -c 700 gpurun_out/r03f/bench.json
timeout 2400 python -m pytest tests/test_gpu_dist.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r03f/pytest_dist.txt
cat gpurun_out/r03f/pytest_dist.txt
