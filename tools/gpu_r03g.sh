#!/bin/bash
set -x
mkdir -p gpurun_out/r03g
cd /root/repo
export TMPDIR=/tmp
free -g | head -3; nproc
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "bench_two_ranks and weak" 2>&1 | grep -v "^E    *$" > gpurun_out/r03g/pytest_dist.txt
grep -n "Error\|error\|Traceback\|File \"/tmp/code" gpurun_out/r03g/pytest_dist.txt | head -40
tail -5 gpurun_out/r03g/pytest_dist.txt
timeout 900 python bench.py --workload genome4 --scale 0.02 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03g/bench_genome4_small.json 2> gpurun_out/r03g/bench_genome4_small.err
tail -c 1500 gpurun_out/r03g/bench_genome4_small.json; tail -20 gpurun_out/r03g/bench_genome4_small.err
timeout 900 python bench.py --workload diploid --scale 0.02 --steps 3 --warmup 1 --no-e2e --no-cpu-baseline > gpurun_out/r03g/bench_diploid_small.json 2> gpurun_out/r03g/bench_diploid_small.err
tail -c 1500 gpurun_out/r03g/bench_diploid_small.json; tail -20 gpurun_out/r03g/bench_diploid_small.err
