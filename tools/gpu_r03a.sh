#!/bin/bash
# round 3, call a: the rewritten partitioned join (wide entries, deferred name confirmation) -- parity, then timing
set -x
mkdir -p gpurun_out/r03a
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_seams.py -x -q -m gpu -k "join or counting" 2>&1 | tail -15 > gpurun_out/r03a/pytest_join.txt
cat gpurun_out/r03a/pytest_join.txt
timeout 900 python -m pytest tests/test_gpu_genome.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r03a/pytest_genome.txt
cat gpurun_out/r03a/pytest_genome.txt
./tools/hwtests/random_read 4096 > gpurun_out/r03a/random_read.txt 2>&1
./tools/hwtests/random_read 128 >> gpurun_out/r03a/random_read.txt 2>&1
cat gpurun_out/r03a/random_read.txt
timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r03a/bench.json 2> gpurun_out/r03a/bench.err
tail -c 3000 gpurun_out/r03a/bench.json; tail -5 gpurun_out/r03a/bench.err
