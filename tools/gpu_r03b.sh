#!/bin/bash
# round 3, call b: record pages -- converter vs the Python statement, paged filter vs the stream filter, then timing
set -x
mkdir -p gpurun_out/r03b
cd /root/repo
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_pages.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r03b/pytest_pages.txt
cat gpurun_out/r03b/pytest_pages.txt
timeout 1200 python -m pytest tests/test_gpu_seams.py -x -q -m gpu -k "bam_filter" 2>&1 | tail -25 > gpurun_out/r03b/pytest_filter.txt
cat gpurun_out/r03b/pytest_filter.txt
for k1 in pages stream; do
  timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 --k1 $k1 > gpurun_out/r03b/bench_$k1.json 2> gpurun_out/r03b/bench_$k1.err
  tail -c 1500 gpurun_out/r03b/bench_$k1.json; tail -3 gpurun_out/r03b/bench_$k1.err
done
