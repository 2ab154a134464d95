#!/bin/bash
set -x
mkdir -p gpurun_out/r03h
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "bench_two_ranks" 2>&1 | grep -v "^E    *$" > gpurun_out/r03h/pytest_dist.txt
grep -n "bench: world\|GciError\|passed\|failed" gpurun_out/r03h/pytest_dist.txt | head -20
