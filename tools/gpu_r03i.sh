#!/bin/bash
# round 3, call i: the whole GPU suite on the current build (CLI through record pages, sharded join, two-type workloads)
set -x
mkdir -p gpurun_out/r03i
cd /root/repo
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r03i/pytest_all.txt
cat gpurun_out/r03i/pytest_all.txt
