#!/bin/bash
# round 3, call p: BASELINE configs[3] (CHM13, --hifi + --nano, one BAM + one PAF per read type) at full size on one GPU
set -x
mkdir -p gpurun_out/r03p
cd /root/repo
export TMPDIR=/tmp
free -g | head -2
timeout 2400 python bench.py --workload genome4 --steps 5 --warmup 1 > gpurun_out/r03p/genome4.json 2> gpurun_out/r03p/genome4.err
echo "genome4 rc=$?"
tail -c 2500 gpurun_out/r03p/genome4.json
grep -v "group" gpurun_out/r03p/genome4.err | tail -30
