#!/bin/bash
# round 3, call r: BASELINE configs[3] (CHM13, --hifi + --nano, one BAM + one PAF per read type) at full size on one GPU
set -x
mkdir -p gpurun_out/r03r
cd /root/repo
export TMPDIR=/tmp
(while true; do cat /sys/fs/cgroup/memory.current; sleep 20; done) > gpurun_out/r03r/mem_trace.txt 2>&1 &
MON=$!
timeout 2700 python bench.py --workload genome4 --steps 5 --warmup 1 > gpurun_out/r03r/genome4.json 2> gpurun_out/r03r/genome4.err
echo "genome4 rc=$?"
kill $MON
tail -c 3000 gpurun_out/r03r/genome4.json
grep -v "part " gpurun_out/r03r/genome4.err | tail -30
sort -n gpurun_out/r03r/mem_trace.txt | tail -1
