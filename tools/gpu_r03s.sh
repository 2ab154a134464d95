#!/bin/bash
# round 3, call r: BASELINE configs[4] (diploid mat + pat, HiFi 100x + ONT 20x, gaps, regions) at full size on one GPU
set -x
mkdir -p gpurun_out/r03s
cd /root/repo
export TMPDIR=/tmp
(while true; do cat /sys/fs/cgroup/memory.current; sleep 20; done) > gpurun_out/r03s/mem_trace.txt 2>&1 &
MON=$!
timeout 2700 python bench.py --workload diploid --steps 5 --warmup 1 > gpurun_out/r03s/diploid.json 2> gpurun_out/r03s/diploid.err
echo "diploid rc=$?"
kill $MON
tail -c 3000 gpurun_out/r03s/diploid.json
grep -v "part " gpurun_out/r03s/diploid.err | tail -30
sort -n gpurun_out/r03s/mem_trace.txt | tail -1
