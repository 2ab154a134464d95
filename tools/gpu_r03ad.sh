#!/bin/bash
# round 3, call ad: per-kernel times of configs[3] at full size (the paged filter and k_cigar_chunks on 40.7 GB of ONT records)
set -x
mkdir -p gpurun_out/r03ad
cd /root/repo
export TMPDIR=/tmp
(cd /tmp && timeout 2400 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03ad/stats -o g4 -- python /root/repo/bench.py --workload genome4 --steps 5 --warmup 1 > /root/repo/gpurun_out/r03ad/genome4.json 2> /root/repo/gpurun_out/r03ad/genome4.err); echo "rc=$?"
find gpurun_out/r03ad/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03ad/genome4_kernel_stats.csv
rm -rf gpurun_out/r03ad/stats
head -25 gpurun_out/r03ad/genome4_kernel_stats.csv | cut -c1-160
