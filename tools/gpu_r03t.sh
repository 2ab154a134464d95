#!/bin/bash
# round 3, call t: the round's profile evidence in one call on one box -- the default bench line, rocprofv3 --kernel-trace --stats of
# the same command, the PMC passes (one counter group per run, kernel-trace only) and the HBM traffic of the dominant kernel
set -x
tag=r03t
mkdir -p gpurun_out/$tag
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/$tag/stats -o bench -- python /root/repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e > /root/repo/gpurun_out/$tag/bench_rocprof.json 2> /root/repo/gpurun_out/$tag/bench_rocprof.err); echo "stats rc=$?"
timeout 1800 bash tools/prof_pmc.sh $tag
python tools/pmc_summary.py gpurun_out/pmc_$tag > gpurun_out/$tag/pmc_summary.txt 2>&1
find gpurun_out/$tag/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/$tag/bench_kernel_stats.csv
rm -rf gpurun_out/$tag/stats/*/*kernel_trace.csv
du -sh gpurun_out/$tag gpurun_out/pmc_$tag
tail -c 600 gpurun_out/$tag/bench.json
head -30 gpurun_out/$tag/bench_kernel_stats.csv
