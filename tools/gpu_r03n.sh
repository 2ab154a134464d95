#!/bin/bash
# round 3, call n: the bucket join without its output atomic (sparse slots + gather); kernel stats of the whole bench command
set -x
mkdir -p gpurun_out/r03n
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_seams.py tests/test_gpu_genome.py -x -q -m gpu -k "join or counting or genome_dual" 2>&1 | tail -8 > gpurun_out/r03n/pytest.txt
cat gpurun_out/r03n/pytest.txt
timeout 900 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r03n/bench.json 2> gpurun_out/r03n/bench.err
tail -c 800 gpurun_out/r03n/bench.json
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r03n/stats -o bench -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-e2e > /root/repo/gpurun_out/r03n/bench_rocprof.json 2> /root/repo/gpurun_out/r03n/bench_rocprof.err)
find gpurun_out/r03n/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'head -30 {}'
