#!/bin/bash
# round 3, call e: NAME16 name comparisons in the join; where the paged filter's time goes (variants with parts compiled out)
set -x
mkdir -p gpurun_out/r03e
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py tests/test_gpu_genome.py -x -q -m gpu -k "pages or join or counting or bam_filter or genome" 2>&1 | tail -25 > gpurun_out/r03e/pytest.txt
cat gpurun_out/r03e/pytest.txt
timeout 1200 python tools/exp_k1_pages.py 0.3 > gpurun_out/r03e/k1_variants.txt 2>&1
cat gpurun_out/r03e/k1_variants.txt
timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r03e/bench.json 2> gpurun_out/r03e/bench.err
tail -c 700 gpurun_out/r03e/bench.json
GCI_JOIN_NOVERIFY=1 timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r03e/bench_noverify.json 2> gpurun_out/r03e/bench_noverify.err
tail -c 700 gpurun_out/r03e/bench_noverify.json
