#!/bin/bash
# round 3, call c: staged scatters + register-resident bucket join + ref table in LDS for the paged filter; then PMC
set -x
mkdir -p gpurun_out/r03c
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py tests/test_gpu_genome.py -x -q -m gpu -k "pages or join or counting or bam_filter or genome" 2>&1 | tail -25 > gpurun_out/r03c/pytest.txt
cat gpurun_out/r03c/pytest.txt
timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r03c/bench.json 2> gpurun_out/r03c/bench.err
tail -c 1200 gpurun_out/r03c/bench.json; tail -3 gpurun_out/r03c/bench.err
bash tools/prof_pmc.sh r03c > gpurun_out/r03c/pmc.log 2>&1
tail -5 gpurun_out/r03c/pmc.log
