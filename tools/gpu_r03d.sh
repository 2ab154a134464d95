#!/bin/bash
# round 3, call d: leaner paged filter (mad24 CIGAR sums, f32 pre-test of the divisions) -- parity, then A/B on one box
set -x
mkdir -p gpurun_out/r03d
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py -x -q -m gpu -k "pages or bam_filter" 2>&1 | tail -25 > gpurun_out/r03d/pytest.txt
cat gpurun_out/r03d/pytest.txt
for v in pages stream; do
  timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 --k1 $v > gpurun_out/r03d/bench_$v.json 2> gpurun_out/r03d/bench_$v.err
  tail -c 700 gpurun_out/r03d/bench_$v.json
done
GCI_JOIN_NOVERIFY=1 timeout 1200 python bench.py --no-e2e --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/r03d/bench_noverify.json 2> gpurun_out/r03d/bench_noverify.err
tail -c 700 gpurun_out/r03d/bench_noverify.json
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d gpurun_out/pmc_r03d/sq -o pmc -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e > gpurun_out/r03d/pmc_sq.json 2> gpurun_out/r03d/pmc_sq.err
python tools/pmc_summary.py gpurun_out/pmc_r03d > gpurun_out/r03d/pmc_summary.txt 2>&1
grep -A9 "k_bam_filter_pages" gpurun_out/r03d/pmc_summary.txt | head -12
