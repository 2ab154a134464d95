#!/bin/bash
# PMC passes over bench.py (run on the GPU box via gpurun).  One counter group per run, kernel-trace only
# (never combined with sys/hip/hsa tracing).  Output: gpurun_out/pmc_<tag>/<group>/..._counter_collection.csv
set -u
tag=${1:-r01}
export TMPDIR=/tmp
cmd="python bench.py --steps 3 --warmup 1 --only-step"
run() { # name, counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}/${name} -o pmc -- $cmd > gpurun_out/pmc_${tag}_${name}.json 2> gpurun_out/pmc_${tag}_${name}.err
  echo "$name rc=$?"
}
run sq SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
find gpurun_out/pmc_${tag} -name "*.csv" | head -20
