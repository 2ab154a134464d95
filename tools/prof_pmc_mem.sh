#!/bin/bash
# L2 -> memory write-path counters for one command (default: the tile experiment).  kernel-trace + pmc only.
set -u
tag=${1:-mem}
export TMPDIR=/tmp
cmd=${2:-"python tools/exp_tile.py"}
run() { name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}/${name} -o pmc -- $cmd > gpurun_out/pmc_${tag}_${name}.log 2>&1
  echo "$name rc=$?"; }
run ea1 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum
run ea2 TCC_EA0_WRREQ_LEVEL_sum TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_IB_STALL_sum
run ta TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE
