#!/bin/bash
# round 3, call v: the whole GPU suite on the head of the round
set -x
mkdir -p gpurun_out/r03v
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r03v/pytest.txt
cat gpurun_out/r03v/pytest.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
