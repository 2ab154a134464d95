#!/bin/bash
# round 3, last call: the whole GPU suite and the default bench line on the head of the round, one box
set -x
mkdir -p gpurun_out/r03final
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r03final/pytest.txt
cat gpurun_out/r03final/pytest.txt
timeout 1500 python bench.py > gpurun_out/r03final/bench.json 2> gpurun_out/r03final/bench.err
tail -c 1200 gpurun_out/r03final/bench.json
