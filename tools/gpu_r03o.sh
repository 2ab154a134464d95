#!/bin/bash
# round 3, call o: the whole GPU suite on the build with the device-side member offsets, then the default bench line
set -x
mkdir -p gpurun_out/r03o
cd /root/repo
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r03o/pytest.txt
cat gpurun_out/r03o/pytest.txt
timeout 1500 python bench.py > gpurun_out/r03o/bench.json 2> gpurun_out/r03o/bench.err
tail -c 1500 gpurun_out/r03o/bench.json
tail -5 gpurun_out/r03o/bench.err
