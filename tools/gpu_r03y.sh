#!/bin/bash
# round 3, call y: the command line after the threaded member table and the background unmapping; ingest at genome size; tests that touch both
set -x
mkdir -p gpurun_out/r03y
cd /root/repo
export TMPDIR=/tmp
timeout 600 python tools/exp_cli_teardown.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03y/teardown.txt
timeout 600 python tools/exp_cli_profile.py 2>&1 | grep -v amdgpu | head -40 > gpurun_out/r03y/cli_profile.txt
head -12 gpurun_out/r03y/cli_profile.txt
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_e2e.py tests/test_gpu_paf.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03y/bench.json 2> gpurun_out/r03y/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03y/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(json.dumps(d["survey_window_step"]))
for k, v in d["survey_8d"].items(): print(k, json.dumps(v)[:700])
PY
