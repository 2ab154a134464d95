#!/bin/bash
# round 3, call m: deflate passes over run lists (k_depth_runs); where k_tile_build's outputs lie
set -x
mkdir -p gpurun_out/r03m
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_seams.py tests/test_gpu_e2e.py -x -q -m gpu -k "gzip or deflate or cli or depth_gz or MH63 or mh63 or reference" 2>&1 | tail -15 > gpurun_out/r03m/pytest.txt
cat gpurun_out/r03m/pytest.txt
timeout 900 python tools/exp_tile_place.py > gpurun_out/r03m/tile_place.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03m/tile_place.txt
timeout 1500 python bench.py --no-cpu-baseline --ingest-gb 0 > gpurun_out/r03m/bench.json 2> gpurun_out/r03m/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r03m/bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['kernel_us_per_launch'])
print(d.get('cli_shaped_step'))
print(d['survey_8d'].get('2_device_pipeline_incl_h2d_d2h'))
print(d['survey_8d'].get('3_command_line_chr19_realistic_bam'))
PY
