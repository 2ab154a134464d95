#!/bin/bash
# round 3, call u: the paged filter as lean pass (PG_LPR lanes per record) + full pass, A/B of 4 / 2 / 1 lanes; the page writer
# without its serial blob pass; correctness of both under the page and seam tests (product build and the 2-lane build)
set -x
mkdir -p gpurun_out/r03u
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r03u/pytest.txt
cat gpurun_out/r03u/pytest.txt
GCI_LIB_PATH=$PWD/gci_amd/csrc/libgci_hip_lpr2.so timeout 900 python -m pytest tests/test_gpu_pages.py tests/test_gpu_seams.py -x -q -m gpu -k "pages or filter or bam" 2>&1 | tail -6 > gpurun_out/r03u/pytest_lpr2.txt
cat gpurun_out/r03u/pytest_lpr2.txt
GCI_LIB_PATH=$PWD/gci_amd/csrc/libgci_hip_lpr1.so timeout 900 python -m pytest tests/test_gpu_pages.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r03u/pytest_lpr1.txt
cat gpurun_out/r03u/pytest_lpr1.txt
timeout 1200 python tools/exp_k1_pages.py 0.3 product lpr2 lpr1 lpr2w6 product lpr2 > gpurun_out/r03u/k1_variants.txt 2>&1
grep -v amdgpu.ids gpurun_out/r03u/k1_variants.txt
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --ingest-gb 0 > gpurun_out/r03u/bench.json 2> gpurun_out/r03u/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03u/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["kernel_us_per_launch"])
print(json.dumps(d["survey_8d"]["2_device_pipeline_incl_h2d_d2h"]))
print(json.dumps(d["survey_8d"]["3_command_line_chr19_realistic_bam"]))
PY
