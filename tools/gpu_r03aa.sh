#!/bin/bash
# round 3, call aa: runs of a large file sized to whole decode rounds; ingest at genome size with 4 GiB and 8 GiB runs
set -x
mkdir -p gpurun_out/r03aa
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_inflate.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -4
python - <<'PY'
from gci_amd.device import Engine
print("decode round:", Engine(0).inflate_round(), "members")
PY
for chunk in 4294967296 8589934592; do
  GCI_BAM_CHUNK_BYTES=$chunk timeout 900 python - <<'PY'
import json, os, sys
sys.argv = ["bench.py"]
import bench
out = bench.ingest_number(40.0, 64.0)
print("chunk", os.environ["GCI_BAM_CHUNK_BYTES"], json.dumps({k: out[k] for k in ("seconds", "member_table_seconds", "gb_per_s_in", "gb_per_s_out", "records", "records_expected")}))
PY
done 2>&1 | grep -v amdgpu | tee gpurun_out/r03aa/ingest.txt
