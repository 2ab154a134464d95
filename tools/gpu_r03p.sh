#!/bin/bash
# round 3, call p: run-ahead uploads in the run-by-run ingestion (tests + 3b), then BASELINE configs[3] (CHM13, --hifi + --nano, one BAM +
# one PAF per read type) at full size on one GPU
set -x
mkdir -p gpurun_out/r03p
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r03p/pytest.txt
cat gpurun_out/r03p/pytest.txt
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r03p/bench_ingest.json 2> gpurun_out/r03p/bench_ingest.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r03p/bench_ingest.json").read().strip().splitlines()[-1])
print(json.dumps(d["survey_8d"]["3b_ingest_genome"]))
print(json.dumps(d["survey_8d"]["3_command_line_chr19_realistic_bam"]))
PY
free -g | head -2
timeout 2400 python bench.py --workload genome4 --steps 5 --warmup 1 > gpurun_out/r03p/genome4.json 2> gpurun_out/r03p/genome4.err
echo "genome4 rc=$?"
tail -c 2500 gpurun_out/r03p/genome4.json
grep -v "group" gpurun_out/r03p/genome4.err | tail -30
