# the last call of round 6: where the host's memory and the GPU sit, the new tests, the default bench line, rocprofv3 --stats of the step
export TMPDIR=/tmp; mkdir -p gpurun_out
(lscpu | grep -i "numa\|socket\|model name\|^CPU(s)"; for d in /sys/class/drm/card*/device; do echo "$d numa_node $(cat $d/numa_node 2>/dev/null)"; done; grep -i "MemTotal\|Shmem:" /proc/meminfo; cat /sys/devices/system/node/node*/meminfo 2>/dev/null | grep -i "MemTotal\|MemFree\|Shmem:" ) > gpurun_out/r06l_host.txt 2>&1; cat gpurun_out/r06l_host.txt | head -24
(timeout 600 python -m pytest tests/test_gpu_native.py tests/test_gpu_paf.py -m gpu -x -q 2>&1 | tail -4)
timeout 1100 python bench.py > gpurun_out/r06l_bench.json 2> gpurun_out/r06l_bench.err; echo bench rc=$?
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r06l_stats -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 3 --only-step > $OLDPWD/gpurun_out/r06l_bench_step_only.json 2> $OLDPWD/gpurun_out/r06l_bench_rocprof.err); echo "stats rc=$?"
f=$(find gpurun_out/r06l_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06l_bench_step_only_kernel_stats.csv; head -8 $f | cut -c1-90,150-260
find gpurun_out/r06l_stats -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
for fn in ("gpurun_out/r06l_bench.json", "gpurun_out/r06l_bench_step_only.json"):
    d = json.load(open(fn)); c = d["config"]
    print(fn, round(d["value"]), round(d["ms_per_step"], 3), {k: (round(c[k], 3) if isinstance(c[k], float) else c[k]) for k in c if k[:2] in ("n1", "n2", "n3")}, "frac", round(d["roofline"]["frac"], 4), "avg ms", round(d["roofline"]["avg_launch_ms"], 4), "traffic", d["roofline"]["traffic"], "step", round(d["step_roofline"]["frac"], 4))
PY
