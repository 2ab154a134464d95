# round 6, the last full run: the GPU suite, smoke() with and without torch in the process, the default bench line, rocprofv3 --stats of the step
export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 2>&1 | tail -10) > gpurun_out/r06o_gputests.txt 2>&1; tail -2 gpurun_out/r06o_gputests.txt
python -c "import __graft_entry__ as g; g.smoke(); import sys; print('torch in the process:', 'torch' in sys.modules)" 2>&1 | tail -2
python -c "import torch, __graft_entry__ as g; g.smoke(); import sys; print('torch in the process:', 'torch' in sys.modules)" 2>&1 | tail -2
timeout 1100 python bench.py > gpurun_out/r06o_bench.json 2> gpurun_out/r06o_bench.err; echo bench rc=$?
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r06o_stats -o bench -- python $OLDPWD/bench.py --steps 20 --warmup 3 --only-step > $OLDPWD/gpurun_out/r06o_bench_step_only.json 2> $OLDPWD/gpurun_out/r06o_bench_rocprof.err); echo "stats rc=$?"
f=$(find gpurun_out/r06o_stats -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r06o_bench_step_only_kernel_stats.csv; head -6 $f | cut -c1-60,150-250
find gpurun_out/r06o_stats -name "*kernel_trace.csv" -delete
python - <<'PY'
import json
for fn in ("gpurun_out/r06o_bench.json", "gpurun_out/r06o_bench_step_only.json"):
    d = json.load(open(fn)); c = d["config"]
    print(fn, round(d["value"]), round(d["ms_per_step"], 3), {k: (round(c[k], 3) if isinstance(c[k], float) else c[k]) for k in c if k[:2] in ("n1", "n2", "n3")}, "frac", round(d["roofline"]["frac"], 4), "avg ms", round(d["roofline"]["avg_launch_ms"], 4), "step", round(d["step_roofline"]["frac"], 4))
    print({k: v for k, v in d["kernel_us_per_launch"].items()})
PY
