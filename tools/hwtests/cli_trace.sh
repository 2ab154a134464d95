# the per-run timeline of the ingestion (GCI_PHASES_TRACE=1) of the command line on two BAMs at a scale of the genome, under variants of
# the environment (GCI_TRACE_AB: JSON list of [label, {env}]): per variant the median rate of the uploads behind the first of a file
mkdir -p /root/repo/gpurun_out/$1
cd /root/repo
AB=${GCI_TRACE_AB:-'[["traced", {}]]'}
GCI_EXP_PROFILE=1 GCI_EXP_NO_ROCPROF=1 GCI_PHASES_TRACE=1 GCI_EXP_SAVE=/tmp/ph GCI_EXP_AB="$AB" timeout ${CLI_TRACE_TIMEOUT:-560} python tools/exp_cli_genome.py ${2:-0.3} 2>&1 | grep -E "rc [0-9]+ wall" | cut -c1-330
python - <<'PY'
import json,glob,statistics
for f in sorted(glob.glob("/tmp/ph/phases_*.json")):
    d=json.load(open(f))
    tr=d["notes"].get("trace",[])
    ups=[t for t in tr if t[0]=="upload"]
    gt=d.get("gpu_trace",[])
    inf=[g for g in gt if g[0].startswith("bgzf_inflate")]
    rates=[u[4]/1e9/max(1e-9,u[3]-u[2]) for u in ups if u[1] not in (0,) and u[4] > 2e9]
    durs=[u[3]-u[2] for u in ups if u[1] not in (0,) and u[4] > 2e9]
    idur=[g[2]-g[1] for g in inf if g[2]-g[1] > 0.03]
    print("%-40s uploads behind the first: median %.1f GB/s (%.3f s per run of %.2f GB); inflate per run median %.3f s; first runs %s GB/s" % (
        f.split("phases_")[1][:-5], statistics.median(rates) if rates else 0, statistics.median(durs) if durs else 0, ups[1][4]/1e9 if len(ups)>1 else 0,
        statistics.median(idur) if idur else 0, ["%.0f" % (u[4]/1e9/max(1e-9,u[3]-u[2])) for u in ups if u[1]==0]))
    # the head of the ingestion: when the first uploads left and arrived, when the first runs were taken / inflated
    head = sorted([t for t in tr if any(isinstance(x, float) and x < 1.0 for x in t[1:])], key=lambda t: min(x for x in t[1:] if isinstance(x, float)))[:10]
    print("    head:", "; ".join("%s %s %s" % (t[0], t[1], " ".join("%.3f" % x for x in t[2:] if isinstance(x, float))) for t in head))
    print("    first inflates:", ["%.3f-%.3f" % (g[1], g[2]) for g in sorted(inf, key=lambda g: g[1])[:4]], "phase clock started at process age", d["notes"].get("process_age_s_when_the_phase_clock_started"))
PY
