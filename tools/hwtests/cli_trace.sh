# the per-run timeline of the ingestion (GCI_PHASES_TRACE=1) of the command line on two 0.3-genome BAMs: uploads against device stages
mkdir -p /root/repo/gpurun_out/$1
cd /root/repo
GCI_EXP_PROFILE=1 GCI_EXP_NO_ROCPROF=1 GCI_PHASES_TRACE=1 GCI_EXP_SAVE=/tmp/ph GCI_EXP_AB='[["traced", {}]]' timeout 500 python tools/exp_cli_genome.py ${2:-0.3} 2>&1 | grep -E "rc [0-9]+ wall" | cut -c1-300
python - <<'PY'
import json,glob
for f in sorted(glob.glob("/tmp/ph/phases_traced.json")):
    d=json.load(open(f))
    tr=d["notes"].get("trace",[])
    ups=[t for t in tr if t[0]=="upload"]
    gt=d.get("gpu_trace",[])
    inf=[g for g in gt if g[0].startswith("bgzf_inflate")]
    print("run: upload begin end GB/s | inflate begin end")
    for k,u in enumerate(ups):
        i=inf[k] if k < len(inf) else ["",0,0]
        print("%2d  %.3f %.3f %5.1f | %.3f %.3f" % (u[1], u[2], u[3], u[4]/1e9/max(1e-9,u[3]-u[2]), i[1], i[2]))
    others=[t for t in tr if t[0]!="upload"][:12]
    print("other trace kinds:", sorted({t[0] for t in tr}))
PY
