export TMPDIR=/tmp; mkdir -p gpurun_out
cmd="python bench.py --steps 3 --warmup 1 --only-step"
for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OLDPWD/gpurun_out/pmc_r06/$c -o pmc -- python $OLDPWD/bench.py --steps 3 --warmup 1 --only-step > $OLDPWD/gpurun_out/pmc_r06_$c.json 2> $OLDPWD/gpurun_out/pmc_r06_$c.err); echo "$c rc=$?"
done
python tools/make_traffic.py gpurun_out/pmc_r06 r06 genome; cp profiles/r06_genome_traffic.json gpurun_out/
find gpurun_out/pmc_r06 -name "*kernel_trace.csv" -delete; find gpurun_out/pmc_r06 -name "*counter_collection.csv" -size +20M -delete
echo ==== RANKS
python - <<'PY'
import json, os, subprocess, sys, tempfile, time, shutil
sys.path.insert(0, os.getcwd())
from gci_amd import workloads, synth, hostio
inp = workloads.genome_dual(0.05, 40.0, contigs=synth.CHM13)
tmp = tempfile.mkdtemp(prefix="gci_ranks_", dir="/dev/shm")
from gci_amd.formats import bam as bamfmt
bams = []
for k, f in enumerate(inp.files):
    p = os.path.join(tmp, "a%d.bam" % k)
    workloads.write_bgzf_from_heads(p, f.stream, f.offsets, seed=20250919 + k)
    bamfmt.write_bai(p + ".bai", len(inp.contigs), bam_path=p)
    bams.append(p)
fa = os.path.join(tmp, "ref.fa"); synth.write_reference_fasta(fa, inp.contigs)
for label, extra, gpus in (("one process", {}, []), ("--gpus 2, ranks started directly", {}, ["--gpus", "2"]), ("--gpus 2 under torch.distributed.run", {"GCI_LAUNCHER": "torchrun"}, ["--gpus", "2"]), ("--gpus 2, direct again", {}, ["--gpus", "2"])):
    od, ph = os.path.join(tmp, "out"), os.path.join(tmp, "ph.json")
    shutil.rmtree(od, ignore_errors=True)
    env = dict(os.environ, GCI_PHASES=ph, PYTHONPATH=os.getcwd(), GCI_DIST_BACKEND="gloo", GCI_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(extra)
    t0 = time.time()
    r = subprocess.run([sys.executable, "GCI.py"] + gpus + ["-r", fa, "--hifi"] + bams + ["-d", od, "-t", "32"], env=env, capture_output=True, text=True)
    w = time.time() - t0
    d = json.load(open(ph)) if os.path.exists(ph) else {}
    print("%-40s rc %d wall %.2f s | in front of the phase log %.2f | rank_start %s" % (label, r.returncode, w, d.get("notes", {}).get("process_age_s_when_the_phase_clock_started", -1), d.get("notes", {}).get("rank_start")), flush=True)
    if r.returncode: print(r.stderr[-800:])
shutil.rmtree(tmp, ignore_errors=True)
PY
echo ==== DIPLOID
timeout 900 python bench.py --workload diploid --steps 5 --warmup 2 > gpurun_out/r06k_bench_diploid.json 2> gpurun_out/r06k_bench_diploid.err; echo diploid rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r06k_bench_diploid.json')); print(d['ms_per_step'], d['value'], d.get('parity_vs_oracle_full_size'), d.get('plot_front_end_n3',{}).get('parity_vs_oracle'))"
echo ==== GENOME4 published PAF sizes
timeout 1500 python bench.py --workload genome4 --cli --paf-published-gb 3.6,48 --steps 3 --warmup 1 > gpurun_out/r06k_bench_genome4_cli_published_pafs.json 2> gpurun_out/r06k_bench_genome4.err; echo genome4 rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r06k_bench_genome4_cli_published_pafs.json')); g=d['survey_8d']['3_command_line_two_read_types']; print(d['ms_per_step'], d.get('parity_vs_oracle_full_size')); print({k:g[k] for k in g if k in ('seconds','seconds_first_pass_over_freshly_written_files','parity','files','error','phases_wall_s','phases_device_s')})"
tail -5 gpurun_out/r06k_bench_genome4.err | cut -c1-300
