"""The contig-sharded command line with two ranks: `python -m torch.distributed.run --nproc-per-node 2
GCI.py ...` must write the files -- and rank 0 print the transcript -- of a single process, i.e. of the reference
(tests/golden/*).  Run on ONE GPU: both ranks use device 0 (GCI_DIST_DEVICE) and the collectives go through host memory
(GCI_DIST_BACKEND=gloo; RCCL refuses two ranks on one device), so every kernel, the per-contig ingestion through the
BAM index, the record all-gather, the replicated join with its ownership filter, the gathered outputs and the integer
all-reduce of the mean depth run for real."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from golden_util import GOLDEN, cli_args, expected, images, manifest, read_outputs

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_two_ranks(argv, port):
    env = dict(os.environ, GCI_DIST_BACKEND="gloo", GCI_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + (["--tee", "3", "--log-dir", os.environ["GCI_TEST_TLOG"]] if os.environ.get("GCI_TEST_TLOG") else []) + [
           os.path.join(ROOT, "GCI.py")] + argv[1:]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.parametrize("k,case", list(enumerate(["c3_two_bam", "c3_three_bam_chrs", "c4_two_paf", "c5_two_type", "c6_plot", "c7_t2t_geometry"])))
def test_two_ranks_reproduce_the_reference(k, case, tmp_path):
    out = str(tmp_path / "out")
    r = run_two_ranks(cli_args(case, out), 29710 + k)
    if r.returncode != 0:                                   # keep the whole story of both ranks
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "dist_%s.err" % case), "w").write(r.stderr)
    assert r.returncode == 0, r.stderr[-3000:]
    got, want = read_outputs(out), expected(case)
    assert sorted(got) == sorted(want)
    for fn in want:
        assert got[fn] == want[fn], fn
    got_img, want_img = images(out), images(os.path.join(GOLDEN, case, "expected"))
    assert sorted(got_img) == sorted(want_img)
    for fn in want_img:
        assert np.array_equal(got_img[fn], want_img[fn]), fn
    # the gloo backend's own chatter, written by both ranks past Python's buffering (its pieces can interleave)
    stdout = re.sub(r"\[Gloo\] Rank \d+ is connected to| \d+ peer ranks\. Expected number of connected peer ranks is : \d+\n", "", r.stdout)
    first, _, rest = stdout.partition("\n")
    assert first.startswith("Used arguments:{")
    inp = os.path.join(GOLDEN, case, "inputs")
    assert rest.replace(out, "{OUT}").replace(inp, "{IN}") == manifest(case)["stdout"]


@pytest.mark.parametrize("launcher", ["direct", "torchrun"])
def test_gpus_switch_starts_the_ranks_itself(launcher, tmp_path):
    """`python GCI.py --gpus 2 ...` (this implementation's own switch): the process starts two ranks of itself -- directly, with the
    env:// rendezvous variables (cli._spawn_ranks; round 5 re-executed itself under torch.distributed.run, kept as
    GCI_LAUNCHER=torchrun) -- and they write the reference's files; what rank 0 spent before its first kernel is in its phase log."""
    import json
    case = "c3_two_bam"
    out, ph = str(tmp_path / "out"), str(tmp_path / "phases.json")
    argv = cli_args(case, out)
    env = dict(os.environ, GCI_DIST_BACKEND="gloo", GCI_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, GCI_PHASES=ph)
    if launcher == "torchrun":
        env["GCI_LAUNCHER"] = "torchrun"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "GCI.py"), "--gpus", "2"] + argv[1:], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    got, want = read_outputs(out), expected(case)
    assert sorted(got) == sorted(want)
    for fn in want:
        assert got[fn] == want[fn], fn
    start = json.load(open(ph))["notes"]["rank_start"]
    assert start["world"] == 2 and start["init_process_group_s"] > 0
    assert (start["s_since_the_launcher_started_the_ranks"] is not None) == (launcher == "direct")


def test_a_failing_rank_ends_the_run(tmp_path):
    """One rank of a directly started run dies (an unreadable input on every rank here: both exit with the reference's message): the
    launcher leaves with a non-zero status instead of waiting for ever."""
    argv = cli_args("c3_two_bam", str(tmp_path / "out"))
    argv[argv.index("-r") + 1] = str(tmp_path / "no_such_reference.fa")
    env = dict(os.environ, GCI_DIST_BACKEND="gloo", GCI_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "GCI.py"), "--gpus", "2"] + argv[1:], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "is not an available file" in r.stderr


def test_two_ranks_fragmented_assembly(tmp_path):
    """2 500 scaffolds over two ranks (tests/frag_util.py): the sharded run writes the files of the one-process run (which
    tests/test_gpu_e2e.py holds against the oracle) -- LPT packing of thousands of contigs, the per-contig ingestion through
    the index for 1 250 contigs per rank, gathers of thousands of small pieces."""
    import frag_util
    inp = str(tmp_path / "in")
    _, args = frag_util.write_inputs(inp)
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    r1 = subprocess.run([sys.executable, os.path.join(ROOT, "GCI.py")] + args + ["-d", one], capture_output=True, text=True, timeout=900,
                        env=dict(os.environ, PYTHONPATH=ROOT))
    assert r1.returncode == 0, r1.stderr[-3000:]
    r2 = run_two_ranks(["GCI.py"] + args + ["-d", two], 29790)
    assert r2.returncode == 0, r2.stderr[-3000:]
    a, b = read_outputs(one), read_outputs(two)
    assert sorted(a) == sorted(b) and len(a) == 8
    for fn in a:
        assert a[fn] == b[fn], fn
    strip = lambda t: re.sub(r"\[Gloo\][^\n]*\n", "", t)      # noqa: E731
    assert strip(r2.stdout).replace(two, "{OUT}") == r1.stdout.replace(one, "{OUT}")


def test_more_ranks_than_contigs_is_refused(tmp_path):
    r = run_two_ranks(cli_args("c1_single_bam", str(tmp_path / "out")), 29731)
    assert r.returncode != 0 and "2 GPUs for 1 contig(s)" in r.stderr


@pytest.mark.parametrize("scaling,shared", [("strong", 0.0), ("weak", 0.0), ("weak", 0.02)])
def test_bench_two_ranks_against_the_oracle(scaling, shared, tmp_path):
    """bench.py's N > 1 step with two ranks on one GPU (gloo staging).  strong (what `--gpus N` runs): ONE genome, its contigs
    dealt to the ranks, the join sharded by name hash (gci_route_* + three kinds of all-to-all) -- every contig of every rank
    against the oracle over the undivided files.  weak (round 2): a haplotype per rank, the cross-rank name check (hash
    all-to-all + conflict kernel), the speculative local join or -- with read names shared between the ranks -- the replicated
    join over gathered records + names; in every case the fused build and the integer all-reduce."""
    import json
    env = dict(os.environ, GCI_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    port = 29750 + int(shared * 100) + (7 if scaling == "strong" else 0)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--scale", "0.01", "--backend", "gloo", "--verify-oracle", "--shared-names", str(shared), "--scaling", scaling]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["parity_vs_oracle_all_ranks"] is True and out["scaling"] == scaling
    if scaling == "strong":
        assert "sharded by name hash" in out["config"]["join"]
    else:
        assert ("replicated" in out["config"]["join"]) == (shared > 0)


def test_bench_launches_itself_for_two_gpus():
    """`python bench.py --gpus 2 ...` with NO outer torch.distributed.run (the shape of the driver's command): bench.py becomes
    the launcher, rank 0 generates the workload once into the shared tmpfs directory, rank 1 maps it; one JSON line, n_gpus 2,
    every contig of both ranks equal to the oracle over the undivided files."""
    import json
    env = dict(os.environ, GCI_DIST_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, MASTER_PORT="29793")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--scale", "0.02",
           "--backend", "gloo", "--verify-oracle"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["parity_vs_oracle_all_ranks"] is True
    assert "sharded by name hash" in out["config"]["join"] and out["config"]["workload_generated"] == "once, by rank 0"


# ---- RCCL itself: the branch an 8-GPU node takes, executed on ONE GPU over a world of one ------------------
# Two ranks cannot share a device under RCCL, so everything above stages its collectives through host memory (gloo).  These
# run the nccl branch for real: device tensors straight into all_to_all_single / all_reduce / all_gather_into_tensor on an RCCL
# communicator of world size 1 -- dtype, contiguity, divisibility and stream ordering against the library's kernels are those
# of N = 8.

def _run_one_rank_rccl(script_argv, port, extra_env=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT, **(extra_env or {}))
    env.pop("GCI_DIST_DEVICE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_argv
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)


@pytest.mark.parametrize("mode", ["sharded", "exchange", "replicated"])
def test_bench_step_over_rccl_world_of_one(mode):
    """bench.py's multi-GPU step with --backend nccl: the name-hash-sharded join (2 F + 1 all-to-alls + the all-reduce), the
    round-2 name check (hash all-to-all + all-reduce) and the replicated join (all-gathers) -- each checked against the oracle."""
    import json
    flag = {"sharded": ["--force-sharded"], "exchange": ["--force-exchange"], "replicated": ["--force-exchange", "--force-replicated"]}[mode]
    r = _run_one_rank_rccl([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--scale", "0.01",
                            "--backend", "nccl", "--verify-oracle"] + flag + (["--scaling", "weak"] if mode != "sharded" else []),
                           29770 + len(mode))
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["parity_vs_oracle_all_ranks"] is True
    assert out["config"]["collectives"]["backend"] == "nccl"
    if mode == "sharded":
        assert "sharded by name hash" in out["config"]["join"]
    if mode == "replicated":
        assert "replicated" in out["config"]["join"]


@pytest.mark.parametrize("k,case", list(enumerate(["c3_two_bam", "c5_two_type"])))
def test_command_line_over_rccl_world_of_one(k, case, tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 1 GCI.py ...` with GCI_FORCE_SHARDED=1 and the default backend (nccl =
    RCCL): the contig-sharded command line -- index-driven ingestion, PAF by byte range, ShardedJoin, the gathers to rank 0, the
    all-reduces -- writes the reference's files and transcript."""
    out = str(tmp_path / "out")
    argv = cli_args(case, out)
    r = _run_one_rank_rccl([os.path.join(ROOT, "GCI.py")] + argv[1:], 29780 + k, {"GCI_FORCE_SHARDED": "1", "GCI_DIST_BACKEND": "nccl"})
    assert r.returncode == 0, r.stderr[-3000:]
    got, want = read_outputs(out), expected(case)
    assert sorted(got) == sorted(want)
    for fn in want:
        assert got[fn] == want[fn], fn
    first, _, rest = r.stdout.partition("\n")
    assert first.startswith("Used arguments:{")
    inp = os.path.join(GOLDEN, case, "inputs")
    assert rest.replace(out, "{OUT}").replace(inp, "{IN}") == manifest(case)["stdout"]
