"""`GCI.py --gpus N` starts its ranks itself (gci_amd/cli.py: _spawn_ranks): the environment a rank sees, the status the launcher
leaves with, and what it does when a rank fails -- with a stand-in for the command line, off the GPU."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(tmp_path, body, n=3, args=()):
    entry = tmp_path / "rank.py"
    entry.write_text(textwrap.dedent(body))
    code = ("import sys; sys.path.insert(0, %r); from gci_amd import cli; sys.exit(cli._spawn_ranks(%d, %r, %r))" % (ROOT, n, str(entry), list(args)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GCI_LAUNCHER")}
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)


def test_ranks_get_the_rendezvous_environment_and_the_launcher_leaves_with_rank_zeros_status(tmp_path):
    r = _launch(tmp_path, """
        import os, sys
        e = os.environ
        print("rank %s of %s local %s of %s at %s:%s args %s ipc %s launched %s" % (e["RANK"], e["WORLD_SIZE"], e["LOCAL_RANK"], e["LOCAL_WORLD_SIZE"],
              e["MASTER_ADDR"], e["MASTER_PORT"].isdigit(), sys.argv[1:], e["HSA_ENABLE_IPC_MODE_LEGACY"], float(e["GCI_LAUNCHED_AT"]) > 0), flush=True)
        """, n=3, args=["-r", "ref.fa"])
    assert r.returncode == 0, r.stderr
    lines = sorted(r.stdout.splitlines())
    assert lines == ["rank %d of 3 local %d of 3 at 127.0.0.1:True args ['-r', 'ref.fa'] ipc 0 launched True" % (k, k) for k in range(3)]


def test_a_failing_rank_ends_the_others_and_the_launcher_says_so(tmp_path):
    # rank 1 dies at once, the others would wait for ever (a collective that never completes)
    r = _launch(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(600)
        """, n=3)
    assert r.returncode == 7
    # every rank fails alike (a bad argument): rank 0's own status, and it had the time to say why
    r = _launch(tmp_path, """
        import os, sys
        sys.exit("ERROR!!! said by rank %s" % os.environ["RANK"] if os.environ["RANK"] == "0" else 1)
        """, n=2)
    assert r.returncode == 1 and "said by rank 0" in r.stderr
