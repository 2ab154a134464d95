"""R14: the command line's refusals, warnings and exit behaviour against transcripts of the UNMODIFIED reference run
through its own `__main__` block (tests/golden/cli_errors.json, tools/make_golden.py::case_cli_errors): exit message or
code, stdout and stderr of every scenario.  Everything that is refused before the first GPU call runs on the CPU; the
scenarios that get as far as the pipeline are marked gpu."""
import io
import json
import os
import contextlib

import pytest

from golden_util import GOLDEN

SCENARIOS = json.load(open(os.path.join(GOLDEN, "cli_errors.json")))
INPUTS = os.path.join(GOLDEN, "cli_errors", "inputs")
NEEDS_GPU = {"hifi_nano_lengths_differ", "mapq_warning_then_runs", "refuses_to_overwrite_depth", "refuses_to_overwrite_gaps", "refuses_to_overwrite_gaps_again"}


def run_scenario(sc, out_root, monkeypatch):
    from gci_amd import cli
    monkeypatch.setenv("COLUMNS", "100")
    sub = lambda t: t.replace("{IN}", INPUTS).replace("{OUT}", out_root)       # noqa: E731
    norm = lambda t: t.replace(out_root, "{OUT}").replace(INPUTS, "{IN}")      # noqa: E731
    so, se = io.StringIO(), io.StringIO()
    code = "completed"
    try:
        with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
            cli.main(["GCI.py"] + [sub(a) for a in sc["argv"]])
    except SystemExit as e:
        code = e.code
    got = {"exit": norm(code) if isinstance(code, str) else code, "stdout": norm(so.getvalue()), "stderr": norm(se.getvalue())}
    want = {k: sc[k] for k in ("exit", "stdout", "stderr")}
    assert got == want, sc["name"]


@pytest.mark.parametrize("sc", [s for s in SCENARIOS if s["name"] not in NEEDS_GPU], ids=lambda s: s["name"])
def test_refused_before_any_gpu_work(sc, tmp_path, monkeypatch):
    run_scenario(sc, str(tmp_path / "out"), monkeypatch)
    if "-d" in sc["argv"] and sc["name"] not in ("regions_missing",):
        # the reference creates the output directory before it looks at the prefix / the reference (GCI.py:914-923)
        assert os.path.isdir(str(tmp_path / "out"))


@pytest.mark.gpu
def test_warning_and_overwrite_guards(engine, tmp_path, monkeypatch):
    from gci_amd import pipeline
    pipeline._ENGINE = engine
    ran = 0
    for sc in SCENARIOS:                       # in file order: the overwrite scenarios re-use the directory of the run before
        if sc["name"] in NEEDS_GPU:            # (hifi_nano_lengths_differ is refused after the gap scan, which runs on the GPU)
            run_scenario(sc, str(tmp_path / "out"), monkeypatch)
            ran += 1
    assert ran == len(NEEDS_GPU)


def test_the_launcher_has_no_quick_exit_and_decides_who_wakes_the_gpu(monkeypatch):
    """GCI.py ends through the interpreter's ordinary exit (round 5 left through os._exit to hide torch's teardown: with the
    single-GPU run holding its buffers itself there is nothing to hide), and it starts the HIP runtime early only for a run that
    does its work in this process on one GPU -- not for --help / --version, not as the launcher or a rank of a --gpus N run, not
    when GCI_HBM=torch hands the runtime's start to torch."""
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "GCI.py")).read()
    assert "os._exit" not in text.replace("left through os._exit", "") and "import torch" not in text.replace("`import torch`", "").replace("no `import torch`", "")
    spec = importlib.util.spec_from_file_location("gci_launcher_under_test", os.path.join(root, "GCI.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                              # (not __main__: nothing runs)
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "GCI_HBM"):
        monkeypatch.delenv(k, raising=False)
    run = ["GCI.py", "-r", "ref.fa", "--hifi", "a.bam"]
    assert mod._a_single_gpu_run(run) is True and mod._a_single_gpu_run(run + ["--gpus", "1"]) is True
    assert mod._a_single_gpu_run(["GCI.py"]) is False and mod._a_single_gpu_run(["GCI.py", "-v"]) is False and mod._a_single_gpu_run(run + ["-h"]) is False
    assert mod._a_single_gpu_run(run + ["--gpus", "8"]) is False and mod._a_single_gpu_run(run + ["--gpus=2"]) is False
    monkeypatch.setenv("GCI_HBM", "torch")
    assert mod._a_single_gpu_run(run) is False
    monkeypatch.delenv("GCI_HBM")
    for k, v in (("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_ADDR", "127.0.0.1")):
        monkeypatch.setenv(k, v)
    assert mod._a_single_gpu_run(run) is False
