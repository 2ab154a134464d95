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


def test_the_quick_exit_is_not_taken_under_a_profiler_or_as_a_rank(monkeypatch):
    """GCI.py leaves through os._exit after a run that went through (_leave_at_once): not when a tool collects at exit, not as one
    rank of several, not when GCI_EXIT=clean says so."""
    import importlib.util, os, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gci_launcher_under_test", os.path.join(root, "GCI.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                              # (not __main__: nothing runs)
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_", "HSA_TOOLS_LIB", "COVERAGE_", "COV_CORE_", "PYTHONFAULTHANDLER")) or k in ("GCI_EXIT", "WORLD_SIZE", "LD_PRELOAD"):
            monkeypatch.delenv(k, raising=False)
    assert mod._leave_at_once() is (sys.gettrace() is None)
    monkeypatch.setenv("GCI_EXIT", "clean")
    assert mod._leave_at_once() is False
    monkeypatch.delenv("GCI_EXIT")
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert mod._leave_at_once() is False
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("ROCPROFILER_REGISTER_ROOT", "/opt/rocm")
    assert mod._leave_at_once() is False
    monkeypatch.delenv("ROCPROFILER_REGISTER_ROOT")
    monkeypatch.setenv("LD_PRELOAD", "/opt/rocm/lib/librocprofiler-sdk-tool.so")
    assert mod._leave_at_once() is False
