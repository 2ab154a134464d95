#!/usr/bin/env python3
"""Drop-in for the reference's `python GCI.py ...` command line (same flags, same outputs);
the work is done by the gfx950 HIP path in gci_amd/ -- see gci_amd/cli.py."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (gci_amd/__init__.py says why; here as well: the runtime is woken below, before that import)


def _wake_the_gpu():
    """The HIP runtime's own start (driver, device, primary context: a few tenths of a second without the interpreter) on a
    thread of its own while the interpreter imports torch -- through the very library file torch will load, so that the process
    holds one runtime.  Anything that goes wrong here is left for the ordinary path to report."""
    try:
        import ctypes
        import importlib.util
        spec = importlib.util.find_spec("torch")
        lib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if not os.path.isfile(lib):
            return
        hip = ctypes.CDLL(lib, mode=ctypes.RTLD_GLOBAL)
        if hip.hipInit(0) == 0 and hip.hipSetDevice(int(os.environ.get("LOCAL_RANK", "0") or 0)) == 0:
            hip.hipFree(None)
    except Exception:                                 # noqa: BLE001
        pass


if __name__ == "__main__" and os.environ.get("GCI_STUCK_TRACE"):
    # measurements: every N seconds the stacks of all threads on stderr (a process that sits somewhere says where)
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["GCI_STUCK_TRACE"]), repeat=True)

def _a_run_of_this_process(argv) -> bool:
    """True for a command line that will do its work HERE: not --help / --version / no arguments (they end in argparse), not the
    launcher of a multi-GPU run (it execs torch.distributed.run right away: a runtime that is starting on another thread would be
    torn down under it)."""
    if len(argv) < 2 or any(a in ("-h", "--help", "-v", "--version") for a in argv[1:]):
        return False
    launched = all(k in os.environ for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR"))
    for k, a in enumerate(argv[1:], 1):
        if a == "--gpus" or a.startswith("--gpus="):
            n = a.split("=", 1)[1] if "=" in a else (argv[k + 1] if k + 1 < len(argv) else "1")
            if n.isdigit() and int(n) > 1 and not launched:
                return False
    return True


_WAKER = None
if __name__ == "__main__" and _a_run_of_this_process(sys.argv) and os.environ.get("GCI_EARLY_HIP", "1") != "0":
    import threading
    _WAKER = threading.Thread(target=_wake_the_gpu, daemon=True)
    _WAKER.start()

from gci_amd.cli import main  # noqa: E402

def _leave_at_once() -> bool:
    """After a run that went through -- every output file written and closed -- the process leaves through os._exit: what an
    orderly interpreter exit does from there on (the allocator handing 150 GB of device memory back piece by piece, pinned
    slots unmapped, the HIP runtime and a hundred modules taken down) took 0.3 - 0.4 s of a 6 s command line and changes
    nothing a caller can see.  Not under a profiler or a tool that collects at exit (they write their results in atexit /
    library destructors), not as one rank of several (the process group is taken down in order), and GCI_EXIT=clean turns
    it off."""
    if os.environ.get("GCI_EXIT", "") == "clean":
        return False
    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
        return False
    tools = ("ROCPROFILER_", "ROCPROF", "ROCP_", "HSA_TOOLS_LIB", "COVERAGE_", "COV_CORE_", "PYTHONFAULTHANDLER")
    if any(k.startswith(t) for k in os.environ for t in tools):
        return False
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or sys.gettrace() is not None:
        return False
    return True


if __name__ == "__main__":
    done = False
    try:
        main(sys.argv)
        done = True
    finally:
        if _WAKER is not None:
            _WAKER.join(timeout=10.0)                 # (an early exit -- a refused argument -- does not leave while the runtime is starting)
    if done and _leave_at_once():
        try:
            import threading as _th
            for t in _th.enumerate():                 # what an interpreter exit waits for as well
                if t is not _th.main_thread() and not t.daemon:
                    t.join()
            sys.stdout.flush()
            sys.stderr.flush()
        finally:
            os._exit(0)
