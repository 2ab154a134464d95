#!/usr/bin/env python3
"""Drop-in for the reference's `python GCI.py ...` command line (same flags, same outputs);
the work is done by the gfx950 HIP path in gci_amd/ -- see gci_amd/cli.py.

A single-GPU run imports no tensor library: its HBM buffers, streams and events are the library's own (gci_amd/hbm.py over the
gci_dev_* exports of libgci_hip.so).  torch is imported only by the ranks of a `--gpus N` run (torch.distributed over RCCL)."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # (gci_amd/__init__.py says why; here as well: the runtime is woken below, before that import)


def _wake_the_gpu():
    """The HIP runtime's own start (driver, device, primary context: a few tenths of a second) on a thread of its own while the
    interpreter imports numpy and the rest of the package -- through libgci_hip.so itself, the library every later call goes through
    (ctypes releases the interpreter lock for the duration of a call).  Anything that goes wrong here is left for the ordinary path to
    report."""
    try:
        import ctypes
        from gci_amd import _lib
        lib = _lib.load()
        n = ctypes.c_int(0)
        if lib.gci_dev_count(ctypes.byref(n)) == 0 and n.value > 0:
            lib.gci_dev_mem_info(0, None, None)       # (the device's primary context)
    except Exception:                                 # noqa: BLE001
        pass


if __name__ == "__main__" and os.environ.get("GCI_STUCK_TRACE"):
    # measurements: every N seconds the stacks of all threads on stderr (a process that sits somewhere says where)
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["GCI_STUCK_TRACE"]), repeat=True)


def _a_single_gpu_run(argv) -> bool:
    """True for a command line that will do its work HERE on one GPU: not --help / --version / no arguments (they end in
    argparse), not the launcher of a multi-GPU run (it starts the ranks and waits), not one of those ranks (torch brings its own
    copy of the runtime and must be the one to load it)."""
    if len(argv) < 2 or any(a in ("-h", "--help", "-v", "--version") for a in argv[1:]):
        return False
    if all(k in os.environ for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR")) or os.environ.get("GCI_HBM") == "torch":
        return False
    for k, a in enumerate(argv[1:], 1):
        if a == "--gpus" or a.startswith("--gpus="):
            n = a.split("=", 1)[1] if "=" in a else (argv[k + 1] if k + 1 < len(argv) else "1")
            if n.isdigit() and int(n) > 1:
                return False
    return True


_WAKER = None
if __name__ == "__main__" and _a_single_gpu_run(sys.argv) and os.environ.get("GCI_EARLY_HIP", "1") != "0":
    import threading
    _WAKER = threading.Thread(target=_wake_the_gpu, daemon=True)
    _WAKER.start()

from gci_amd.cli import main  # noqa: E402

def _exit_trace():
    """GCI_EXIT_TRACE=<path of tools/hwtests/exit_stamp.c built as a shared library> (tools/hwtests/exit_cost.py): wall-clock stamps on
    stderr when main() has returned, when the interpreter's atexit handlers run and -- from that library, loaded here, i.e. behind the
    HIP runtime -- when the C library's handlers start: what the process does between its last output file and its end."""
    import atexit
    import ctypes
    import time
    sys.stderr.write("exit-trace main_returned %.6f\n" % time.time())
    atexit.register(lambda: sys.stderr.write("exit-trace python_atexit %.6f\n" % time.time()))
    ctypes.CDLL(os.environ["GCI_EXIT_TRACE"])


if __name__ == "__main__":
    try:
        main(sys.argv)
        if os.environ.get("GCI_EXIT_TRACE"):
            _exit_trace()
    finally:
        if _WAKER is not None:
            _WAKER.join(timeout=10.0)                 # (an early exit -- a refused argument -- does not leave while the runtime is starting)
