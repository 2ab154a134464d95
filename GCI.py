#!/usr/bin/env python3
"""Drop-in for the reference's `python GCI.py ...` command line (same flags, same outputs);
the work is done by the gfx950 HIP path in gci_amd/ -- see gci_amd/cli.py."""
import os
import sys


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

if __name__ == "__main__" and len(sys.argv) > 1 and os.environ.get("GCI_EARLY_HIP", "1") != "0":
    import threading
    threading.Thread(target=_wake_the_gpu, daemon=True).start()

from gci_amd.cli import main  # noqa: E402

if __name__ == "__main__":
    main(sys.argv)
