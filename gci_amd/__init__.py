"""gci_amd: an MI355X-native implementation of GCI's alignment-filter -> per-base-depth ->
issue-scan hot path.  Host code is Python; all per-record and per-base work
runs in hand-written gfx950 HIP kernels behind the C-ABI declared in include/gci_hip.h.

IMPORT SIDE EFFECT: importing this package puts GPU_MAX_HW_QUEUES=8 into the process environment when the variable is absent and
no HIP runtime has started yet (below).  `HW_QUEUES_OK` says whether the setting can have taken effect; only then does an Engine
ask the library for the inflate's second stream (gci_bgzf_inflate_streams)."""

__version__ = "0.1.0"

import os as _os
import sys as _sys


def _hip_runtime_started() -> bool:
    """Has a HIP runtime of this process been initialised already?  torch knows for the runtime it loaded; without torch, a
    libamdhip64 somebody else mapped may or may not have been initialised -- taken as started."""
    t = _sys.modules.get("torch")
    if t is not None:
        try:
            return bool(t.cuda.is_initialized())
        except Exception:                              # noqa: BLE001
            return True
    try:
        with open("/proc/self/maps") as f:
            return any("libamdhip64" in line for line in f)
    except OSError:
        return False


# HIP maps a process's streams onto FOUR hardware queues unless told otherwise, and streams that share a queue run one after the
# other: with the copy stream of the uploads, the main stream, the inflate's second stream and the assembly's side stream alive at
# once, the DMA of a run's bytes sat behind the inflate kernels of the run in front (uploads at 26 GB/s instead of 58: round 5,
# tools/hwtests/cli_trace.sh).  The runtime reads the variable when it starts: it is set here only when that can still matter, and
# a value that was in the environment before this import (the user's, GCI.py's, bench.py's: they set it before anything touches
# the device) is taken at its word.
_pre = _os.environ.get("GPU_MAX_HW_QUEUES")
if _pre is None:
    HW_QUEUES_OK = not _hip_runtime_started()
    if HW_QUEUES_OK:
        _os.environ["GPU_MAX_HW_QUEUES"] = "8"
else:
    try:
        HW_QUEUES_OK = int(_pre) >= 8
    except ValueError:
        HW_QUEUES_OK = False
