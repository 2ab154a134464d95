"""gci_amd: an MI355X-native implementation of GCI's alignment-filter -> per-base-depth ->
issue-scan hot path.  Host code is Python; all per-record and per-base work
runs in hand-written gfx950 HIP kernels behind the C-ABI declared in include/gci_hip.h."""

__version__ = "0.1.0"

import os as _os

# HIP maps a process's streams onto FOUR hardware queues unless told otherwise, and streams that share a queue run one after the
# other: with the copy stream of the uploads, the main stream, the inflate's second stream and the assembly's side stream alive at
# once, the DMA of a run's bytes sat behind the inflate kernels of the run in front (uploads at 26 GB/s instead of 58: round 5,
# tools/hwtests/cli_trace.sh).  Read by the runtime when it starts, so it is set here, before anything touches the device.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
