"""gci_amd: an MI355X-native implementation of GCI's alignment-filter -> per-base-depth ->
issue-scan hot path.  Host code is Python; all per-record and per-base work
runs in hand-written gfx950 HIP kernels behind the C-ABI declared in include/gci_hip.h."""

__version__ = "0.1.0"
