"""The host-side entry points of the C ABI (gci_amd/csrc/host_io.cpp: threaded BGZF inflate, the heads pipeline, gzip framing,
the host PAF filter, FASTA titles -- code the command line executes) built by g++ with AddressSanitizer + UBSan
(tools/asan_host.sh), and tests/test_host_logic.py run over that build in a child process.  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_entry_points_under_address_and_ub_sanitizers():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    asan = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan in this toolchain")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "asan_host.sh")], capture_output=True, text=True, timeout=900,
                       env={k: v for k, v in os.environ.items() if k not in ("GCI_LIB_PATH", "GCI_HOST_ONLY")})
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout and "failed" not in r.stdout and "ERROR: AddressSanitizer" not in r.stderr
