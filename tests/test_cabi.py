"""The C-ABI boundary without a GPU: the library loads, exports every symbol include/gci_hip.h
declares, and its host-only entry points behave.  (No compute calls here: those are -m gpu.)"""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gci_amd import build, _lib
    build.build_hip()
    return _lib.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from gci_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gci_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gci_[a-z0-9_]+)\s*\(", hdr))
    bound = {name for name, _, _ in _lib.EXPORTS}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layouts_match_header():
    from gci_amd import _lib
    from gci_amd.device import REC_DTYPE, IVL_DTYPE
    assert REC_DTYPE.itemsize == 32 and IVL_DTYPE.itemsize == 16
    assert [REC_DTYPE.fields[f][1] for f in ("name_hash", "contig", "start", "end", "qlen", "rec_idx", "mapq", "flags",
                                             "name_len")] == [0, 8, 12, 16, 20, 24, 28, 29, 30]
    assert ctypes.sizeof(_lib.JoinFile) == 32 and ctypes.sizeof(_lib.Window) == 16


def test_status_strings_and_decode(lib):
    from gci_amd import _lib
    assert lib.gci_abi_version() == 1
    assert lib.gci_strerror(0) == b"ok"
    assert b"KeyError" in lib.gci_strerror(_lib.GCI_E_NO_NM)
    rec = ctypes.c_uint32(0)
    assert lib.gci_decode_status((1 << 64) - 1, ctypes.byref(rec)) == 0
    assert lib.gci_decode_status((1234 << 8) | 4, ctypes.byref(rec)) == _lib.GCI_E_ZERO_DIV and rec.value == 1234


def test_name_hash_host_twins_agree(lib):
    from gci_amd import _lib
    from gci_amd.device import name_hash_np
    rng = np.random.default_rng(1)
    names = [b"", b"a", b"12345678", b"123456789", b"m64011_190830_220126/4194370/ccs"]
    names += [bytes(rng.integers(33, 127, int(rng.integers(1, 255))).astype(np.uint8)) for _ in range(200)]
    h = name_hash_np(names)
    for n, v in zip(names, h.tolist()):
        assert _lib.name_hash(n) == v
    assert len(set(h.tolist())) == len(set(names))


def test_no_gpu_means_loud_failure(lib):
    """The product path never falls back to the CPU: without a device, creating an Engine raises."""
    import torch
    from gci_amd.device import Engine
    from gci_amd._lib import GciError
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(GciError):
        Engine(0)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "gci_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, fn)).read()
                assert "import oracle" not in src and "from oracle" not in src and "gci_oracle" not in src, fn
    for fn in ("GCI.py",):
        assert "oracle" not in open(os.path.join(ROOT, fn)).read()
