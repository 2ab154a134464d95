"""gci_amd/hbm.py off the GPU: the size classes of its allocator, the view arithmetic of its buffers (no device call is made for
a slice, a row, a reinterpretation or a reshape) and the provider rule."""
import sys

import numpy as np
import pytest

from gci_amd import hbm


def test_size_classes():
    assert hbm._round_size(0) == 512 and hbm._round_size(1) == 512 and hbm._round_size(513) == 1024
    for n in (1 << 20, (1 << 20) + 1, 3_240_000_000, 7_700_000_001, 12_468_000_000, (1 << 34) - 5):
        r = hbm._round_size(n)
        assert n <= r <= n + n // 16 + 1 and hbm._round_size(r) == r
    # a run that is a few MB smaller than the one before it lands in the class the bigger one left behind more often than not
    assert hbm._round_size(4_294_000_000) == hbm._round_size(4_290_000_000)


def test_views_are_pointer_arithmetic():
    dev = hbm.Device(0)
    b = hbm.Buf(None, 0x1000, (10, 32), hbm.uint8, dev)
    assert b.nbytes == 320 and b[2:5].data_ptr() == 0x1000 + 64 and b[2:5].shape == (3, 32) and b[7].shape == (32,)
    assert b[-1].data_ptr() == 0x1000 + 9 * 32 and b[4:2].shape == (0, 32) and b[:100].shape == (10, 32)
    v = b.view(hbm.int64)
    assert v.shape == (10, 4) and v.dtype is hbm.int64 and v.data_ptr() == 0x1000
    assert b.reshape(-1).shape == (320,) and b.reshape(5, -1).shape == (5, 64) and b.contiguous() is b
    with pytest.raises(ValueError):
        hbm.Buf(None, 0, (3,), hbm.uint8, dev).view(hbm.int32)
    with pytest.raises(NotImplementedError):
        b[:, 29]
    with pytest.raises(NotImplementedError):
        b[::2]
    with pytest.raises(IndexError):
        b[10]
    assert hbm.Buf(None, 0, (0, 4), hbm.int32, dev).numel() == 0 and len(b) == 10


def test_provider_rule(monkeypatch):
    monkeypatch.delenv("GCI_HBM", raising=False)
    assert hbm.provider("native") is hbm.native() and hbm.native().name == "native"
    monkeypatch.setenv("GCI_HBM", "native")
    assert hbm.provider() is hbm.native()
    monkeypatch.setenv("GCI_HBM", "something")
    with pytest.raises(ValueError):
        hbm.provider()
    monkeypatch.delenv("GCI_HBM")
    if "torch" in sys.modules:
        assert hbm.provider().name == "torch"
