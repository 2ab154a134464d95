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


def test_struct_layouts_agree_with_the_header_as_a_c_compiler_sees_it(tmp_path):
    """The ctypes mirrors of the header's structs (gci_amd/_lib.py) field by field against offsetof / sizeof from gcc over
    include/gci_hip.h: a field added on one side only (gci_build_opts.want_runs took what was padding) shows here, off the GPU."""
    import shutil, subprocess
    from gci_amd import _lib
    from gci_amd.device import REC_DTYPE, IVL_DTYPE
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {
        "gci_join_file": (_lib.JoinFile, [f for f, _ in _lib.JoinFile._fields_]),
        "gci_window": (_lib.Window, [f for f, _ in _lib.Window._fields_]),
        "gci_build_opts": (_lib.BuildOpts, [f for f, _ in _lib.BuildOpts._fields_]),
    }
    dtypes = {"gci_rec": (REC_DTYPE, ["name_hash", "contig", "start", "end", "qlen", "rec_idx", "mapq", "flags", "name_len"]),
              "gci_ivl": (IVL_DTYPE, ["contig", "start", "end"])}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "gci_hip.h"', 'int main(void) {']
    for name, (_, fields) in list(structs.items()) + list(dtypes.items()):
        lines.append('printf("%s size %%zu\\n", sizeof(%s));' % (name, name))
        for f in fields:
            lines.append('printf("%s %s %%zu\\n", offsetof(%s, %s));' % (name, f, name, f))
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run([cc, "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    got = {}
    for ln in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n"):
        if ln:
            a, b, c = ln.split()
            got[(a, b)] = int(c)
    for name, (cls, fields) in structs.items():
        assert ctypes.sizeof(cls) == got[(name, "size")], name
        for f in fields:
            assert getattr(cls, f).offset == got[(name, f)], (name, f)
    for name, (dt, fields) in dtypes.items():
        assert dt.itemsize == got[(name, "size")], name
        for f in fields:
            assert dt.fields[f][1] == got[(name, f)], (name, f)
