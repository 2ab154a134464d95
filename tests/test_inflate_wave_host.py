"""The next inflate (a wave per BGZF member, tools/hwtests/inflate_wave.hip -- an experiment outside libgci_hip.so) as far as it can
be held to account without a GPU: its decoding helpers (inflate_wave_core.hpp) compiled for the host and the kernel's phases played
lane by lane (inflate_wave_host_check.cpp) over a small BAM -- every member the scheme decodes equals zlib's output, the rest is
handed back with a reason."""
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

from gci_amd import synth
from gci_amd.formats import bam as bamfmt
from gci_amd.formats import bgzf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
def test_wave_scheme_on_the_host_equals_zlib(tmp_path):
    exe = str(tmp_path / "iw_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(ROOT, "tools", "hwtests", "inflate_wave_host_check.cpp"), "-lz"])
    rs = synth.simulate_reads((("chr19", 400_000),), 40, "hifi", seed=synth.seed_for(2, 0))
    stream, _ = synth.to_bam_stream(rs, seq_qual="random", seed=7)
    p = str(tmp_path / "x.bam")
    bamfmt.write_bam_stream(p, stream, level=1, threads=2)
    # members of other kinds behind it: stored blocks (level 0), fixed-code blocks (tiny payloads), an empty member
    with open(p, "ab") as f:
        for payload, level in ((bytes(np.random.default_rng(1).integers(0, 256, 40000, dtype=np.uint8)), 0), (b"ACGT" * 9, 9), (b"", 6)):
            f.write(bgzf._member(payload, level))
    r = subprocess.run([exe, p], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    m = re.match(r"(\d+) members: (\d+) decoded by the wave scheme and equal to zlib byte for byte; by status: ok (\d+),", r.stdout)
    assert m, r.stdout
    members, same, ok = (int(x) for x in m.groups())
    assert members > 300 and same == ok and ok >= members * 0.98, r.stdout
