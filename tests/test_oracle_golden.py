"""Pins the oracle: its whole-path restatement must reproduce, byte for byte, what the unmodified
reference wrote for every golden case (tests/golden/*, made by tools/make_golden.py), and the
reference's own example/MH63 triple."""
import gzip
import json
import os

import numpy as np
import pytest

from golden_util import CASES, GOLDEN, expected, inputs, manifest
from gci_amd.formats import bam as bamfmt
from gci_amd.formats import fasta


def _kind(case, names):
    if not names:
        return None
    d = {"paf": [], "bam": []}
    for p in inputs(case, names):
        if p.endswith(".bam"):
            stream, hdr, offs = bamfmt.read_bam(p, threads=2)
            d["bam"].append((stream, offs, list(hdr.references)))
        else:
            d["paf"].append(p)
    return d


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_outputs(oracle, case):
    m = manifest(case)
    ref_fa = os.path.join(GOLDEN, case, "inputs", "ref.fa")
    _, ns_bed = fasta.n_runs(ref_fa)
    first_bam = [p for p in inputs(case, (m["hifi"] or m["nano"])) if p.endswith(".bam")][0]
    hdr = bamfmt.read_header(first_bam)
    regions = {}
    if m["regions"]:
        for line in open(os.path.join(GOLDEN, case, "inputs", "regions.bed")):
            t, s, e = line.strip().split("\t")
            regions.setdefault(t, []).append((int(s), int(e)))
    a = dict(m["args"])
    chrs = a.pop("chrs", None)
    for k in ("plot", "window_size", "depth_min", "depth_max"):      # -p: the figures are checked in tests/test_plot.py
        a.pop(k, None)
    got = oracle.run_path(hifi=_kind(case, m["hifi"]), nano=_kind(case, m["nano"]), references=hdr.references,
                          lengths=hdr.lengths, ns_bed=ns_bed or None, chrs_list=chrs.split(",") if chrs else (),
                          regions_bed=regions, **a)
    want = expected(case)
    assert sorted(got) == sorted(want)
    for fn in want:
        assert got[fn] == want[fn], fn


def test_oracle_reproduces_mh63_example(oracle):
    """The reference's own example: MH63.depth.gz -> MH63.0.depth.bed + MH63.gci."""
    d = os.path.join(GOLDEN, "MH63")
    text = gzip.open(os.path.join(d, "MH63.depth.gz"), "rb").read()
    depths = oracle.parse_depth_text(text)
    assert len(depths) == 12 and sum(v.shape[0] for v in depths.values()) == 395_765_488
    merged = oracle.collapse_depth_range(depths, -1, 0, 15, 0)
    assert oracle.bed_text(merged) == open(os.path.join(d, "MH63.0.depth.bed")).read()
    tl = {k: int(v.shape[0]) for k, v in depths.items()}
    gci, _ = oracle.compute_index_text(tl, [merged], ["HiFi"])
    assert gci == open(os.path.join(d, "MH63.gci")).read()
    assert oracle.depth_text(depths) == text


def test_oracle_kats(oracle):
    k = json.load(open(os.path.join(GOLDEN, "kats.json")))
    for c in k["collapse_depth_range"]:
        want = [tuple(x) for x in c["out"]]
        d = np.array(c["depth"], dtype=np.int64)
        assert oracle.collapse_contig(d, c["lo"], c["hi"], c["fl"], c["sp"]) == want
        assert oracle.collapse_contig_py(c["depth"], c["lo"], c["hi"], c["fl"], c["sp"]) == want
    for c in k["interval_algebra"]:
        segs = [tuple(x) for x in c["segs"]]
        assert oracle.complement_merged_depth({"t": segs}, {"t": c["L"]}, c["fl"], c["start"], c["end"])["t"] == c["complement"]
        assert [list(x) for x in oracle.merge_merged_depth_bed({"t": segs}, {"t": c["L"]}, c["dp"], c["fl"], c["start"],
                                                               c["end"])["t"]] == c["merged"]
    for c in k["compute_n50"]:
        assert oracle.compute_n50(c["lengths"]) == c["out"]
    for c in k["slice_add"]:
        d = oracle.depth_build({"q": ("t", c["s"], c["e"])}, {"t": c["L"]}, c["fl"])["t"]
        assert d.tolist() == c["out"]
        assert oracle.depth_build_py([("t", c["s"], c["e"])], {"t": c["L"]}, c["fl"])["t"].tolist() == c["out"]
    for c in k["merge_alns_properties"]:
        alns = [tuple(a) for a in c["alns"]]
        assert list(oracle._merge_blocks(alns, 1, 2)) == c["q"]
        assert list(oracle._merge_blocks(alns, 3, 4)) == c["t"]
    for c in k["score_repr"]:
        assert repr(oracle._score(c["obs_n50"], c["exp_n50"], c["obs_n"], c["exp_n"])) == c["out"]
    # N3: the -p numeric front-end
    for c in k["sliding_window_average_depth"]:
        pos, val = oracle.sliding_window_average_depth(c["depth"], c["ws"], c["max_depth"], c["start"])
        assert pos == c["pos"] and val.tolist() == c["val"]
    for c in k["pre_plot_base"]:
        dl = [{t: np.array(v) for t, v in d.items()} for d in c["depths"]]
        av, y_frac, y_min, y_max = oracle.pre_plot_base(dl, c["max_depths"], c["ws"], 0)
        assert (y_frac, y_min, y_max) == (c["y_frac"], c["y_min"], c["y_max"])
        for a, want in zip(av, c["series"]):
            for t, (p, v) in a.items():
                assert p == want[t][0] and v.tolist() == want[t][1]
