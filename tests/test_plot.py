"""N3 (`-p`): the renderer draws the reference's figure when it is given the reference's numbers.

tests/golden/plot/ holds four figures drawn by the unmodified reference (plot_base, GCI.py:742-834) from the depth
arrays in inputs.npz (tools/make_golden.py plot).  Here the numbers of the figure are restated by the oracle (no GPU),
handed to gci_amd.plot.render and the PNGs compared pixel by pixel.  The GPU side -- figure_spec() producing the same
numbers from tracks in HBM -- is in tests/test_gpu_e2e.py."""
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN

PLOT = os.path.join(GOLDEN, "plot")


def oracle_spec(oracle, plot, depths_list, means, start, end, region, path, title):
    maxd = [m * 4.0 for m in means]
    sl = [{"ctgP": d["ctgP"][start:end]} for d in depths_list]
    averaged, y_frac, y_min, y_max = oracle.pre_plot_base(sl, maxd, 500, start)
    layers = []
    for i, d in enumerate(sl):
        spans = []
        for lo, hi in ((0, means[i] * 0.1), (-1, 0)):
            bed = oracle.collapse_contig(d["ctgP"], lo, hi, 0, start)
            if bed:
                bed = [tuple(x) for x in oracle.merge_merged_depth_bed({"t": bed}, {"t": end - start}, 0.005, start, start, end)["t"]]
            spans.append(bed)
        pos, val = averaged[i]["ctgP"]
        layers.append(plot.Layer(pos, val, means[i], spans[0], spans[1], *((y_frac, 1) if i == 0 else (0, y_frac))))
    return plot.FigureSpec(layers, y_min, y_max, title, path, 0.1), (y_frac, y_min, y_max)


@pytest.mark.parametrize("name", ["one", "two"])
def test_renderer_reproduces_reference_figures(oracle, tmp_path, name):
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.image as mpimg
    from gci_amd import plot
    z = np.load(os.path.join(PLOT, "inputs.npz"))
    meta = json.load(open(os.path.join(PLOT, "meta.json")))[name]
    dl = [{"ctgP": z["hifi"].astype(np.int64)}] + ([{"ctgP": z["nano"].astype(np.int64)}] if name == "two" else [])
    means = [float(np.mean(d["ctgP"])) for d in dl]
    assert means == meta["means"]
    L = int(z["hifi"].shape[0])
    for (start, end, region, fn, title, ykey) in (
            (0, L, False, f"{name}.ctgP.png", "Filtered depth across the whole genome:ctgP", "y"),
            (8000, 12000, True, f"{name}.ctgP:8000-12000.png", "Filtered depth across the region:ctgP:8000-12000", "y_region")):
        out = str(tmp_path / fn)
        spec, ys = oracle_spec(oracle, plot, dl, means, start, end, region, out, title)
        assert list(ys) == meta[ykey]
        plot.render(spec)
        got, want = mpimg.imread(out), mpimg.imread(os.path.join(PLOT, "images", fn))
        assert got.shape == want.shape
        assert np.array_equal(got, want), f"{fn}: {int((got != want).any(axis=-1).sum())} pixels differ"
