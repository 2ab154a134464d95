"""`-p/--plot`: the depth figures of the reference (plot_depth / plot_base, /root/reference/GCI.py:742-895).

Split in two so that the numbers can be checked without looking at pixels:

  figure_spec()   everything a figure shows, as plain data -- the windowed mean series (pipeline.pre_plot_base: zero
                  runs + window sums on the GPU), the shaded low-depth / zero-depth spans (issue scan on the GPU +
                  the host's interval merge), the mean lines and the axis limits;
  render()        draws one FigureSpec with matplotlib (host only; imported lazily so that nothing else needs it).

plot_depth() is the reference's driver: image type check, mean depths, overwrite checks, one figure per contig and
one per region of the regions file.
"""
from __future__ import annotations

import os
import sys
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import pipeline, score

COLOR_TYPE = ("#2ca25f", "#3C5488")        # first read type (HiFi) above the axis, second (Nano) below
COLOR_LOW, COLOR_ZERO = "#B7DBEA", "#FAD7DD"
LABEL_TYPE = ("HiFi", "Nano")


@dataclass
class Layer:
    """One read type in one figure."""
    positions: List[float]                  # Mb
    values: np.ndarray                      # windowed mean depth (clamped)
    mean: float
    low_spans: List[Tuple[int, int]]        # threshold < depth <= mean * depth_min, merged (bp)
    zero_spans: List[Tuple[int, int]]       # depth <= threshold, merged (bp)
    y_from: float                           # vertical extent of the spans in axes fractions
    y_to: float


@dataclass
class FigureSpec:
    layers: List[Layer]
    y_min: float
    y_max: float
    title: str
    path: str
    depth_min: float = 0.1
    extra: Dict[str, object] = field(default_factory=dict)


def _spans(tracks: pipeline.DepthTracks, target: str, start: int, end: int, lo: float, hi: float, dist_percent: float
           ) -> List[Tuple[int, int]]:
    """collapse_depth_range({target: depths[start:end]}, lo, hi, 0, start) followed -- when anything was found -- by
    merge_merged_depth_bed(..., {target: end - start}, dist_percent, start, start, end)   (GCI.py:785-797; the
    reference passes `start` for flank_len there)."""
    bed = pipeline.collapse_regions(tracks, [(target, start, end)], lo, hi)[0]
    if not bed:
        return []
    return [tuple(x) for x in score.merge_merged_depth_bed({target: bed}, {target: end - start}, dist_percent, start, start, end)[target]]


def figure_spec(depths_list: Sequence[pipeline.DepthTracks], target: str, averaged_dicts, mean_depths: Sequence[float],
                y_frac: float, start: int, depth_min: float, dist_percent: float, y_min: float, y_max: float, image_type: str,
                directory: str, prefix: str, end: int, regions_flag: bool, threshold) -> FigureSpec:
    layers = []
    for i, tracks in enumerate(depths_list):
        y_from, y_to = (y_frac, 1) if i == 0 else (0, y_frac)
        pos, val = averaged_dicts[i][target]
        layers.append(Layer(pos, val, mean_depths[i],
                            _spans(tracks, target, start, end, threshold, mean_depths[i] * depth_min, dist_percent),
                            _spans(tracks, target, start, end, -1, threshold, dist_percent), y_from, y_to))
    if not regions_flag:
        title, path = f"Filtered depth across the whole genome:{target}", f"{directory}/images/{prefix}.{target}.{image_type}"
    else:
        title = f"Filtered depth across the region:{target}:{start}-{end}"
        path = f"{directory}/images/{prefix}.{target}:{start}-{end}.{image_type}"
    return FigureSpec(layers, y_min, y_max, title, path, depth_min)


def render(spec: FigureSpec) -> None:
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.lines as mlines
    import matplotlib.pyplot as plt
    from matplotlib.ticker import AutoMinorLocator

    two = len(spec.layers) == 2
    fig, ax = plt.subplots(figsize=(20, 8 if two else 4))
    if two:
        ax.axhline(0, color="black")
        handles = [mlines.Line2D([], [], color=c, label=l, lw=0.8) for c, l in zip(COLOR_TYPE, LABEL_TYPE)]
        ax.add_artist(plt.legend(handles=handles, loc="upper left"))
    any_low = any_zero = False
    for i, layer in enumerate(spec.layers):
        sign = 1 if i == 0 else -1
        for a, b in layer.low_spans:
            ax.axvspan(a / 1e6, b / 1e6, layer.y_from, layer.y_to, facecolor=COLOR_LOW)
        for a, b in layer.zero_spans:
            ax.axvspan(a / 1e6, b / 1e6, layer.y_from, layer.y_to, facecolor=COLOR_ZERO)
        any_low |= bool(layer.low_spans)
        any_zero |= bool(layer.zero_spans)
        ax.stackplot(layer.positions, sign * layer.values, lw=0.8, color=COLOR_TYPE[i], zorder=4)
        ax.axhline(sign * layer.mean, color="r", ls="-.", dash_capstyle="butt", lw=1, zorder=5)
    ax.set_ylim(bottom=-spec.y_min, top=spec.y_max)
    ax.xaxis.set_minor_locator(AutoMinorLocator())
    ax.yaxis.set_minor_locator(AutoMinorLocator())
    keys = []
    if any_low:
        keys.append(mlines.Line2D([], [], color=COLOR_LOW,
                                  label=f"The region with the depth in the range of (0, {spec.depth_min}*mean_depth]"))
    if any_zero:
        keys.append(mlines.Line2D([], [], color=COLOR_ZERO, label="The region of zero depth"))
    keys.append(mlines.Line2D([], [], color="r", ls="-.", dash_capstyle="butt", lw=1, label="Mean Coverage"))
    ax.add_artist(plt.legend(handles=keys, loc="lower center", bbox_to_anchor=(0.5, 1), ncols=len(keys)))
    plt.xlabel("Genomic Position (Mb)", fontsize=14)
    plt.ylabel("Depth", fontsize=14)
    plt.xticks(fontsize=12)
    plt.yticks(fontsize=12)
    plt.title(spec.title, fontsize=18, pad=30)
    plt.tight_layout()
    plt.savefig(spec.path, dpi=200)
    plt.close()


def plot_depth(depths_list: Sequence[pipeline.DepthTracks] = (), depth_min=0.1, depth_max=4.0, window_size=50000,
               image_type="png", directory=".", prefix="GCI", force=False, targets_length: Optional[Dict[str, int]] = None,
               dist_percent=0.005, regions_bed: Optional[Dict[str, list]] = None, threshold=0):
    """plot_depth of the reference (GCI.py:837-895): same checks, messages and file names."""
    targets_length = targets_length or {}
    regions_bed = regions_bed or {}
    if image_type not in ("pdf", "png"):
        sys.exit("ERROR!!! The format of output images only supports pdf and png")
    mean_depths = [tracks.mean() for tracks in depths_list]              # np.mean over all contigs (GCI.py:862-868)
    max_depths = [m * depth_max for m in mean_depths]
    targets = depths_list[0].targets                                     # (contig-sharded run: the contigs of this rank)
    sharded, root = pipeline._sharded(), pipeline._is_root()
    for target in depths_list[0].all_targets:
        path = f"{directory}/images/{prefix}.{target}.{image_type}"
        pipeline.refuse_overwrite(path, force)
    print("Plotting whole genome depth ...")
    averaged, y_frac, y_min, y_max = pipeline.pre_plot_base(depths_list, max_depths, window_size, 0)
    specs = [figure_spec(depths_list, target, averaged, mean_depths, y_frac, 0, depth_min, dist_percent, y_min, y_max,
                         image_type, directory, prefix, targets_length[target], False, threshold) for target in targets]
    if sharded:                                  # the numbers of every figure go to rank 0, which draws them in header order
        order = {f"{directory}/images/{prefix}.{t}.{image_type}": i for i, t in enumerate(depths_list[0].all_targets)}
        specs = sorted((x for part in pipeline.SHARD.gather_objects(specs) for x in part), key=lambda sp: order[sp.path])
    if root:
        for spec in specs:
            render(spec)
    print("Plotting whole genome depth done!!!\n\n")
    if len(regions_bed) > 0:
        print("Plotting depth for regions ...")
        specs = []
        for target, segments in regions_bed.items():
            for segment in segments:
                start, end = segment[0], segment[1]
                path = f"{directory}/images/{prefix}.{target}:{start}-{end}.{image_type}"
                pipeline.refuse_overwrite(path, force)
                if target not in depths_list[0]:
                    continue                                             # another rank holds that contig
                averaged, y_frac, y_min, y_max = pipeline.pre_plot_base(depths_list, max_depths, window_size, start,
                                                                        region=(target, start, end))
                specs.append(figure_spec(depths_list, target, averaged, mean_depths, y_frac, start, depth_min, dist_percent, y_min,
                                         y_max, image_type, directory, prefix, end, True, threshold))
        if sharded:
            flat = [f"{directory}/images/{prefix}.{t}:{a}-{b}.{image_type}" for t, segs in regions_bed.items() for a, b in segs]
            order = {pth: i for i, pth in enumerate(flat)}
            specs = sorted((x for part in pipeline.SHARD.gather_objects(specs) for x in part), key=lambda sp: order[sp.path])
        if root:
            for spec in specs:
                render(spec)
        print("Plotting depth for regions done!!!\n\n")
