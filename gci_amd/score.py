"""Interval algebra and the GCI score on the host.

These run on interval lists of at most ~10^4 items per track (CHM13: 11, MH63: 2328), so they
stay on the host; the left-fold merge is inherently sequential.  Function names, argument
meaning and outputs mirror /root/reference/GCI.py:

    complement_merged_depth   GCI.py:422-462      compute_n50             GCI.py:465-480
    merge_merged_depth_bed    GCI.py:483-519      compute_index (text)    GCI.py:522-657

Interval lists are Python lists of (start, end) tuples keyed by contig, in contig order.
"""
from __future__ import annotations

from math import log2
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

Bed = Dict[str, List[Tuple[int, int]]]

RULE = "-" * 136 + "\n\n\n"
GCI_COLUMNS = ("Chromosome\tTheoretical maximum N50\tCurated N50\tTheoretical minimum contigs number\t"
               "Curated contigs number\tGCI score\n")


def _bounds(length: int, flank_len: int, start, end):
    if start is not None and end is not None:
        return start, end
    return flank_len, length - flank_len


def complement_merged_depth(merged_depths_bed: Bed, targets_length: Dict[str, int], flank_len: int = 15,
                            start: Optional[int] = None, end: Optional[int] = None) -> Dict[str, List[int]]:
    """Lengths of the stretches of [start, end) not covered by the issue intervals.  A stretch is
    measured from the END OF THE PREVIOUS interval only (overlapping / unsorted input is not
    normalised -- the reference does not either)."""
    out: Dict[str, List[int]] = {}
    for target, length in targets_length.items():
        lo, hi = _bounds(length, flank_len, start, end)
        segs = merged_depths_bed[target]
        if not segs:
            out[target] = [hi - lo]
            continue
        a = np.asarray(segs, dtype=np.int64).reshape(-1, 2)
        prev_end = np.concatenate(([lo], a[:-1, 1]))
        gaps = a[:, 0] - prev_end
        lens = [int(g) for g in gaps[gaps > 0]]
        if hi > a[-1, 1]:
            lens.append(int(hi - a[-1, 1]))
        out[target] = lens
    return out


def compute_n50(lengths: Sequence[int]) -> int:
    if len(lengths) == 0:
        return 0
    srt = np.sort(np.asarray(lengths, dtype=np.int64))[::-1]
    cum = np.cumsum(srt)
    k = int(np.argmax(cum >= cum[-1] / 2))          # first index whose cumsum reaches half the total
    if not (cum[k] >= cum[-1] / 2):
        return 0
    return int(srt[k])


def merge_merged_depth_bed(merged_depths_bed: Bed, targets_length: Dict[str, int], dist_percent: float = 0.005,
                           flank_len: int = 15, start: Optional[int] = None, end: Optional[int] = None) -> Bed:
    """Left fold: an interval joins the current one when the gap to it is <= length * dist_percent
    (f64); the fold is seeded with the empty interval (start, start) and finally stretched to
    `end` when that is within reach, so an empty list yields [(start, start)] or [(start, end)]."""
    out: Bed = {}
    for target, length in targets_length.items():
        reach = length * dist_percent
        lo, hi = _bounds(length, flank_len, start, end)
        cur_s, cur_e = lo, lo
        res: List[Tuple[int, int]] = []
        for s, e in merged_depths_bed[target]:
            if s - cur_e <= reach:
                cur_e = e
            else:
                res.append((cur_s, cur_e))
                cur_s, cur_e = s, e
        if hi - cur_e <= reach:
            cur_e = hi
        res.append((cur_s, cur_e))
        out[target] = res
    return out


def gci_score(obs_n50, exp_n50, obs_num_ctg, exp_num_ctg):
    """GCI.py:601-604; returns int 0 (not 0.0) when there is no curated contig, as the reference."""
    if obs_num_ctg == 0:
        return 0
    return round(100 * log2(obs_n50 / exp_n50 + 1) / log2(obs_num_ctg / exp_num_ctg + 1), 4)


def expected_table(targets_length: Dict[str, int], chrs_list: Sequence[str] = ()):
    """GCI.py:553-565: per contig the theoretical maximum N50 (its length) and contig count (1), plus the genome row
    ('Genome', or 'All_chromosomes' with --chrs).  -> (row labels, exp_n50, exp_ctg)"""
    genome = "Genome" if len(chrs_list) == 0 else "All_chromosomes"
    rows = list(targets_length.keys()) + [genome]
    exp_n50 = dict(targets_length)
    exp_n50[genome] = compute_n50(list(targets_length.values()))
    exp_ctg = {t: 1 for t in targets_length}
    exp_ctg[genome] = len(targets_length)
    return rows, exp_n50, exp_ctg


def curated_table(bed: Bed, targets_length: Dict[str, int], flank_len: int, dist_percent: float, genome: str):
    """GCI.py:568-588: curated N50 from the complement of the issues as they are, curated contig count from the
    complement of the dp-merged issues.  -> (obs_n50, obs_ctg), both with the genome row."""
    free = complement_merged_depth(bed, targets_length, flank_len)
    obs_n50 = {t: compute_n50(v) for t, v in free.items()}
    obs_n50[genome] = compute_n50([x for v in free.values() for x in v])
    merged = merge_merged_depth_bed(bed, targets_length, dist_percent, flank_len)
    free2 = complement_merged_depth(merged, targets_length, flank_len)
    obs_ctg = {t: len(v) for t, v in free2.items()}
    obs_ctg[genome] = sum(len(v) for v in free2.values())
    return obs_n50, obs_ctg


def section_text(label: str, rows: Sequence[str], exp_n50, exp_ctg, obs_n50, obs_ctg) -> str:
    """One read type's block of the `.gci` file (GCI.py:592-606)."""
    parts = [f"{label}:\n", GCI_COLUMNS]
    for t in rows:
        parts.append(f"{t}\t{exp_n50[t]}\t{obs_n50[t]}\t{exp_ctg[t]}\t{obs_ctg[t]}\t"
                     f"{gci_score(obs_n50[t], exp_n50[t], obs_ctg[t], exp_ctg[t])}\n")
    parts.append(RULE)
    return "".join(parts)


def index_text(targets_length: Dict[str, int], merged_depths_bed_list: Sequence[Bed], type_list: Sequence[str],
               flank_len: int = 15, dist_percent: float = 0.005, chrs_list: Sequence[str] = ()) -> str:
    """The `.gci` file body (GCI.py:553-607)."""
    rows, exp_n50, exp_ctg = expected_table(targets_length, chrs_list)
    parts: List[str] = []
    for label, bed in zip(type_list, merged_depths_bed_list):
        obs_n50, obs_ctg = curated_table(bed, targets_length, flank_len, dist_percent, rows[-1])
        parts.append(section_text(label, rows, exp_n50, exp_ctg, obs_n50, obs_ctg))
    return "".join(parts)


def regions_text(regions_bed: Bed, type_list: Sequence[str], n_tracks: int,
                 region_issues: Callable[[int, str, int, int], List[Tuple[int, int]]],
                 dist_percent: float = 0.005, warn: Callable[[str], None] = lambda m: None) -> str:
    """The `.regions.gci` body (GCI.py:610-657).  `region_issues(track, target, start, end)` must
    return collapse_depth_range({target: depths[target][start:end]}, -1, threshold, 0, start)."""
    out = ["Chromosome\tStart\tEnd\t" + "\t".join(type_list) + "\n"]
    all_exp: List[int] = []
    all_free: List[List[int]] = [[] for _ in range(n_tracks)]
    all_ctg = [0] * n_tracks
    for target, segments in regions_bed.items():
        for (start, end) in segments:
            exp_n50 = end - start
            if exp_n50 > 0:
                all_exp.append(exp_n50)
            else:
                warn(f'Warning!!! The region "{target}:{start}-{end}" is not available')
            scores = []
            for i in range(n_tracks):
                bed = {target: region_issues(i, target, start, end)}
                tl = {target: exp_n50}
                free = complement_merged_depth(bed, tl, start, start, end)[target]
                obs_n50 = compute_n50(free)
                merged = merge_merged_depth_bed(bed, tl, dist_percent, start, start, end)
                obs_ctg = len(complement_merged_depth(merged, tl, start, start, end)[target])
                if exp_n50 > 0:
                    all_free[i] += free
                    all_ctg[i] += obs_ctg
                scores.append(gci_score(obs_n50, exp_n50, obs_ctg, 1))
            out.append(f"{target}\t{start}\t{end}\t" + "\t".join(map(str, scores)) + "\n")
    exp_all = compute_n50(all_exp)
    n_all = len(all_exp)
    totals = [gci_score(compute_n50(all_free[i]), exp_all, all_ctg[i], n_all) for i in range(n_tracks)]
    out.append(RULE)
    out.append("All_regions\t*\t*\t" + "\t".join(map(str, totals)) + "\n")
    return "".join(out)
