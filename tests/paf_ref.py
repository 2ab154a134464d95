"""The PAF path of filter() (/root/reference/GCI.py:211-254) restated in plain Python over the package's own helpers: a second
opinion next to the oracle for the native host filter and the device kernels (tests/test_host_logic.py, tests/test_gpu_paf.py).
Test infrastructure only."""
from typing import Dict, List, Sequence, Set, Tuple

from gci_amd.pipeline import _merge_span


def paf_filter_py(paf_files: Sequence[str], targets: Sequence[str], map_qual: int, mq_cutoff: int, iden_percent: float
                  ) -> Tuple[List[Dict[str, Tuple[str, int, int, int]]], Set[str]]:
    """The same filter in plain Python (the readable statement of the rules; tests hold the native one against it).
    GCI.py:211-254.  Block lists accumulate ACROSS files (the reference creates `synteny` once,
    before the per-file loop), so file i re-emits every query seen in files < i."""
    tset = set(targets)
    high_qual: Set[str] = set()
    blocks: Dict[str, Dict[str, list]] = {}
    per_file: List[Dict[str, Tuple[str, int, int, int]]] = []
    for path in paf_files:
        with open(path, "r") as f:
            for line in f:
                col = line.strip().split("\t")
                if col[5] not in tset:
                    continue
                qlen, qs, qe, ts, te = int(col[1]), int(col[2]), int(col[3]), int(col[7]), int(col[8])
                nmatch, alnlen, mapq = int(col[9]), int(col[10]), int(col[11])
                identity = nmatch / alnlen
                if mapq >= map_qual and identity >= iden_percent:
                    blocks.setdefault(col[0], {}).setdefault(col[5], []).append((qlen, qs, qe, ts, te, identity))
                    if mapq >= mq_cutoff:
                        high_qual.add(col[0])
        emitted: Dict[str, Tuple[str, int, int, int]] = {}
        for query, by_target in blocks.items():
            best_key, best_val = None, None
            for target, alns in by_target.items():
                covered, _, _ = _merge_span([(a[1], a[2]) for a in alns])
                qlen = alns[0][0]
                total = 0
                for a in alns:                       # file-order f64 accumulation, as sum() does
                    total = total + a[5]
                rank = (total / len(alns) * (covered / qlen), target)
                if best_key is None or rank > best_key:
                    _, s, e = _merge_span([(a[3], a[4]) for a in alns])
                    best_key, best_val = rank, (target, s, e, qlen)
            emitted[query] = best_val
        per_file.append(emitted)
    return per_file, high_qual


# ==============================================================================================
# filter
# ==============================================================================================

