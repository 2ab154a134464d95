"""PAF text: tokenise the twelve mandatory columns the path uses
(/root/reference/GCI.py:218-229) and write synthetic files.

Columns (0-based): 0 qname, 1 qlen, 2 qstart, 3 qend, 4 strand, 5 tname, 6 tlen,
7 tstart, 8 tend, 9 nmatch, 10 alnlen, 11 mapq.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import numpy as np


@dataclass
class PafTable:
    """Column-wise view of one PAF file, rows in file order."""
    qname: List[str]
    tname: List[str]
    qlen: np.ndarray
    qstart: np.ndarray
    qend: np.ndarray
    tstart: np.ndarray
    tend: np.ndarray
    nmatch: np.ndarray
    alnlen: np.ndarray
    mapq: np.ndarray

    def __len__(self) -> int:
        return len(self.qname)


def read_table(path: str) -> PafTable:
    qname: List[str] = []
    tname: List[str] = []
    nums: List[Tuple[int, ...]] = []
    with open(path, "r") as f:
        for line in f:
            c = line.strip().split("\t")
            qname.append(c[0])
            tname.append(c[5])
            nums.append((int(c[1]), int(c[2]), int(c[3]), int(c[7]), int(c[8]), int(c[9]), int(c[10]), int(c[11])))
    a = np.asarray(nums, dtype=np.int64).reshape(-1, 8)
    return PafTable(qname, tname, a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5], a[:, 6], a[:, 7])


def format_line(qname: str, qlen: int, qs: int, qe: int, strand: str, tname: str, tlen: int,
                ts: int, te: int, nmatch: int, alnlen: int, mapq: int, extra: Sequence[str] = ()) -> str:
    cols = [qname, str(qlen), str(qs), str(qe), strand, tname, str(tlen), str(ts), str(te),
            str(nmatch), str(alnlen), str(mapq), *extra]
    return "\t".join(cols) + "\n"


def write(path: str, lines: Iterable[str]) -> None:
    with open(path, "w") as f:
        for ln in lines:
            f.write(ln)
