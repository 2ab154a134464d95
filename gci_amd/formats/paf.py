"""PAF text: tokenise the twelve mandatory columns the path uses
(/root/reference/GCI.py:218-229) and write synthetic files.

Columns (0-based): 0 qname, 1 qlen, 2 qstart, 3 qend, 4 strand, 5 tname, 6 tlen,
7 tstart, 8 tend, 9 nmatch, 10 alnlen, 11 mapq.
"""
from __future__ import annotations

from typing import Iterable, Sequence


def format_line(qname: str, qlen: int, qs: int, qe: int, strand: str, tname: str, tlen: int,
                ts: int, te: int, nmatch: int, alnlen: int, mapq: int, extra: Sequence[str] = ()) -> str:
    cols = [qname, str(qlen), str(qs), str(qe), strand, tname, str(tlen), str(ts), str(te),
            str(nmatch), str(alnlen), str(mapq), *extra]
    return "\t".join(cols) + "\n"


def write(path: str, lines: Iterable[str]) -> None:
    with open(path, "w") as f:
        for ln in lines:
            f.write(ln)
