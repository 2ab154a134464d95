"""Stand-in for the twelve-attribute pysam surface GCI.py uses, so the
*unmodified* reference can be imported in this container to generate golden vectors
(tools/make_golden.py).  It is deliberately dumb: it decodes fields with this repo's BAM reader
and makes no filtering decision -- every decision stays in reference code.

Not shipped, not imported by the product, never present on the GPU box.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from gci_amd.formats import bam as _bam  # noqa: E402

_CACHE = {}


class AlignedSegment:
    __slots__ = ("_r", "_f")

    def __init__(self, rec, f):
        self._r, self._f = rec, f

    @property
    def flag(self):
        return self._r.flag

    @property
    def is_unmapped(self):
        return bool(self._r.flag & 0x4)

    @property
    def is_mapped(self):
        return not (self._r.flag & 0x4)

    @property
    def is_secondary(self):
        return bool(self._r.flag & 0x100)

    @property
    def is_supplementary(self):
        return bool(self._r.flag & 0x800)

    @property
    def mapping_quality(self):
        return self._r.mapq

    @property
    def query_name(self):
        return self._r.name

    @property
    def reference_name(self):
        return self._f.references[self._r.ref_id] if self._r.ref_id >= 0 else None

    @property
    def reference_start(self):
        return self._r.pos

    @property
    def reference_end(self):
        if (self._r.flag & 0x4) or len(self._r.cigar) == 0:
            return None
        rlen = sum(l for o, l in self._r.cigar if (_bam.REF_CONSUMING >> o) & 1)
        return self._r.pos + (rlen if rlen > 0 else 1)

    @property
    def query_length(self):
        return self._r.l_seq

    def get_cigar_stats(self):
        base = [0] * 11
        blocks = [0] * 11
        for o, l in self._r.cigar:
            base[o] += l
            blocks[o] += 1
        if "NM" in self._r.aux:
            base[10] = self._r.aux["NM"][1]
            blocks[10] = 1
        return base, blocks

    def get_tag(self, tag):
        if tag not in self._r.aux:
            raise KeyError("tag '%s' not present" % tag)
        return self._r.aux[tag][1]


class AlignmentFile:
    def __init__(self, path, mode="rb", threads=1, **kw):
        key = (os.path.abspath(path), os.path.getmtime(path))
        if key not in _CACHE:
            stream, hdr, offs = _bam.read_bam(path)
            recs = [_bam.decode_record(stream, o) for o in offs]
            ends = []
            for r in recs:
                rlen = sum(l for o, l in r.cigar if (_bam.REF_CONSUMING >> o) & 1)
                if r.flag & 0x4:
                    rlen = 0
                ends.append(r.pos + (rlen if rlen > 0 else 1))
            _CACHE[key] = (hdr, recs, ends)
        self._hdr, self._recs, self._ends = _CACHE[key]
        self.references = self._hdr.references
        self.lengths = self._hdr.lengths

    def fetch(self, contig=None, start=None, stop=None, multiple_iterators=False, **kw):
        tid = self.references.index(contig)
        lo = 0 if start is None else start
        hi = self.lengths[tid] if stop is None else stop
        for r, e in zip(self._recs, self._ends):
            if r.ref_id == tid and r.pos < hi and e > lo:
                yield AlignedSegment(r, self)

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
