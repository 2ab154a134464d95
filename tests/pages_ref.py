"""Plain-Python statement of the PAGED HEADS format (include/gci_hip.h, "record pages"): what gci_bam_pages_write must
produce byte for byte.  Test infrastructure only."""
import struct

import numpy as np

MAX_REC = 1024
MAGIC = 0x31504347            # "GCP1"
F_EXT, F_OVERSIZE, F_MALFORMED = 1, 2, 4


def a16(x):
    return (x + 15) & ~15


def _measure(stream, off, n_bytes, has_seq):
    """-> dict describing record at `off` the way the converter must see it."""
    if off + 36 > n_bytes:
        return dict(kind=F_MALFORMED, size=48, blob=0, core=bytes(36) if off + 36 > n_bytes else None, short=True)
    core = bytes(stream[off:off + 36])
    block_size, ref_id, pos, lrn, mapq, _bin, n_cig, flag, l_seq, _nr, _np, _tl = struct.unpack("<iiiBBHHHiiii", core)
    seq = (((l_seq + 1) >> 1) + l_seq) if (has_seq and l_seq >= 0) else 0
    rec_end = off + 4 + (block_size & 0xFFFFFFFF)
    aux_off = off + 36 + lrn + 4 * n_cig + seq
    if block_size < 32 or rec_end > n_bytes or l_seq < 0 or aux_off > rec_end:
        return dict(kind=F_MALFORMED, size=48, blob=0, core=core, short=False)
    aux_len = rec_end - aux_off
    name0 = off + 36
    cig0 = name0 + lrn
    d = dict(core=core, lrn=lrn, n_cig=n_cig, aux_len=aux_len, name=bytes(stream[name0:name0 + lrn]),
             cigar=bytes(stream[cig0:cig0 + 4 * n_cig]), aux=bytes(stream[aux_off:rec_end]), short=False)
    cig_at = a16(36 + lrn)
    full = a16(cig_at + a16(4 * n_cig) + aux_len)
    if full <= MAX_REC:
        d.update(kind=0, size=full, blob=0)
    elif a16(cig_at + 16 + aux_len) <= MAX_REC:
        d.update(kind=F_EXT, size=a16(cig_at + 16 + aux_len), blob=a16(4 * n_cig))
    else:
        heads_len = 36 + lrn + 4 * n_cig + aux_len
        d.update(kind=F_OVERSIZE, size=48, blob=a16(heads_len))
    return d


def build_pages(stream, offsets, has_seq, page_bytes):
    """-> (buffer uint8, n_pages, blob_off).  buffer = pages | blob | 16 zero bytes."""
    stream = np.asarray(stream, dtype=np.uint8)
    n_bytes = int(stream.shape[0])
    Q = page_bytes - MAX_REC - 48
    recs = [_measure(stream, int(o), n_bytes, has_seq) for o in offsets]
    S, B, s, b = [], [], 0, 0
    for r in recs:
        S.append(s)
        B.append(b)
        s += r["size"] + 2
        b += r["blob"]
    n_pages = (S[-1] // Q + 1) if recs else 0
    blob_off = n_pages * page_bytes
    buf = bytearray(blob_off + b + 16)
    by_page = {}
    for i, r in enumerate(recs):
        by_page.setdefault(S[i] // Q, []).append(i)
    for k in range(n_pages):
        idx = by_page.get(k, [])
        base = k * page_bytes
        at = 16 + a16(2 * len(idx))
        for j, i in enumerate(idx):
            r = recs[i]
            struct.pack_into("<H", buf, base + 16 + 2 * j, at >> 4)
            p = base + at
            core = bytearray(r["core"] if not r["short"] else bytes(36))
            struct.pack_into("<I", core, 0, r["size"])
            struct.pack_into("<H", core, 14, r["kind"])
            if r["kind"] == F_MALFORMED:
                struct.pack_into("<IQ", core, 24, 0, 0)
                buf[p:p + 36] = core
            else:
                struct.pack_into("<IQ", core, 24, r["aux_len"], (blob_off + B[i]) if r["blob"] else 0)
                buf[p:p + 36] = core
                if r["kind"] == F_OVERSIZE:
                    q = blob_off + B[i]
                    heads = bytearray(r["core"]) + r["name"] + r["cigar"] + r["aux"]
                    struct.pack_into("<i", heads, 0, len(heads) - 4)
                    buf[q:q + len(heads)] = heads
                else:
                    buf[p + 36:p + 36 + r["lrn"]] = r["name"]
                    c = p + a16(36 + r["lrn"])
                    if r["kind"] == F_EXT:
                        q = blob_off + B[i]
                        buf[q:q + len(r["cigar"])] = r["cigar"]
                        buf[c:c + 4] = r["cigar"][:4]          # the first operation stays visible in the page
                        c += 16
                    else:
                        buf[c:c + len(r["cigar"])] = r["cigar"]
                        c += a16(len(r["cigar"]))
                    buf[c:c + r["aux_len"]] = r["aux"]
            at += r["size"]
        struct.pack_into("<IIII", buf, base, len(idx), (idx[0] if idx else 0) & 0xFFFFFFFF, at if idx else 16, MAGIC)
    return np.frombuffer(bytes(buf), dtype=np.uint8), n_pages, blob_off
