"""Helpers shared by the CPU and GPU tests of the BAM ingestion."""
import numpy as np


def heads_expected(stream: np.ndarray, offs: np.ndarray, first: int):
    """The heads stream written out from its definition: header, then every record minus SEQ / QUAL."""
    out, new_offs = [stream[:first].tobytes()], []
    w = first
    for o in offs.tolist():
        bs = int(stream[o:o + 4].view(np.int32)[0])
        l_name, n_cig = int(stream[o + 12]), int(stream[o + 16:o + 18].view(np.uint16)[0])
        l_seq = int(stream[o + 20:o + 24].view(np.int32)[0])
        head, seq = 36 + l_name + 4 * n_cig, (l_seq + 1) // 2 + l_seq
        if l_seq < 0 or head + seq > 4 + bs:
            rec = bytearray(stream[o:o + 36].tobytes())
            rec[20:24] = np.int32(-1).tobytes()
        else:
            rec = bytearray(stream[o:o + head].tobytes() + stream[o + head + seq:o + 4 + bs].tobytes())
        rec[0:4] = np.int32(len(rec) - 4).tobytes()
        new_offs.append(w)
        w += len(rec)
        out.append(bytes(rec))
    return b"".join(out), np.asarray(new_offs, dtype=np.uint64)
