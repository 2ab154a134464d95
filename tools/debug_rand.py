import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_seams as T
from gci_amd.formats import bam
from gci_amd.device import Engine, REC_DTYPE
from oracle import gci_oracle as O
import inspect, re
src = inspect.getsource(T.test_bam_filter_randomised_records)
# run generation part only
body = src.split("    d_bam, d_off = engine.to_device(stream)")[0]
body = "\n".join(l[4:] for l in body.splitlines()[1:])   # strip def + indent
ns = dict(np=np, bam=bam)
exec(body.replace('"""Differential', 'x = """Differential'), ns)
stream, offs = ns["stream"], ns["offs"]
e = Engine(0)
ref_sel = np.array([0, 1, 2, 3, 4], np.int32)
keep = np.ones(len(offs), bool)
while True:
    try:
        want = O.bam_filter_arrays(stream, offs[keep], ref_sel, 30, 50, 0.1, 0.9); break
    except O.OracleRecordError as ex:
        keep[np.flatnonzero(keep)[ex.rec]] = False
o = offs[keep]
got = e.bam_filter(e.to_device(stream), e.to_device(o), e.to_device(ref_sel), 30, 50, 0.1, 0.9).cpu().numpy().reshape(-1).view(REC_DTYPE)
p = want["passed"].astype(bool); g = (got["flags"] & 1).astype(bool)
bad = np.flatnonzero(p != g)
print("bad", len(bad), "of", len(o))
for i in bad[:12]:
    r = bam.decode_record(stream, o[i], restore_long_cigar=False)
    tags = bam.parse_aux(bytes(stream[int(o[i]):int(o[i]) + 4 + int(np.frombuffer(stream[int(o[i]):int(o[i])+4].tobytes(), '<i4')[0])])[-1:]) if False else None
    print("rec", i, "want", p[i], "got", g[i], "n_cigar", r.n_cigar_field, "name_len", len(r.name), "l_seq", r.l_seq, "aux tags", [(k, v[0]) for k, v in r.aux.items()], "off%16", int(o[i]) % 16)
