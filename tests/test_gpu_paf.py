"""K2 (R3 / N4): the PAF path of filter() on the GPU (gci_paf_filter_device, k_paf.hip) against the oracle
(oracle.paf_filter, GCI.py:211-254 statement for statement), the plain-Python statement of the rules and the native
host filter: golden PAFs, a hand-made file (ties broken by target name, touching blocks, longest block), randomised
multi-file inputs (the block table is never reset between files), odd line ends and blanks, and the lines the reference
raises on -- with the line number of the FIRST one."""
import numpy as np
import pytest

from golden_util import inputs, manifest
from gci_amd._lib import GciError, GCI_E_MALFORMED, GCI_E_ZERO_DIV, REC_HQ
from gci_amd.device import REC_DTYPE
from paf_ref import paf_filter_py

pytestmark = pytest.mark.gpu


def gpu_dicts(engine, paths, targets, mq, cut, ip):
    """engine.paf_filter as the reference's (paf_lines, high_qual): per file name -> (target, start, end, qlen)."""
    per_file, hq = [], set()
    for ji in engine.paf_filter(paths, targets, mq, cut, ip):
        r = ji.recs.cpu().numpy().reshape(-1).view(REC_DTYPE)
        off = ji.name_off.cpu().numpy()
        text = ji.name_base.cpu().numpy()
        d = {}
        for i in range(r.shape[0]):
            q = bytes(text[int(off[i]):int(off[i]) + int(r["name_len"][i])]).decode()
            assert q not in d
            d[q] = (targets[int(r["contig"][i])], int(r["start"][i]), int(r["end"][i]), int(r["qlen"][i]))
            if int(r["flags"][i]) & REC_HQ:
                hq.add(q)
        per_file.append(d)
    return per_file, hq


def test_golden_and_hand_made_pafs(engine, oracle, tmp_path):
    case = "c5_two_type"
    m = manifest(case)
    pafs = [p for p in inputs(case, m["hifi"] + m["nano"]) if p.endswith(".paf")]
    targets = ["mat_chr1", "pat_chr1", "mat_chr2"]
    for sel in (targets, targets[:2]):
        for args in ((30, 50, 0.9), (10, 60, 0.99)):
            got = gpu_dicts(engine, pafs, sel, *args)          # two files: exercises the un-reset block table
            want = oracle.paf_filter(pafs, sel, *args)
            assert got[1] == want[1] and got[0] == want[0]
    p = tmp_path / "t.paf"
    rows = [("q1", 1000, 0, 400, "+", "tB", 9000, 100, 500, 400, 400, 60), ("q1", 1000, 400, 800, "+", "tB", 9000, 500, 900, 400, 400, 60),
            ("q1", 1000, 0, 800, "+", "tA", 9000, 2000, 2800, 800, 800, 60), ("q2", 500, 0, 100, "+", "tA", 9000, 10, 110, 95, 100, 5),
            ("q3", 500, 0, 100, "+", "tZ", 9000, 10, 110, 100, 100, 60), ("q4", 500, 0, 200, "-", "tA", 9000, 50, 250, 199, 200, 49),
            ("q4", 500, 300, 350, "-", "tA", 9000, 5000, 5050, 50, 50, 49)]
    p.write_text("".join("\t".join(map(str, r)) + "\n" for r in rows))
    got = gpu_dicts(engine, [str(p)], ["tA", "tB"], 30, 50, 0.9)
    assert got == oracle.paf_filter([str(p)], ["tA", "tB"], 30, 50, 0.9)
    assert got[0][0]["q1"] == ("tB", 100, 900, 1000) and got[0][0]["q4"] == ("tA", 50, 250, 500) and got[1] == {"q1"}


def test_randomised_files_line_ends_and_blanks(engine, oracle, tmp_path):
    rng = np.random.default_rng(11)
    targets = ["t%d" % i for i in range(6)]
    paths = []
    for f in range(3):
        rows = []
        for _ in range(3000):
            q = "read_%d" % int(rng.integers(0, 700))
            qlen = int(rng.choice([1000, 2000, 5000]))
            qs = int(rng.integers(0, qlen // 100)) * 50
            qe = min(qlen, qs + int(rng.integers(1, 20)) * 50)
            t = str(rng.choice(targets + ["other"]))
            ts = int(rng.integers(0, 100)) * 100
            aln = qe - qs
            nm = int(aln * rng.choice([0.85, 0.9, 0.95, 1.0]))
            cols = list(map(str, (q, qlen, qs, qe, "+-"[int(rng.integers(0, 2))], t, 100000, ts, ts + aln, nm, aln,
                                  int(rng.choice([0, 29, 30, 49, 50, 60])))))
            u = rng.random()
            if u < 0.5:
                cols += ["tp:A:P", "cm:i:5"]                    # extra columns
            elif u < 0.6:
                cols[11] = cols[11] + "  "                      # blanks the reference's strip() / int() swallow
            elif u < 0.7:
                cols[1] = " +" + cols[1]
            if len(cols[7]) >= 2 and rng.random() < 0.1:
                cols[7] = cols[7][0] + "_" + cols[7][1:]           # int('1_200') is 1200
            row = "\t".join(cols)
            if 0.7 <= u < 0.75:
                row = "  " + row + " \t "
            rows.append(row)
        p = tmp_path / ("f%d.paf" % f)
        sep = ("\r\n", "\n", "\r")[f]
        p.write_bytes(sep.join(rows).encode() + (b"" if f == 2 else sep.encode()))
        paths.append(str(p))
    for sel in (targets, targets[1:4]):
        for args in ((30, 50, 0.9), (0, 60, 0.0), (50, 30, 0.95)):
            got, py, want = gpu_dicts(engine, paths, sel, *args), paf_filter_py(paths, sel, *args), oracle.paf_filter(paths, sel, *args)
            assert got[1] == py[1] == want[1]
            assert got[0] == py[0] == want[0]
            assert len(got[0][2]) >= len(got[0][0]) > 20


def test_large_file_and_error_lines(engine, oracle, tmp_path):
    from gci_amd import hostio
    rng = np.random.default_rng(12)
    targets = ["t%d" % i for i in range(6)]
    big = []
    for f in range(2):
        rows = []
        for i in range(60_000):
            q = "m64011_190830_220126/%d/ccs" % int(rng.integers(0, 30_000))
            qlen = int(rng.choice([9000, 15000, 21000]))
            qs = int(rng.integers(0, 50)) * 100
            qe = min(qlen, qs + int(rng.integers(10, 150)) * 100)
            t = str(rng.choice(targets + ["other"]))
            ts = int(rng.integers(0, 900)) * 100
            aln = qe - qs
            rows.append("\t".join(map(str, (q, qlen, qs, qe, "+", t, 100000, ts, ts + aln, int(aln * rng.choice([0.88, 0.93, 0.99])), aln,
                                            int(rng.choice([0, 29, 30, 49, 50, 60])), "tp:A:P"))))
        p = tmp_path / ("big%d.paf" % f)
        p.write_bytes(("\r\n" if f else "\n").join(rows).encode() + b"\n")
        big.append(str(p))
    got = gpu_dicts(engine, big, targets, 30, 50, 0.9)
    want = paf_filter_py(big, targets, 30, 50, 0.9)
    assert got[1] == want[1] and got[0] == want[0]
    # the native host filter emits the same records (as sets: the device's order is unspecified)
    host = hostio.paf_filter(big, targets, 30, 50, 0.9)
    for ji, (r, nm, off) in zip(engine.paf_filter(big, targets, 30, 50, 0.9), host):
        a = ji.recs.cpu().numpy().reshape(-1).view(REC_DTYPE)
        b = r.reshape(-1).view(REC_DTYPE)
        key = lambda x: sorted(zip(x["name_hash"].tolist(), x["contig"].tolist(), x["start"].tolist(), x["end"].tolist(),   # noqa: E731
                                   x["qlen"].tolist(), x["flags"].tolist(), x["name_len"].tolist()))
        assert key(a) == key(b)
    # the FIRST offending line is reported
    lines = open(big[0], "rb").read().split(b"\n")
    lines[7000] = lines[7000].replace(b"\t", b" ", 20)                      # no tabs left: IndexError in the reference
    lines[9000] = b"x"
    (tmp_path / "bigbad.paf").write_bytes(b"\n".join(lines))
    with pytest.raises(GciError) as err:
        engine.paf_filter([str(tmp_path / "bigbad.paf")], targets, 30, 50, 0.9)
    assert err.value.status == GCI_E_MALFORMED and err.value.rec == 7001
    bad = tmp_path / "bad.paf"
    for text, status in (("q\t100\t0\t50\t+\n", GCI_E_MALFORMED), ("q\t100\t0\t50\t+\tt0\t1000\t0\t50\t50\t0\t60\n", GCI_E_ZERO_DIV),
                         ("q\t100\t0\tx\t+\tt0\t1000\t0\t50\t50\t50\t60\n", GCI_E_MALFORMED),
                         ("q\t100\t0\t50\t+\tt0\t1000\t0\t50\t50\t50\n", GCI_E_MALFORMED), ("\n", GCI_E_MALFORMED),
                         ("q\t0\t0\t50\t+\tt0\t1000\t0\t50\t50\t50\t60\n", GCI_E_ZERO_DIV)):          # qlen 0: raised while scoring
        bad.write_text(text)
        with pytest.raises(GciError) as e:
            engine.paf_filter([str(bad)], ["t0"], 30, 50, 0.9)
        assert e.value.status == status, text
        with pytest.raises((IndexError, ValueError, ZeroDivisionError)):
            paf_filter_py([str(bad)], ["t0"], 30, 50, 0.9)
    # a scoring error in an EARLIER file comes before a malformed line of a later one (the reference reads and scores
    # file after file)
    f0, f1 = tmp_path / "zero_qlen.paf", tmp_path / "later_bad.paf"
    f0.write_text("q\t0\t0\t50\t+\tt0\t1000\t0\t50\t50\t50\t60\n")
    f1.write_text("x\n")
    with pytest.raises(GciError) as e:
        engine.paf_filter([str(f0), str(f1)], ["t0"], 30, 50, 0.9)
    assert e.value.status == GCI_E_ZERO_DIV
    bad.write_text("q\t100\t0\tx\t+\tother\t1000\t0\t50\t50\t50\t60\n")      # not a selected target: never parsed
    assert gpu_dicts(engine, [str(bad)], ["t0"], 30, 50, 0.9) == ([{}], set())
    empty = tmp_path / "empty.paf"
    empty.write_bytes(b"")
    assert gpu_dicts(engine, [str(empty), str(bad)], ["t0"], 30, 50, 0.9) == ([{}, {}], set())


@pytest.mark.parametrize("world", [2, 5])
def test_byte_ranges_routed_hits_scored_per_owner(engine, oracle, tmp_path, world):
    """The two halves of the PAF filter the sharded command line uses (shard.paf_by_byte_range), with the ranks played one after
    the other on this GPU: stage A over every rank's byte range of three files (gci_paf_hits_device), the hits routed by query
    hash (gci_route_hits), the buckets of owner d put together as its all-to-all would deliver them, stage B per owner
    (gci_paf_score_device) -- the union over the owners is the whole-file filter's per-file dicts, every query on its owner."""
    import torch
    from gci_amd import shard
    from gci_amd.device import name_hash_np
    rng = np.random.default_rng(21)
    targets = ["t%d" % i for i in range(5)]
    paths = []
    for f in range(3):
        rows = []
        for _ in range(2500):
            q = "m64/%d/ccs" % int(rng.integers(0, 900)) if rng.random() < 0.9 else "a_rather_long_query_name_%d_of_more_than_48_bytes_in_all____" % int(rng.integers(0, 50))
            qlen = int(rng.choice([1000, 2000, 5000]))
            qs = int(rng.integers(0, qlen // 100)) * 50
            qe = min(qlen, qs + int(rng.integers(1, 20)) * 50)
            ts = int(rng.integers(0, 100)) * 100
            aln = qe - qs
            rows.append("\t".join(map(str, (q, qlen, qs, qe, "+", str(rng.choice(targets + ["other"])), 100000, ts, ts + aln,
                                            int(aln * rng.choice([0.85, 0.9, 0.95, 1.0])), aln, int(rng.choice([0, 29, 30, 49, 50, 60])), "tp:A:P"))))
        p = tmp_path / ("r%d.paf" % f)
        sep = ("\n", "\r\n", "\r")[f]
        p.write_bytes(sep.join(rows).encode() + (b"" if f == 1 else sep.encode()))
        paths.append(str(p))
    args = (30, 50, 0.9)
    want, want_hq = oracle.paf_filter(paths, targets, *args)
    raws = [np.fromfile(p, dtype=np.uint8) for p in paths]
    dev = engine.device
    B = engine.PAF_HIT_BYTES
    # stage A + routing, rank after rank
    sent = []                                                     # [rank][file] = (buckets, names, cap)
    slot_bytes = 64
    for r in range(world):
        parts = [raw[slice(*shard.byte_range_of_rank(raw, r, world))] for raw in raws]
        ends = np.cumsum([x.shape[0] for x in parts], dtype=np.uint64)
        d_text = engine.to_device(np.concatenate(parts + [np.zeros(16, np.uint8)]))
        hits = engine.paf_hits_text(d_text, ends, targets, *args)
        per_file = []
        for f, h in enumerate(hits):
            cap = int(h.shape[0]) + 8
            sh = torch.zeros((world * (cap + 1), B), dtype=torch.uint8, device=dev)
            sn = torch.zeros(world * cap * slot_bytes, dtype=torch.uint8, device=dev)
            st = torch.full((1,), -1, dtype=torch.int64, device=dev)
            engine.route_hits(h, d_text, world, cap, sh, sn, slot_bytes, st)
            assert int(st.item()) == -1
            per_file.append((sh.view(world, cap + 1, B), sn.view(world, cap, slot_bytes), cap))
        sent.append(per_file)
    assert sum(int(b[:, 0, 8:16].contiguous().view(torch.int64).sum().item()) for pf in sent for b, _, _ in pf) > 3000
    # stage B per owner
    got = [dict() for _ in paths]
    got_hq = set()
    for d in range(world):
        dense, names, upto, base = [], [], [0], 0
        for f in range(len(paths)):
            n_f = 0
            for r in range(world):
                bk, nm, cap = sent[r][f]
                c = int(bk[d, 0, 8:16].contiguous().view(torch.int64).item())
                part = bk[d, 1:1 + c].clone()
                off = base + torch.arange(c, dtype=torch.int64, device=dev) * slot_bytes
                if c:
                    part[:, 0:8] = off.view(torch.uint8).view(c, 8)
                dense.append(part)
                names.append(nm[d, :c].reshape(-1))
                base += c * slot_bytes
                n_f += c
            upto.append(upto[-1] + n_f)
        d_names = torch.cat(names + [torch.zeros(16, dtype=torch.uint8, device=dev)])
        d_hits = torch.cat(dense) if upto[-1] else torch.zeros((1, B), dtype=torch.uint8, device=dev)
        for f, ji in enumerate(engine.paf_score_hits(d_names, d_hits, upto, targets)):
            r = ji.recs.cpu().numpy().reshape(-1).view(REC_DTYPE)
            off = ji.name_off.cpu().numpy()
            text = ji.name_base.cpu().numpy()
            for i in range(r.shape[0]):
                q = bytes(text[int(off[i]):int(off[i]) + int(r["name_len"][i])])
                assert (int(name_hash_np([q])[0]) >> 33) % world == d and int(r["name_hash"][i]) == int(name_hash_np([q])[0])
                assert q.decode() not in got[f]
                got[f][q.decode()] = (targets[int(r["contig"][i])], int(r["start"][i]), int(r["end"][i]), int(r["qlen"][i]))
                if int(r["flags"][i]) & REC_HQ:
                    got_hq.add(q.decode())
    assert got == want and got_hq == want_hq and len(got[2]) > len(got[0]) > 100
    # a line the reference raises on: stage A reports it (the sharded caller then takes the whole files)
    bad = tmp_path / "bad.paf"
    bad.write_text("q\t100\t0\t50\t+\tt0\t1000\t0\t50\tfifty\t50\t60\n")
    raw = np.fromfile(str(bad), dtype=np.uint8)
    with pytest.raises(GciError) as e:
        engine.paf_hits_text(engine.to_device(np.concatenate([raw, np.zeros(16, np.uint8)])), np.asarray([raw.shape[0]], dtype=np.uint64),
                             targets, *args)
    assert e.value.status == GCI_E_MALFORMED
