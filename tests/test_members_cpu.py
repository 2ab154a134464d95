"""The run bookkeeping of the run-by-run BGZF ingestion (pipeline._Members / _runs_of_members) without a GPU: runs cut from a
member table that arrives in pieces cover every member once, in order, within the budget; a failure of a later piece surfaces where
the run behind the known ones is asked for."""
from concurrent.futures import Future

import numpy as np
import pytest

from gci_amd import pipeline


class _Rounds:
    def __init__(self, members_per_round):
        self.n = members_per_round

    def inflate_round(self):
        return self.n


def _table(rng, n):
    size = rng.integers(200, 30000, n).astype(np.uint64)
    isz = rng.integers(0, 65536, n).astype(np.uint64)
    isz[rng.integers(0, n, max(1, n // 50))] = 0                 # empty members (htslib's EOF blocks in concatenated files)
    pos = np.concatenate([[0], np.cumsum(size)]).astype(np.uint64)
    return pos, isz


def _loop_runs(engine, isz, chunk_bytes):
    """The statement the vectorised _runs_of_members replaces."""
    total, rnd, per_run = int(isz.sum()), engine.inflate_round(), 0
    if rnd > 0:
        per_run = max(1, int(chunk_bytes // max(1, total // max(1, len(isz)))) // rnd) * rnd
    groups, a, acc = [], 0, 0
    for i, sz in enumerate(isz.tolist()):
        if acc and (acc + sz > chunk_bytes or (per_run and i - a >= per_run)):
            groups.append((a, i))
            a, acc = i, 0
        acc += sz
    groups.append((a, len(isz)))
    return groups


def test_runs_of_members_equal_the_loop():
    rng = np.random.default_rng(11)
    for _ in range(200):
        n = int(rng.integers(1, 4000))
        _, isz = _table(rng, n)
        chunk = int(rng.integers(70_000, 6_000_000))
        eng = _Rounds(int(rng.choice([0, 1, 7, 64, 1000])))
        assert pipeline._runs_of_members(eng, isz, chunk) == _loop_runs(eng, isz, chunk)


@pytest.mark.parametrize("pieces", [1, 2, 3])
def test_members_in_pieces_cover_the_file(pieces):
    rng = np.random.default_rng(pieces)
    eng = _Rounds(16)
    pos, isz = _table(rng, 5000)
    chunk = 3_000_000
    cuts = sorted(rng.choice(np.arange(200, 4800), pieces - 1, replace=False).tolist()) + [5000]
    rests = [Future() for _ in cuts[1:]]
    first = cuts[0]
    m = pipeline._Members(eng, chunk, pos[:first + 1], isz[:first], rests[0] if rests else None)
    assert m.lazy() == (pieces > 1)
    known_before = len(m.run_bytes())
    for j, r in enumerate(rests):                                 # the pieces arrive (each the table from the file's start)
        c = cuts[j + 1]
        r.set_result((pos[:c + 1], isz[:c], rests[j + 1] if j + 1 < len(rests) else None))
    groups, k = [], 0
    while True:
        g = m.group(k)
        if g is None:
            break
        lo, hi = g
        assert int(m.pos[hi]) == int(pos[hi]) and m.isz.shape[0] >= hi       # the arrays cover the run that was asked for
        groups.append(g)
        k += 1
    assert not m.lazy() and m.is_last(k - 1) and not m.is_last(0) and known_before <= len(groups)
    assert groups[0][0] == 0 and groups[-1][1] == 5000
    assert all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
    for lo, hi in groups:
        assert hi > lo and (int(isz[lo:hi].sum()) <= chunk or hi - lo == 1)
    wp, wi = m.whole()
    assert np.array_equal(wp, pos) and np.array_equal(wi, isz)
    assert sum(m.run_bytes()) == int(pos[-1])


def test_a_failed_piece_is_raised_by_the_run_that_needs_it():
    rng = np.random.default_rng(5)
    eng = _Rounds(0)
    pos, isz = _table(rng, 3000)
    rest = Future()
    m = pipeline._Members(eng, 2_000_000, pos[:1001], isz[:1000], rest)
    known = len(m.run_bytes())
    assert known >= 1 and m.group(0) is not None                  # what the beginning holds is served without the rest
    rest.set_exception(pipeline.GciError(-5, "damaged member"))
    with pytest.raises(pipeline.GciError):
        m.group(known)
