"""The transfer plan of the O(n)-per-rank resample exchange (mcl_exchange_plan: pure host arithmetic, the function the
library sizes its ncclSend / ncclRecv calls with), property-tested without a GPU for random worlds:
every lost slot is filled exactly once, every surplus copy is used exactly once, what q sends r is what r expects
from q (same count, consistent global positions), and the counts must add up."""
import ctypes as C

import numpy as np
import pytest


def _plan(lib, lost, surplus, rank):
    w = len(lost)
    L, S = np.asarray(lost, np.uint32), np.asarray(surplus, np.uint32)
    out = [np.zeros(w, np.uint32) for _ in range(4)]
    rc = lib.mcl_exchange_plan(w, L.ctypes.data, S.ctypes.data, rank, *[o.ctypes.data for o in out])
    return rc, out


@pytest.mark.parametrize('seed', range(40))
def test_exchange_plan_covers_every_lost_slot_exactly_once(seed):
    from smarc_navigation_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(seed)
    w = int(rs.choice([1, 2, 3, 4, 8, 16]))
    n = int(rs.choice([1, 7, 1000, 524288]))
    kind = seed % 4
    if kind == 0:     # balanced: surplus ~ lost per shard
        lost = rs.randint(0, n + 1, size=w)
        surplus = lost.copy()
        rs.shuffle(surplus)
    elif kind == 1:   # all the weight in one shard
        lost = rs.randint(0, n + 1, size=w)
        surplus = np.zeros(w, np.int64)
        surplus[rs.randint(w)] = lost.sum()
    elif kind == 2:   # nothing lost
        lost = np.zeros(w, np.int64)
        surplus = np.zeros(w, np.int64)
    else:             # random split of the same total
        lost = rs.randint(0, n + 1, size=w)
        cuts = np.sort(rs.randint(0, lost.sum() + 1, size=w - 1)) if w > 1 else np.array([], int)
        surplus = np.diff(np.concatenate([[0], cuts, [lost.sum()]]))
    Lpre, Spre = np.concatenate([[0], np.cumsum(lost)]), np.concatenate([[0], np.cumsum(surplus)])
    plans = []
    for q in range(w):
        rc, p = _plan(lib, lost, surplus, q)
        assert rc == 0
        plans.append(p)
    for q in range(w):
        so, sc, ro, rc_ = plans[q]
        # everything a shard holds goes somewhere, everything it needs comes from somewhere, ranges tile without gaps
        assert int(sc.sum()) == surplus[q] and int(rc_.sum()) == lost[q]
        pos = 0
        for r in range(w):
            if sc[r]:
                assert so[r] == pos
                pos += int(sc[r])
        pos = 0
        for r in range(w):
            if rc_[r]:
                assert ro[r] == pos
                pos += int(rc_[r])
        for r in range(w):
            # what q sends r is what r expects from q: same count, same global dupes positions
            assert sc[r] == plans[r][3][q]
            if sc[r]:
                assert Spre[q] + so[r] == Lpre[r] + plans[r][2][q]
        # ONE contiguous block per peer: the library issues one ncclSend per non-empty send range and one ncclRecv per
        # non-empty receive range (a surplus copy travels as one record, mcl_resample.h: k_pack_dupes), so an exchange is at
        # most 2 (world - 1) point-to-point operations per rank -- and the blocks of successive peers are adjacent and in
        # rank order, which is what makes each of them contiguous in the packed list (VERDICT r4 next 2a;
        # mcl_exchange_ops counts the operations really issued, asserted on the GPU in tests/test_gpu_config45.py)
        ops = sum(1 for r in range(w) if r != q and sc[r]) + sum(1 for r in range(w) if r != q and rc_[r])
        assert ops <= 2 * (w - 1)
        peers = [r for r in range(w) if sc[r]]
        assert peers == sorted(peers) and all(so[b] == so[a] + sc[a] for a, b in zip(peers, peers[1:]))


def test_exchange_plan_rejects_counts_that_do_not_add_up():
    from smarc_navigation_amd import _lib
    lib = _lib.load()
    rc, _ = _plan(lib, [3, 4], [3, 5], 0)
    assert rc == -1
    rc, _ = _plan(lib, [3, 4], [4, 3], 2)   # rank out of range
    assert rc == -1
