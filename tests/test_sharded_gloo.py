"""world_size-2 CPU test (gloo): the particle-sharded resample algorithm -- local weights,
all-reduce(max), shard totals all-gather, offspring-CDF all-gather, state all-gather, per-slot
reassign -- gives bit-identical particles to the unsharded filter.  Compute is the oracle (no GPU
here); the exchange pattern is exactly the one libmcl_hip.so runs over RCCL."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _unsharded(lw, soa, u53, mode):
    from oracle import oracle as orc
    idx, ncum, q = orc.systematic_fixed(lw, mode, u53)
    lost, dupes = orc.lost_dupes(idx)
    out = soa.copy()
    orc.reassign(out, lost, dupes)
    return out, idx, ncum


def _worker(rank, world, port, n, seed, ret):
    import torch
    import torch.distributed as dist
    from oracle import oracle as orc
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(seed)
        lw_all = -0.5 * (rs.randn(n) * 2.5) ** 2 - 40.0
        soa_all = rs.randn(6, n)
        u53 = orc.u_to_u53(rs.random_sample())
        nl = n // world
        sl = slice(rank * nl, (rank + 1) * nl)
        lw, soa = lw_all[sl].copy(), np.ascontiguousarray(soa_all[:, sl])
        # C1: all-reduce(max) of the local max log-weight
        m = torch.tensor([float(np.max(lw))], dtype=torch.float64)  # exact, order-free
        dist.all_reduce(m, op=dist.ReduceOp.MAX)
        q, tot = orc.fixed_weights_shard(lw, 1, n, float(m[0]))
        # all-gather of the shard totals (u64 carried as int64 bit patterns)
        tl = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([np.uint64(tot).astype(np.int64)], dtype=torch.int64))
        totals = [int(np.int64(t[0]).astype(np.uint64)) for t in tl]
        T, off = sum(totals), sum(totals[:rank])
        ncum_loc = orc.systematic_ncum(q, u53, off, T, n)
        # C2: all-gather of the offspring CDF and of the pre-resample state
        gl = [torch.zeros(nl, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(gl, torch.from_numpy(ncum_loc.astype(np.int64)))
        ncum = np.concatenate([g.numpy() for g in gl]).astype(np.uint32)
        sg = [torch.zeros(6, nl, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(sg, torch.from_numpy(soa))
        state_glob = np.concatenate([s.numpy() for s in sg], axis=1)
        # every rank derives the same global lost/dupes lists and applies the slots it owns
        idx = orc.indices_from_ncum(ncum)
        lost, dupes = orc.lost_dupes(idx)
        mine = soa.copy()
        for l, d in zip(lost, dupes):
            if rank * nl <= l < (rank + 1) * nl:
                mine[:, l - rank * nl] = state_glob[:, d]
        ref, idx_ref, ncum_ref = _unsharded(lw_all, soa_all, u53, 1)
        ok = (np.array_equal(mine, ref[:, sl]) and np.array_equal(idx, idx_ref) and np.array_equal(ncum, ncum_ref))
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [4096, 20000])
def test_sharded_resample_world2_gloo(n):
    import torch.multiprocessing as mp
    world = 2
    port = _free_port()
    with mp.Manager() as mgr:
        ret = mgr.dict()
        mp.spawn(_worker, args=(world, port, n, 17, ret), nprocs=world, join=True)
        assert dict(ret) == {0: True, 1: True}
