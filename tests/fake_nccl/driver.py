#!/usr/bin/env python3
"""Child process of tests/test_gpu_multirank_shim.py: W ranks of the sharded filter as W THREADS on one GPU, talking
through tests/fake_nccl (LD_PRELOAD in front of librccl), against the unsharded filter -- bit for bit.

usage: driver.py WORLD PARTICLES_PER_RANK EXCHANGE(p2p|allgather) [landmarks]"""
import ctypes
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from smarc_navigation_amd import engine as eng, synth  # noqa: E402

COV = dict(init_cov=[2.0, 2.0, 0.0, 0.0, 0.0, 0.05], process_cov=[1e-4, 1e-4, 0.0, 0.0, 0.0, 1e-6],
           resample_cov=[1e-3, 1e-3, 0.0, 0.0, 0.0, 1e-5])
SIGMA, R_MAX, B = 0.2, 100.0, 128


def main():
    W, NS, exchange = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    with_lm = len(sys.argv) > 4 and sys.argv[4] == 'landmarks'
    absent = len(sys.argv) > 4 and sys.argv[4] == 'absent'   # rank W-1 stays away from the start-up self-test
    N = W * NS
    os.environ['MCL_EXCHANGE'] = exchange
    # the preload must really be in front of librccl
    assert ctypes.CDLL(None).fake_nccl_present() == 1
    origin = (-64.0, -128.0)
    z = synth.bathymetry_grid(256, 256, 1.0, origin, seed=3)
    verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    steps = 4
    stream = synth.odom_stream(steps + 5)
    ba = synth.beam_angles(B)
    rs = np.random.RandomState(4)
    ranges = (20.0 + 2.0 * rs.rand(steps + 5, B)).astype(np.float32)
    lm = synth.landmark_map(1024, (-60.0, -120.0, 180.0, 120.0))
    dets = []
    for k in range(steps + 5):
        t = stream['truth'][k]
        T = synth.rigid_matrix(*t)
        near = lm[np.argsort(np.sum((lm[:, :2] - t[:2]) ** 2, axis=1))[:8]]
        dets.append((near - T[:3, 3]).dot(T[:3, :3]) + 0.05 * rs.randn(8, 3))

    def step(e, k):
        od = (stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
        if with_lm:
            e.step_mbes_landmarks(*od, ranges[k], ba, SIGMA, R_MAX, dets[k], 0.3, k=2, gate=11.345)
        else:
            e.step_mbes(*od, ranges[k], ba, SIGMA, R_MAX)

    # ---- the unsharded filter
    one = eng.Engine(N, seed=5, **COV)
    one.set_map_mesh(verts, tris)
    one.set_landmarks(lm)
    one.init_particles()
    ref = []
    for k in range(steps):
        step(one, k)
        ref.append(dict(lw=one.get_log_weights(), idx=one.last_indices(), st=one.get_particles(), mc=one.last_mean_cov()))
    # (a separate-call round at the end: predict, update, resample, mean / cov with their own collectives)
    od = (stream['v'][steps], stream['wz'][steps], stream['q'][steps], stream['z'][steps], stream['dt'])
    one.predict(*od)
    one.update_mbes(ranges[steps], ba, SIGMA, R_MAX)
    one.resample()
    ref_sep = dict(idx=one.last_indices(), st=one.get_particles(), mc=one.mean_cov())
    # three fused steps back to back, nothing read in between: in the sharded filter the moments of a step travel with the
    # NEXT step's records (no all-reduce of their own) and only the last one is completed by the reader's flush
    for k in range(steps + 1, steps + 4):
        step(one, k)
    ref_tail = dict(hist=one.mean_history(3), st=one.get_particles(), mc=one.last_mean_cov())
    one.close()

    # ---- W ranks, one thread each
    uid = [eng.comm_unique_id(), eng.comm_unique_id(), eng.comm_unique_id()]
    gate = threading.Barrier(W)
    out = [None] * W
    err = [None] * W

    def counters():
        out = (ctypes.c_ulonglong * 3)()
        ctypes.CDLL(None).fake_nccl_counters(out)
        return list(out)

    def rank_main(r):
        try:
            e = eng.Engine(NS, rank=r, world=W, n_global=N, global_offset=r * NS, seed=5, **COV)
            e.set_map_mesh(verts, tris)
            e.set_landmarks(lm)
            e.comm_init(uid[0])
            res = dict(ranks=e.comm_ranks(), steps=[])
            if absent:
                # a peer that never shows up: the others get MCL_ERR_COMM back (not a hang), abort, and everybody starts
                # again under a fresh id
                gate.wait()
                if r == 0:
                    ctypes.CDLL(None).fake_nccl_set_timeout(3)   # (only the self-test below waits in vain)
                gate.wait()
                if r != W - 1:
                    try:
                        e.comm_selftest(20000)
                        raise AssertionError('self-test passed without rank %d' % (W - 1))
                    except eng.MclError as ex:
                        assert ex.status == -6 and 'comm_selftest' in str(ex), (ex.status, str(ex))   # MCL_ERR_COMM
                        res['selftest_error'] = str(ex)
                gate.wait()
                if r == 0:
                    ctypes.CDLL(None).fake_nccl_set_timeout(0)
                e.comm_shutdown(abort=True)
                gate.wait()
                e.comm_init(uid[2])
                res['ranks'] = e.comm_ranks()
            e.comm_selftest(20000)
            e.init_particles()
            for k in range(steps):
                if k == 2:
                    # tear the communicators down and build new ones under a fresh id: the filter goes on
                    e.sync()
                    e.comm_shutdown()
                    e.comm_init(uid[1])
                c0 = counters() if r == 0 else None
                step(e, k)
                if r == 0:
                    # the exchange's budget: a fused step asks rank 0 for at most THREE collectives -- the
                    # shards' records (maximum and totals in one), the hand-over records, the moments -- and one group of
                    # point-to-point operations; the all-gather scheme: records, CDF (+ state when it did not travel under
                    # the update), moments
                    e.sync()
                    c1 = counters()
                    res.setdefault('collectives_per_step', []).append(c1[0] - c0[0])
                    res.setdefault('p2p_groups_per_step', []).append(c1[1] - c0[1])
                res['steps'].append(dict(lw=e.get_log_weights(), idx=e.last_indices(), st=e.get_particles(),
                                         mc=e.last_mean_cov()))
            res['ops'] = e.exchange_ops()
            res['stats'] = e.exchange_stats()
            e.predict(*od)
            e.update_mbes(ranges[steps], ba, SIGMA, R_MAX)
            e.resample()
            res['sep'] = dict(idx=e.last_indices(), st=e.get_particles(), mc=e.mean_cov())
            c0 = counters() if r == 0 else None
            for k in range(steps + 1, steps + 4):
                step(e, k)
            e.sync()
            if r == 0:
                c1 = counters()
                res['tail_collectives'] = c1[0] - c0[0]
            res['tail'] = dict(hist=e.mean_history(3), st=e.get_particles(), mc=e.last_mean_cov())
            e.sync()
            e.comm_shutdown()
            e.close()
            out[r] = res
        except BaseException as ex:  # noqa: B902 -- reported by the main thread
            err[r] = '%s: %s' % (type(ex).__name__, ex)

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if any(err):
        print(json.dumps(dict(ok=False, errors=err)))
        return 1
    # ---- compare
    for r in range(W):
        assert out[r]['ranks'][0] == W, out[r]['ranks']
        if exchange == 'allgather':
            assert out[r]['ranks'][1], 'the overlap communicator (ncclCommSplit) is missing'
    for k in range(steps):
        lw = np.concatenate([out[r]['steps'][k]['lw'] for r in range(W)])
        idx = np.concatenate([out[r]['steps'][k]['idx'] for r in range(W)])
        st = np.concatenate([out[r]['steps'][k]['st'] for r in range(W)], axis=1)
        assert np.array_equal(lw, ref[k]['lw']), ('lw', k)
        assert np.array_equal(idx, ref[k]['idx']), ('idx', k)
        assert np.array_equal(st, ref[k]['st']), ('state', k)
        for r in range(W):   # every rank holds the global moments
            np.testing.assert_allclose(out[r]['steps'][k]['mc'][0], ref[k]['mc'][0], rtol=0, atol=1e-10)
            np.testing.assert_allclose(out[r]['steps'][k]['mc'][2], ref[k]['mc'][2], rtol=1e-8, atol=1e-12)
    assert np.array_equal(np.concatenate([out[r]['sep']['idx'] for r in range(W)]), ref_sep['idx'])
    assert np.array_equal(np.concatenate([out[r]['sep']['st'] for r in range(W)], axis=1), ref_sep['st'])
    for r in range(W):
        np.testing.assert_allclose(out[r]['sep']['mc'][0], ref_sep['mc'][0], rtol=0, atol=1e-10)
    # the back-to-back steps: every rank holds the unsharded filter's history of the three, to the moments' usual tolerance
    assert np.array_equal(np.concatenate([out[r]['tail']['st'] for r in range(W)], axis=1), ref_tail['st'])
    for r in range(W):
        np.testing.assert_allclose(out[r]['tail']['hist'], ref_tail['hist'], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out[r]['tail']['mc'][0], ref_tail['mc'][0], rtol=0, atol=1e-10)
        np.testing.assert_allclose(out[r]['tail']['mc'][2], ref_tail['mc'][2], rtol=1e-8, atol=1e-12)
    ops = [out[r]['ops'] for r in range(W)]
    sent = sum(out[r]['stats'][0] for r in range(W))
    lost = sum(out[r]['stats'][1] for r in range(W))
    if exchange == 'p2p':
        for o, rounds in ops:
            assert rounds == steps and o <= 2 * (W - 1) * rounds, ops
        assert 0 < sent < lost, (sent, lost)
    cps, gps = out[0]['collectives_per_step'], out[0]['p2p_groups_per_step']
    if exchange == 'p2p':
        # records (maximum + totals + the previous step's moments), hand-over records; the moments' own all-reduce only
        # when a reader asks before the next step -- three steps back to back: 2 collectives each
        ride = os.environ.get('MCL_MOMENTS_RIDE') != '0'   # (0: the all-reduce of the moments after every step, as in rounds 1-5)
        assert max(cps) <= (2 if ride else 3) and max(gps) <= 1, (cps, gps)
        assert out[0]['tail_collectives'] <= (6 if ride else 9), out[0]['tail_collectives']
    print(json.dumps(dict(ok=True, world=W, per_rank=NS, exchange=exchange, landmarks=with_lm, p2p_ops=[o for o, _ in ops],
                          states_sent=sent, lost_slots=lost, collectives_per_step=cps, p2p_groups_per_step=gps,
                          tail_collectives=out[0]['tail_collectives'])))
    return 0


if __name__ == '__main__':
    sys.exit(main())
