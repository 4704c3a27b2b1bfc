"""The multi-rank branches of the product's HOST code with world > 1 on ONE GPU: W ranks as W threads of a child
process, librccl's entry points answered by tests/fake_nccl (LD_PRELOAD) -- the boxes of this pool have one device and
RCCL refuses two ranks on one, so mcl_comm_init_ex, the max / totals / hand-over-record collectives, exchange_dupes'
grouped ncclSend / ncclRecv with its pinned-word spin, the moments all-reduce, mcl_comm_selftest, shutdown and re-init
had only ever run with one rank (VERDICT r4 missing 1).  The double checks every count / type / peer the way RCCL relies
on them and moves the bytes; the sharded filter must equal the unsharded one bit for bit.  What this does NOT cover:
RCCL's own transport, stream-ordered asynchrony, timing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'fake_nccl', 'fake_nccl.cpp')
LIB = os.path.join(ROOT, 'build', 'fake_nccl', 'libfake_nccl.so')


@pytest.fixture(scope='module')
def fake_nccl():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(['/opt/rocm/bin/hipcc', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-o', LIB, SRC], check=True)
    return LIB


def _run(lib, *args, timeout_s=60, **env_extra):
    env = dict(os.environ, LD_PRELOAD=lib, FAKE_NCCL_TIMEOUT_S=str(timeout_s), **env_extra)
    env.pop('MCL_FORCE_COMM', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'fake_nccl', 'driver.py')] + [str(a) for a in args],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900)
    tail = (p.stdout.strip().split('\n') or [''])[-1]
    assert p.returncode == 0, 'driver failed:\n%s\n%s' % (p.stdout[-2000:], p.stderr[-4000:])
    res = json.loads(tail)
    assert res['ok'], res
    print(res)
    return res


@pytest.mark.parametrize('world,per_rank', [(2, 65536), (8, 16384), (3, 40001)])
def test_ranks_over_the_exchange_of_copies_equal_the_unsharded_filter(fake_nccl, world, per_rank):
    res = _run(fake_nccl, world, per_rank, 'p2p')
    assert res['states_sent'] > 0   # copies did cross rank borders through ncclSend / ncclRecv
    # the latency chain of a sharded step: TWO collectives -- the shards' records (maximum AND totals: one
    # latency where rounds 1-5 had two; the previous step's moments ride in them), the hand-over records -- and one group of
    # sends / receives; the moments' own all-reduce only when a reader comes before the next step
    assert max(res['collectives_per_step']) == 2 and max(res['p2p_groups_per_step']) <= 1, res
    assert res['tail_collectives'] == 6, res


def test_moments_all_reduced_after_every_step_when_asked_for(fake_nccl):
    """MCL_MOMENTS_RIDE=0: the step's moments are all-reduced at its end (rounds 1-5) instead of riding with the next step's
    records -- three collectives per step, the same filter bit for bit"""
    res = _run(fake_nccl, 3, 30000, 'p2p', MCL_MOMENTS_RIDE='0')
    assert max(res['collectives_per_step']) == 3 and res['tail_collectives'] == 9, res


def test_ranks_over_the_all_gather_exchange_with_the_overlap_communicator(fake_nccl):
    _run(fake_nccl, 4, 32768, 'allgather')


def test_ranks_with_the_visiting_order_and_the_landmark_step(fake_nccl):
    """shards above the visiting order's threshold (393 216), BASELINE config 5's fused step"""
    _run(fake_nccl, 2, 393216, 'p2p', 'landmarks')


def test_a_rank_that_stays_away_from_the_self_test_is_an_error_not_a_hang(fake_nccl):
    """mcl_comm_selftest with a peer missing returns MCL_ERR_COMM; abort + re-initialisation under a fresh id, then the
    filter runs as if nothing had happened"""
    _run(fake_nccl, 3, 20000, 'p2p', 'absent')


def test_the_double_itself_rejects_what_rccl_would_hang_on(fake_nccl):
    """tests/fake_nccl on its own (ctypes, two threads): matched operations move the right bytes; ranks that disagree on
    a collective's count, a receive that expects another size than its send, and a receive nobody sends to are errors."""
    code = r'''
import ctypes as C, threading, sys, os
import numpy as np
lib = C.CDLL(sys.argv[1])
hip = C.CDLL('libamdhip64.so')
class UID(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]
vp = C.c_void_p
lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, UID, C.c_int]
lib.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
lib.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
F64, U64, SUM, MAX = 8, 5, 0, 2   # ncclFloat64, ncclUint64, ncclSum, ncclMax (rccl.h)
uid = UID()
assert lib.ncclGetUniqueId(C.byref(uid)) == 0
res = [None, None]
def dev(a):
    p = vp()
    assert hip.hipMalloc(C.byref(p), a.nbytes) == 0
    assert hip.hipMemcpy(p, a.ctypes.data_as(vp), a.nbytes, 1) == 0
    return p
def host(p, n, dt):
    a = np.zeros(n, dt)
    assert hip.hipMemcpy(a.ctypes.data_as(vp), p, a.nbytes, 2) == 0
    return a
def rank(r):
    c = vp()
    assert lib.ncclCommInitRank(C.byref(c), 2, uid, r) == 0
    out = {}
    buf = dev(np.zeros(8))
    mine = dev(np.arange(4, dtype=np.float64) + 10 * r)
    out['ag'] = lib.ncclAllGather(mine, buf, 4, F64, c, None), host(buf, 8, np.float64)
    red = dev(np.array([1.5 + r, -2.0 * r]))
    out['ar'] = lib.ncclAllReduce(red, red, 2, F64, SUM, c, None), host(red, 2, np.float64)
    mx = dev(np.array([7 + r], np.uint64))
    out['mx'] = lib.ncclAllReduce(mx, mx, 1, U64, MAX, c, None), host(mx, 1, np.uint64)
    # grouped exchange both ways
    got = dev(np.zeros(3))
    lib.ncclGroupStart()
    lib.ncclSend(mine, 3, F64, 1 - r, c, None)
    lib.ncclRecv(got, 3, F64, 1 - r, c, None)
    out['p2p'] = lib.ncclGroupEnd(), host(got, 3, np.float64)
    # ranks disagree on the count of a collective
    out['bad_count'] = lib.ncclAllGather(mine, buf, 4 if r == 0 else 3, F64, c, None)
    # a receive that expects another size than its send
    if r == 0:
        out['bad_p2p'] = lib.ncclSend(mine, 2, F64, 1, c, None)
    else:
        out['bad_p2p'] = lib.ncclRecv(got, 3, F64, 0, c, None)
    # a receive nobody sends to (rank 1 only): times out
    out['lonely'] = lib.ncclRecv(got, 1, F64, 0, c, None) if r == 1 else 0
    res[r] = out
th = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
[t.start() for t in th]; [t.join() for t in th]
for r in range(2):
    o = res[r]
    assert o['ag'][0] == 0 and np.array_equal(o['ag'][1], [0, 1, 2, 3, 10, 11, 12, 13]), o['ag']
    assert o['ar'][0] == 0 and np.array_equal(o['ar'][1], [4.0, -2.0]), o['ar']
    assert o['mx'][0] == 0 and o['mx'][1][0] == 8, o['mx']
    assert o['p2p'][0] == 0 and np.array_equal(o['p2p'][1], np.arange(3) + 10 * (1 - r)), o['p2p']
    assert o['bad_count'] != 0, o['bad_count']
assert res[1]['bad_p2p'] != 0
assert res[1]['lonely'] != 0
print('double ok')
'''
    env = dict(os.environ, FAKE_NCCL_TIMEOUT_S='3')
    p = subprocess.run([sys.executable, '-c', code, fake_nccl], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       universal_newlines=True, timeout=300)
    assert p.returncode == 0 and 'double ok' in p.stdout, p.stdout[-2000:] + p.stderr[-3000:]
