"""The librccl test double's CROSS-PROCESS transport (tests/fake_nccl/fake_nccl.cpp, FAKE_NCCL_SHM=1) on its own, on the
CPU: the double is compiled with -DFAKE_NCCL_HOST_ONLY (a "device" copy is a memcpy) and driven from two and three REAL
processes through ctypes -- rendezvous by unique id, all-gather / all-reduce(sum, max), a grouped send / receive both
ways, ncclCommSplit; a count mismatch, a receive of the wrong size and a receive nobody sends to are errors, and a rank
that is KILLED in the middle of a collective makes the others return an error within seconds instead of hanging.  The GPU
form of the same library carries `python bench.py --gpus 2` on one device (tests/test_gpu_bench_multirank.py)."""
import os
import signal
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tests', 'fake_nccl', 'fake_nccl.cpp')
LIB = os.path.join(ROOT, 'build', 'fake_nccl', 'libfake_nccl_host.so')

RANK = r'''
import ctypes as C, os, sys, time
import numpy as np
lib = C.CDLL(sys.argv[1])
rank, world, idfile, mode = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
class UID(C.Structure):
    _fields_ = [('internal', C.c_char * 128)]
vp = C.c_void_p
lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, UID, C.c_int]
lib.ncclCommSplit.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp), vp]
lib.ncclAllGather.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, vp]
lib.ncclAllReduce.argtypes = [vp, vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
lib.ncclCommDestroy.argtypes = [vp]
F64, U64, SUM, MAX = 8, 5, 0, 2   # ncclFloat64, ncclUint64, ncclSum, ncclMax (rccl.h)
uid = UID()
if rank == 0:
    assert lib.ncclGetUniqueId(C.byref(uid)) == 0
    assert uid.internal.startswith(b'/fake-nccl-shm-'), uid.internal
    with open(idfile + '.tmp', 'wb') as f:
        f.write(bytes(uid))
    os.rename(idfile + '.tmp', idfile)
else:
    while not os.path.exists(idfile):
        time.sleep(0.01)
    C.memmove(C.byref(uid), open(idfile, 'rb').read(), 128)
def p(a):
    return a.ctypes.data_as(vp)
c = vp()
assert lib.ncclCommInitRank(C.byref(c), world, uid, rank) == 0
mine = np.arange(4, dtype=np.float64) + 10 * rank
buf = np.zeros(4 * world)
assert lib.ncclAllGather(p(mine), p(buf), 4, F64, c, None) == 0
assert np.array_equal(buf, np.concatenate([np.arange(4) + 10.0 * r for r in range(world)])), buf
# in place (the product's all-gathers): my block already lies in the receive buffer
buf2 = np.zeros(2 * world)
buf2[2 * rank:2 * rank + 2] = [rank, -rank]
assert lib.ncclAllGather(p(buf2[2 * rank:]), p(buf2), 2, F64, c, None) == 0
assert np.array_equal(buf2, np.array([[r, -r] for r in range(world)], dtype=np.float64).ravel()), buf2
red = np.array([1.5 + rank, -2.0 * rank])
assert lib.ncclAllReduce(p(red), p(red), 2, F64, SUM, c, None) == 0
assert np.array_equal(red, [sum(1.5 + r for r in range(world)), sum(-2.0 * r for r in range(world))]), red
mx = np.array([7 + rank], np.uint64)
assert lib.ncclAllReduce(p(mx), p(mx), 1, U64, MAX, c, None) == 0 and mx[0] == 7 + world - 1
if mode == 'killed':
    # rank 1 dies inside the next collective: the others must come back with an error, soon
    if rank == 1:
        os.kill(os.getpid(), 9)
    t0 = time.time()
    rc = lib.ncclAllReduce(p(red), p(red), 2, F64, SUM, c, None)
    assert rc != 0, 'a collective with a dead rank returned success'
    assert time.time() - t0 < 20.0
    print('rank %d: dead peer noticed after %.1f s' % (rank, time.time() - t0))
    sys.exit(0)
# a ring of grouped sends / receives: to the right, from the left -- and, at once, the other way round
right, left = (rank + 1) % world, (rank - 1) % world
g1, g2 = np.zeros(3), np.zeros(5)
lib.ncclGroupStart()
lib.ncclSend(p(mine), 3, F64, right, c, None)
lib.ncclRecv(p(g1), 3, F64, left, c, None)
if world > 2:   # (with two ranks left == right: one send and one receive per peer, like the product)
    lib.ncclSend(p(buf), 5, F64, left, c, None)
    lib.ncclRecv(p(g2), 5, F64, right, c, None)
assert lib.ncclGroupEnd() == 0
assert np.array_equal(g1, np.arange(3) + 10.0 * left), g1
if world > 2:
    assert np.array_equal(g2, buf[:5]), g2
# a second communicator split off the first (the product's overlap communicator), used right away
c2 = vp()
assert lib.ncclCommSplit(c, 0, rank, C.byref(c2), None) == 0
one = np.array([1.0])
assert lib.ncclAllReduce(p(one), p(one), 1, F64, SUM, c2, None) == 0 and one[0] == world
# ranks disagree on the count of a collective: an error on every rank, and the world stays usable
assert lib.ncclAllGather(p(mine), p(buf), 4 if rank == 0 else 3, F64, c, None) != 0
one[0] = 1.0
assert lib.ncclAllReduce(p(one), p(one), 1, F64, SUM, c, None) == 0 and one[0] == world
# a receive that expects another size than its send
if rank == 0:
    rc = lib.ncclSend(p(mine), 2, F64, 1, c, None)
elif rank == 1:
    rc = lib.ncclRecv(p(g1), 3, F64, 0, c, None)
    assert rc != 0
# a receive nobody sends to: times out (FAKE_NCCL_TIMEOUT_S)
if rank == 1:
    t0 = time.time()
    assert lib.ncclRecv(p(g1), 1, F64, 0, c, None) != 0
    assert time.time() - t0 < 30.0
assert lib.ncclCommDestroy(c2) == 0 and lib.ncclCommDestroy(c) == 0
print('rank %d ok' % rank)
'''


@pytest.fixture(scope='module')
def host_lib():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.run(['g++', '-O2', '-std=c++17', '-fPIC', '-shared', '-Wall', '-DFAKE_NCCL_HOST_ONLY', '-D__HIP_PLATFORM_AMD__',
                        '-I/opt/rocm/include', '-o', LIB, SRC, '-lpthread', '-lrt'], check=True)
    return LIB


def _spawn(lib, world, mode, tmp_path):
    idfile = str(tmp_path / ('uid_%s_%d' % (mode, world)))
    env = dict(os.environ, FAKE_NCCL_SHM='1', FAKE_NCCL_TIMEOUT_S='3', FAKE_NCCL_SLOT_MB='4')
    return [subprocess.Popen([sys.executable, '-c', RANK, lib, str(r), str(world), idfile, mode], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True) for r in range(world)]


def _leftovers():
    return [f for f in os.listdir('/dev/shm') if f.startswith('fake-nccl-shm-')] if os.path.isdir('/dev/shm') else []


@pytest.mark.parametrize('world', [2, 3])
def test_ranks_as_processes_meet_in_shared_memory(host_lib, world, tmp_path):
    before = set(_leftovers())
    procs = _spawn(host_lib, world, 'plain', tmp_path)
    outs = [p.communicate(timeout=120) for p in procs]
    for r, (p, (out, err)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and 'rank %d ok' % r in out, 'rank %d:\n%s\n%s' % (r, out[-1500:], err[-3000:])
    assert set(_leftovers()) <= before   # every segment's name was unlinked once its ranks had mapped it


def test_a_killed_rank_is_an_error_on_the_others_not_a_hang(host_lib, tmp_path):
    t0 = time.time()
    procs = _spawn(host_lib, 3, 'killed', tmp_path)
    outs = [p.communicate(timeout=120) for p in procs]
    assert procs[1].returncode == -signal.SIGKILL
    for r in (0, 2):
        assert procs[r].returncode == 0 and 'dead peer noticed' in outs[r][0], 'rank %d:\n%s\n%s' % (r, outs[r][0][-1500:], outs[r][1][-3000:])
    assert time.time() - t0 < 60.0
