#!/usr/bin/env python3
"""bench.py -- filter steps/sec of the MCL hot path (BASELINE.json metric).

One "step" = predict (IMU+DVL motion model + process noise) -> MBES update (per-particle,
per-beam ray-cast + Gaussian log-likelihood) -> weight normalisation -> systematic resample
(+ keep/lost/dupes reassign + resampling noise) -> mean/covariance, on synthetic streams
(SURVEY.md 8(d)), all inputs resident in HBM except the per-ping 512 ranges (2 KiB).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--map mesh|grid|mesh-general|mesh-tin]
                    [--particles P] [--scaling weak|strong] [--total-particles T]

N > 1: one rank per GPU over RCCL.  Either an external launcher provides RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* (python -m torch.distributed.run ... bench.py --gpus N), or -- when WORLD_SIZE is
unset -- this process spawns the N ranks itself (launch(): plain child processes created BEFORE anything
here touches the GPU; the parent never imports torch or the engine) and relays rank 0's JSON line.
A rank count that differs from --gpus is an error (exit 2), a rank that fails or a run that exceeds
--launch-timeout ends every rank and exits non-zero: the bench never hangs and never silently measures
fewer GPUs than asked.  Particles shard by contiguous global id; the data-path collectives are native
RCCL calls inside libmcl_hip.so, torch.distributed (gloo) carries only the RCCL unique id and the timing
barrier.  `rccl_ranks` in the output is an ncclAllReduce(sum) of 1 over the data-path communicator.

--scaling weak (default): --particles per GPU (1 048 576: the metric's configuration on every GPU).
--scaling strong: --total-particles (default 4 194 304 = BASELINE config 4) split over the GPUs.
`value` counts 1 M-particle filter steps per second over the whole job (steps/s x total particles / 2^20),
so at N = 1 with the default workload it is plain steps/s of the metric; `steps_per_s` and
`ms_per_step` are the raw figures of the run.
"""
import argparse
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
METRIC = 'filter steps/sec at 1M particles x 512 MBES beams; pose RMSE vs ref'
MIN_TIMED_STEPS = 200   # wall-clock over >= 200 steps, median + p95
SIGMA, R_MAX = 0.2, 100.0
COV = dict(init_cov=[2.0, 2.0, 0.0, 0.0, 0.0, 0.05], process_cov=[1e-4, 1e-4, 0.0, 0.0, 0.0, 1e-6],
           resample_cov=[1e-3, 1e-3, 0.0, 0.0, 0.0, 1e-5])


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--particles', type=int, default=1048576, help='particles per GPU (weak scaling)')
    ap.add_argument('--total-particles', type=int, default=4194304, help='total particles (strong scaling)')
    ap.add_argument('--scaling', default='weak', choices=['weak', 'strong'])
    ap.add_argument('--beams', type=int, default=512)
    ap.add_argument('--map', default='mesh', choices=['grid', 'mesh', 'mesh-general', 'mesh-adjacency', 'mesh-tin', 'mesh-tin-shuffled', 'mesh-soup'])
    ap.add_argument('--mesh-general', action='store_true', help='same as --map mesh-general')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the extra workload legs (N = 1 runs them by default)')
    ap.add_argument('--only-main', action='store_true', help='same as --no-extra --no-cpu-baseline (profiling runs)')
    ap.add_argument('--cpu-particles', type=int, default=0, help='oracle sample size (0 = auto)')
    ap.add_argument('--rmse-particles', type=int, default=1048576,
                    help='particles of the GPU-vs-oracle trajectory comparison (default: the metric\'s own size, ~100 s of host '
                         'time on 16 threads; 0 = the cpu_baseline sample)')
    ap.add_argument('--no-overlap', action='store_true', help='in-line state all-gather (no second communicator)')
    ap.add_argument('--launch-timeout', type=float, default=1500.0, help='self-spawned ranks: wall-clock limit, seconds')
    # launcher self-test (CPU, gloo): rendezvous + barrier + all-reduce only, no engine, no GPU
    ap.add_argument('--temper', action='store_true', help='experiments: the main leg with the likelihood tempered by 1 / beams (extra.filter_tempered\'s filter)')
    ap.add_argument('--dry-run', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--dry-run-fail-rank', type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument('--dry-run-hang-rank', type=int, default=-1, help=argparse.SUPPRESS)
    # tests/test_gpu_bench_multirank.py: rank R of the REAL worker dies (SIGKILL) before step K of the first timed block
    ap.add_argument('--test-kill-rank', type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument('--test-kill-step', type=int, default=0, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.mesh_general:
        a.map = 'mesh-general'
    if a.only_main:
        a.no_extra = a.no_cpu_baseline = True
    return a


# ------------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(a, argv):
    """Spawn a.gpus ranks of this script (fresh processes; nothing in this parent has touched the GPU)
    and wait for them.  Rank 0 inherits stdout (the ONE JSON line); the other ranks' stdout goes to
    stderr.  Returns the exit code: 0 only if every rank exited 0."""
    port = _free_port()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), MCL_BENCH_SPAWNED='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC only on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else sys.stderr, start_new_session=True))

    def end_all():
        for p in procs:  # exact process groups we created, never a pattern
            if p.poll() is None:
                try:
                    os.killpg(p.pid, 15)
                except OSError:
                    pass
        t_end = time.time() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(p.pid, 9)
                except OSError:
                    pass

    deadline = time.time() + a.launch_timeout
    try:
        while True:
            codes = [p.poll() for p in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                sys.stderr.write('bench launcher: rank %d exited with code %d; ending the other ranks\n' % bad[0])
                end_all()
                return bad[0][1] if 0 < bad[0][1] < 256 else 1
            if all(c == 0 for c in codes):
                return 0
            if time.time() > deadline:
                sys.stderr.write('bench launcher: ranks still running after %.0f s; ending them\n' % a.launch_timeout)
                end_all()
                return 124
            time.sleep(0.05)
    except KeyboardInterrupt:
        end_all()
        return 130


# ------------------------------------------------------------------------------------------ workloads
def build_map(kind):
    from smarc_navigation_amd import synth
    if kind == 'grid':
        origin = (-64.0, -256.0)
        z = synth.bathymetry_grid(512, 512, 1.0, origin, seed=3)
        return dict(kind='grid', z=z, origin=origin, res=1.0, bytes=z.nbytes, bytes_what='512 x 512 fp32 heights',
                    desc='512x512 fp32 height grid, 1 m cells')
    origin = (-64.0, -354.0)
    z = synth.bathymetry_grid(708, 708, 1.0, origin, seed=3)
    if kind in ('mesh-tin', 'mesh-tin-shuffled', 'mesh-soup'):
        verts, tris = synth.mesh_tin(z, 1.0, origin, seed=7)
        desc = '%d-triangle irregular TIN (708x708 nodes jittered in xy, random diagonals)' % tris.shape[0]
        if kind == 'mesh-soup':
            desc += ', cast as an arbitrary triangle soup (MCL_MESH_GENERAL: no height-field assumption)'
        if kind == 'mesh-tin-shuffled':
            verts, tris = synth.mesh_shuffle(verts, tris, seed=9)
            desc += ', handed over in RANDOM order (vertices renumbered, triangles permuted, corners rotated, every second winding reversed)'

    else:
        verts, tris = synth.mesh_from_grid(z, 1.0, origin)
        desc = '%d-triangle mesh (708x708 height field triangulated)' % tris.shape[0]
        if kind == 'mesh-general':
            desc += ', cast as an arbitrary triangle soup (MCL_MESH_GENERAL: no height-field assumption)'
        if kind == 'mesh-adjacency':
            desc += ', swept by triangle adjacency (MCL_MESH_UNSTRUCTURED: no structured-mesh detection)'
    # the bytes the chosen path is REQUIRED to read per ping (M), not the size of the input arrays:
    # structured mesh -> the node heights; adjacency sweep -> 32 B per triangle + 16 B per vertex; triangle-record
    # traversal -> 48 B per (cell, triangle) record (at least one per triangle) + 8 B per cell
    nt, nv = tris.shape[0], verts.shape[0]
    if kind == 'mesh':
        mb, what = 4 * z.size, '708 x 708 fp32 node heights (structured mesh: no triangle records are read)'
    elif kind in ('mesh-tin', 'mesh-tin-shuffled', 'mesh-adjacency'):
        mb, what = 96 * nt, 'three 32 B half-edge records per triangle (adjacency walk)'
    else:
        mb, what = 16 * nt + 16 * nv + 8 * (nt // 2), '>= 16 B vertex-id record per triangle + 16 B per vertex + 8 B per cell (fan slice)'
    return dict(kind=kind, z=z, origin=origin, res=1.0, verts=verts, tris=tris, bytes=mb, bytes_what=what, desc=desc)


def punch_hole(m, x, y, radius=0.8):
    """The mesh without the triangles whose centroid lies within `radius` of (x, y): a data gap (mesh_build: border edges
    inside the bounding box; a slice that reaches one ends the adjacency walk and the particle is handed over)."""
    import numpy as np
    c = m['verts'][m['tris'].astype(np.int64)].mean(axis=1)
    gone = np.hypot(c[:, 0] - x, c[:, 1] - y) < radius
    return dict(m, tris=np.ascontiguousarray(m['tris'][~gone]),
                desc=m['desc'] + ', %d triangles around (%g, %g) missing' % (int(gone.sum()), x, y))


def punch_gaps(m, tile=6.0, seed=11):
    """The mesh with one data gap per tile x tile metres of map, at a random place in its tile, 0.8 .. 5 m across (the outer
    ring of tiles stays intact: a gap that reaches the outline is a ragged border, not a hole) -- a survey with holes
    everywhere: at tile = 6 m about 13 % of the triangles are missing and every ping looks into a few of the gaps."""
    import numpy as np
    rs = np.random.RandomState(seed)
    c = m['verts'][m['tris'].astype(np.int64)].mean(axis=1)
    x0, y0 = m['origin']
    t = (np.floor((c[:, 0] - x0) / tile).astype(np.int64), np.floor((c[:, 1] - y0) / tile).astype(np.int64))
    nt = int(t[0].max()) + 1, int(t[1].max()) + 1
    cx = x0 + tile * (np.arange(nt[0])[:, None] + 0.3 + 0.4 * rs.rand(nt[0], nt[1]))
    cy = y0 + tile * (np.arange(nt[1])[None, :] + 0.3 + 0.4 * rs.rand(nt[0], nt[1]))
    rad = 0.4 + 2.1 * rs.rand(nt[0], nt[1]) ** 2
    gone = np.hypot(c[:, 0] - cx[t], c[:, 1] - cy[t]) < rad[t]
    gone &= (t[0] > 0) & (t[1] > 0) & (t[0] < nt[0] - 1) & (t[1] < nt[1] - 1)
    return dict(m, tris=np.ascontiguousarray(m['tris'][~gone]),
                desc=m['desc'] + ', %d triangles missing in %d gaps' % (int(gone.sum()), (nt[0] - 2) * (nt[1] - 2)))


def attach_map(e, m):
    if m['kind'] == 'grid':
        e.set_map_grid(m['z'], m['origin'], m['res'])
    else:
        e.set_map_mesh(m['verts'], m['tris'], general=(m['kind'] in ('mesh-general', 'mesh-soup')),
                       unstructured=(m['kind'] == 'mesh-adjacency'))


def make_ranges(engine_mod, m, truth, beam_angles, sigma, r_max, device=0, seed=4, m2o=None):
    """Synthetic pings: expected ranges at the truth poses (one-particle engine on the GPU) + noise."""
    import numpy as np
    e = engine_mod.Engine(1, rng_mode=engine_mod.RNG_REPLAY, device=device, m2o=m2o)  # this rank's own GPU
    attach_map(e, m)
    rs = np.random.RandomState(seed)
    out = np.zeros((len(truth), beam_angles.size), np.float32)
    for k in range(len(truth)):
        e.set_particles(truth[k][:, None].copy())
        out[k] = e.mbes_expected(0, 1, beam_angles, r_max)[0] + sigma * rs.randn(beam_angles.size)
    e.close()
    return out


def host_cores():
    """Host threads this process may really use: the affinity mask, capped by the cgroup CPU quota
    (a container with `cpu.max = 1600000 100000` gets 16 CPUs of time however many it can see)."""
    cores = len(os.sched_getaffinity(0))
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()[:2]
        if quota != 'max':
            cores = max(1, min(cores, int(-(-int(quota) // int(period)))))
    except (IOError, ValueError):
        pass
    return cores


def source_hash():
    """Hash of the kernel sources the PMC figures in profiles/ were taken at: a stale profile is never
    attached to a bench line of different kernels."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'smarc_navigation_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.h', '.hip')):
            with open(os.path.join(d, name), 'rb') as f:
                h.update(name.encode() + b'\0' + f.read())
    return h.hexdigest()[:16]


def pctl(v, q):
    import numpy as np
    return float(np.percentile(np.asarray(v, dtype=np.float64), q))


def cpu_baseline(m, stream, ranges, beam_angles, cov, n_full, n_sample, threads, budget_s, max_steps=20):
    """The oracle (C restatement; its particle loops run on `threads` host threads) on a bounded
    sample of the same workload."""
    import numpy as np
    from oracle import oracle as orc
    threads = orc.set_threads(threads)
    amap = orc.Grid(m['z'], m['origin'], m['res']) if m['kind'] == 'grid' else orc.Mesh(m['verts'], m['tris'])
    n = n_sample
    soa = np.zeros((6, n))
    orc.add_noise(soa, cov['init_cov'], orc.native_normals(n, 0, 5, 0, 0))
    t0 = time.perf_counter()
    steps = 0
    sq_err = 0.0
    means = []
    while True:
        k = steps
        orc.predict(soa, stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                    cov['process_cov'], orc.native_normals(n, 0, 5, 1, k))
        lw, _ = orc.mbes_update(soa, np.identity(4), [0] * 6, amap, beam_angles, ranges[k], SIGMA, R_MAX,
                                want_expected=False)
        idx, ncum, q = orc.systematic_fixed(lw, 1, orc.native_u53(5, k))
        lost, dupes = orc.lost_dupes(idx)
        orc.reassign(soa, lost, dupes)
        orc.add_noise(soa, cov['resample_cov'], orc.native_normals(n, 0, 5, 2, k))
        m6, _, _ = orc.mean_cov(soa)
        sq_err += (m6[0] - stream['truth'][k][0]) ** 2 + (m6[1] - stream['truth'][k][1]) ** 2
        means.append([m6[0], m6[1]])
        steps += 1
        el = time.perf_counter() - t0
        if el > budget_s or steps >= min(len(ranges), max_steps):
            break
    per_step = el / steps
    return dict(value=(n / float(n_full)) / per_step, unit='steps/s', cores=threads, kind='port',
                pose_rmse_m=round(float(np.sqrt(sq_err / steps)), 4), _means=np.array(means), _n=n,
                sample='%d particles x %d beams x %d steps of the same stream+map on %d host thread(s) (%.2f s/step), '
                       'scaled linearly to %d particles' % (n, beam_angles.size, steps, threads, per_step, n_full))


# BASELINE.md section 2 (Row A of SURVEY 8(d)(ii)): the reference's own Python, measured in the BUILD container through the
# oracle harness (oracle/ref_harness: the reference's modules imported behind stubs).  It never travels to the GPU box in
# any form, so these are constants with their hardware label, not measurements of this run.
REFERENCE_PYTHON = {
    'hardware': 'build container: 8 vCPU x86-64, Python 3.10.12, numpy 2.2.6, scipy 1.15.3, single thread (the reference has no parallelism)',
    'measured_by': 'oracle/ref_harness (the reference modules imported unmodified); BASELINE.md section 2',
    'motion_pred_us_per_particle': 45.0, 'motion_pred_ref': 'auv_particle_filter/scripts/auv_particle.py:38-70',
    'compute_weight_us_per_particle': 146.0, 'compute_weight_ref': 'auv_particle_filter/scripts/auv_particle.py:100-106',
    'predict_only_n128_ms_per_step': 7.0,
    'gps_update_residual_resample_reassign_noise_n128_ms': 25.0,
    'systematic_resample_n65536_ms': 37.0, 'residual_resample_n65536_ms': 25.0, 'resample_ref': 'auv_particle_filter/scripts/resampling.py:135-168 / :27-76',
    'systematic_resample_n1048576_s': 1.09, 'residual_resample_n1048576_s': 0.97,
    'full_step_n1048576': 'EXTRAPOLATED, not runnable: ~45 s predict + ~146 s GPS update + the O(N^2) keep / lost / dupes lists '
                          '(auv_pf.py:183-187); the reference has no MBES update at all',
}


def cpu_leg_baseline(engine_mod, m, P, B, device, landmarks=None, budget_s=4.0):
    """the oracle (C restatement) on this box's host cores for an extra leg's configuration -- one thread
    and all granted threads, each on a bounded sample (<= budget_s of CPU work per measurement, at least one step),
    scaled linearly to the leg's particle count.  Steps/s of the LEG'S OWN configuration (not 1 M-normalised)."""
    import numpy as np
    from oracle import oracle as orc
    from smarc_navigation_amd import synth
    steps = 3
    stream = synth.odom_stream(steps)
    ba = synth.beam_angles(B)
    ranges = make_ranges(engine_mod, m, stream['truth'], ba, SIGMA, R_MAX, device=device)
    amap = orc.Grid(m['z'], m['origin'], m['res']) if m['kind'] == 'grid' else orc.Mesh(m['verts'], m['tris'])
    dets = None
    if landmarks is not None:
        lm_xyz, n_det = landmarks
        rs = np.random.RandomState(8)
        dets = []
        for k in range(steps):
            t = stream['truth'][k]
            T = synth.rigid_matrix(*t)
            near = lm_xyz[np.argsort(np.sum((lm_xyz[:, :2] - t[:2]) ** 2, axis=1))[:n_det]]
            dets.append((near - T[:3, 3]).dot(T[:3, :3]) + 0.05 * rs.randn(n_det, 3))
    out = {'unit': 'steps/s', 'kind': 'port', 'of': '%d particles x %d beams' % (P, B)}
    cores = host_cores()
    for label, threads, n in (('1thread', 1, min(P, 2048)), ('all', cores, min(P, max(2048 * min(cores, 32), 65536)))):
        threads = orc.set_threads(threads)
        soa = np.zeros((6, n))
        orc.add_noise(soa, COV['init_cov'], orc.native_normals(n, 0, 5, 0, 0))
        t0 = time.perf_counter()
        done = 0
        for k in range(steps):
            orc.predict(soa, stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                        COV['process_cov'], orc.native_normals(n, 0, 5, 1, k))
            lw, _ = orc.mbes_update(soa, np.identity(4), [0] * 6, amap, ba, ranges[k], SIGMA, R_MAX, want_expected=False)
            if dets is not None:
                lw = lw + orc.landmark_update(soa, np.identity(4), [0] * 6, landmarks[0], dets[k], 0.3, k=4, gate=11.345)
            idx, _, _ = orc.systematic_fixed(lw, 1, orc.native_u53(5, k))
            lost, dupes = orc.lost_dupes(idx)
            orc.reassign(soa, lost, dupes)
            orc.add_noise(soa, COV['resample_cov'], orc.native_normals(n, 0, 5, 2, k))
            orc.mean_cov(soa)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        per_step = (time.perf_counter() - t0) / done
        out['value' if label == 'all' else 'value_1thread'] = round((n / float(P)) / per_step, 5)
        out['cores' if label == 'all' else 'cores_1thread'] = threads
        out['sample' if label == 'all' else 'sample_1thread'] = '%d particles x %d step(s) on %d thread(s), %.2f s/step, scaled linearly to %d' % (
            n, done, threads, per_step, P)
    return out


def alg_bytes(P, B, map_bytes, world):
    """Algorithmic (compulsory) HBM bytes per step of each phase.  The fused step keeps
    z, roll, pitch (the odometry's on every particle) out of the predict's stores and the gather's loads: 24 B less each
    than the separate calls (72 and 12 x world + 96)."""
    return {
        'predict': 48.0 * P,
        'update_mbes': 56.0 * P + 8.0 * B + map_bytes,
        'normalise': 24.0 * P,
        'scan': 28.0 * P,
        'resample': 12.0 * P * world + 72.0 * P,
        'mean_cov': 80.0 * P,
    }


def mbes_path(e, P):
    """Which kernels cast the last MBES update (mcl_mbes_last_path): the fan sweep (regularly triangulated
    meshes) or the ray traversal, and how much of the cloud the first stage passed on to the general kernels."""
    path, handed, deferred = e.mbes_last_path()
    by_slice, by_trav = e.mbes_last_handover()
    return {'algorithm': {1: 'fan sweep (mcl_sweep.h: k_mbes_sweep)', 2: 'fan slice (mcl_slice.h: k_mbes_slice)'}.get(
                path, 'ray traversal (mcl_mbes.h: k_mbes_fast)'),
            'dominant_launch': {1: 'k_mbes_sweep<SURF,false>', 2: 'k_mbes_slice<false>'}.get(path, 'k_mbes_fast<SURF,false>'),
            'particles_handed_to_traversal': by_trav if path in (1, 2) else None,
            'particles_handed_to_fan_slice': by_slice if path == 1 else None,
            'groups_deferred_to_general_kernel': deferred, 'of_particles': P}


def kernel_table(tim, alg, steps):
    kernels = {}
    for name, (ms, cnt) in tim.items():
        if cnt == 0:
            continue
        entry = dict(avg_ms=round(ms / cnt, 5), ms_per_step=round(ms / steps, 5), regions=int(cnt))
        if name in alg and ms > 0:
            gbs = alg[name] * steps / (ms * 1e-3) / 1e9
            entry.update(alg_bytes_per_step=alg[name], achieved_gbs=round(gbs, 2), hbm_frac=round(gbs / HBM_PEAK_GBS, 5))
        kernels[name] = entry
    return kernels


def run_leg(engine, name, m, P, B, steps, warmup, device=0, x0=0.0, cov=None, resample=True, landmarks=None, m2o=None,
            rccl_1rank=False, sigma=None):
    """One extra workload leg on a fresh engine: returns ms per step (wall, synchronised around the timed
    block) and the per-phase HIP-event times.  resample=False: predict + MBES update only (the cloud
    keeps its width); landmarks=(xyz, n_det): config 5 -- the landmark k-NN update accumulates onto the
    MBES log-likelihood of the same ping."""
    import numpy as np
    from smarc_navigation_amd import synth
    cov = cov or COV
    sig = sigma or SIGMA   # the likelihood's sigma (the simulated pings keep SIGMA of range noise)
    total = steps + warmup
    n_tim = min(steps, 10)   # the per-phase event block runs on NEW steps total .. total + n_tim - 1
    n_all = total + n_tim
    stream = synth.odom_stream(n_all, x0=x0)
    ba = synth.beam_angles(B)
    ranges = make_ranges(engine, m, stream['truth'], ba, SIGMA, R_MAX, device=device, m2o=m2o)
    if rccl_1rank:
        # the SHARDED pipeline on one GPU: a 1-rank RCCL communicator (MCL_FORCE_COMM, read at mcl_create) makes the
        # step run its exchange phases -- max all-reduce, totals / CDF all-gathers, the state gather on the second
        # communicator under the MBES update, moments all-reduce -- with nobody to talk to
        os.environ['MCL_FORCE_COMM'] = '1'
    try:
        e = engine.Engine(P, seed=5, device=device, m2o=m2o, **cov)
        if rccl_1rank:
            with stdout_to_stderr():
                e.comm_init(engine.comm_unique_id())
    finally:
        if rccl_1rank:
            del os.environ['MCL_FORCE_COMM']
    attach_map(e, m)
    dets = None
    if landmarks is not None:
        lm_xyz, n_det = landmarks
        e.set_landmarks(lm_xyz)
        rs = np.random.RandomState(8)
        dets = np.zeros((n_all, n_det, 3))
        for k in range(n_all):
            t = stream['truth'][k]
            T = synth.rigid_matrix(*t)
            d2 = np.sum((lm_xyz[:, :2] - t[:2]) ** 2, axis=1)
            near = lm_xyz[np.argsort(d2)[:n_det]]
            dets[k] = (near - T[:3, 3]).dot(T[:3, :3]) + 0.05 * rs.randn(n_det, 3)  # R^T (l - p): sensor frame
    e.init_particles()

    def step(k):
        if resample and landmarks is None:
            e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba,
                        sig, R_MAX)
            return
        if resample and os.environ.get('MCL_BENCH_CONFIG5_SEPARATE') != '1':
            # config 5: the same step with the landmark k-NN likelihood of the ping on top (mcl_step_mbes_landmarks)
            e.step_mbes_landmarks(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'], ranges[k], ba,
                                  sig, R_MAX, dets[k], 0.3, k=4, gate=11.345)
            return
        e.predict(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'])
        e.update_mbes(ranges[k], ba, sig, R_MAX)
        if landmarks is not None:
            e.update_landmarks(dets[k], 0.3, k=4, gate=11.345, accumulate=True)
        if resample:
            e.resample()
            e.mean_cov_async()

    for k in range(warmup):
        step(k)
    e.sync()
    # wall clock WITHOUT the per-phase HIP events (each timed region costs the stream a few microseconds: at 65 536
    # particles that is a third of the step) ...
    t0 = time.perf_counter()
    for k in range(warmup, total):
        step(k)
    e.sync()
    dt = time.perf_counter() - t0
    # the cloud the timed block ended on, against the truth of its last step
    cloud = None
    if resample:
        mean, _, c9 = e.mean_cov()
        truth = stream['truth'][total - 1]
        cloud = {'sigma_x_m': round(float(np.sqrt(max(c9[0], 0.0))), 4), 'sigma_y_m': round(float(np.sqrt(max(c9[4], 0.0))), 4),
                 'mean_error_m': round(float(np.hypot(mean[0] - truth[0], mean[1] - truth[1])), 4)}
    # ... then the per-phase breakdown with the events on, over the NEXT n_tim steps of the same stream (the filter
    # goes on with consistent odometry and pings: a steady state, not a replay)
    e.timing_enable(True)
    for k in range(total, total + n_tim):
        step(k)
    e.sync()
    tim = e.timing_get()
    e.timing_enable(False)
    path = mbes_path(e, P)
    e.close()
    ms = 1e3 * dt / steps
    out = dict(mbes_path=path, workload='%d particles x %d beams, %s%s%s' % (
        P, B, m['desc'], '' if resample else ', predict + MBES update only (no resample: the cloud keeps its width)',
        '' if landmarks is None else ', + %d detections x %d landmarks k-NN (k=4) per ping' % (landmarks[1], len(landmarks[0]))),
        steps=steps, warmup=warmup, ms_per_step=round(ms, 4), steps_per_s=round(1e3 / ms, 2),
        kernels={k: round(v[0] / n_tim, 5) for k, v in tim.items() if v[1]},
        kernels_note='HIP-event regions of the %d steps that follow the timed block; ms_per_step is wall clock without events' % n_tim)
    if cloud:
        out['cloud'] = cloud
    if sigma:
        out['likelihood_sigma_m'] = round(sig, 3)
    return out


# ------------------------------------------------------------------------------------------ worker
def dry_run(a, rank, world):
    """Launcher self-test: rendezvous, barrier and a rank count over gloo -- no engine, no GPU."""
    import torch
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if rank == a.dry_run_fail_rank:
        sys.exit(7)
    if rank == a.dry_run_hang_rank:
        time.sleep(3600)
    if world > 1:
        dist.init_process_group('gloo', rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64)
        dist.all_reduce(t)
        dist.barrier()
        n = int(t[0])
        dist.destroy_process_group()
    else:
        n = 1
    if rank == 0:
        print(json.dumps({'dry_run': True, 'n_gpus': world, 'gloo_ranks': n,
                          'spawned_by_bench': os.environ.get('MCL_BENCH_SPAWNED') == '1'}))
    return 0


class stdout_to_stderr(object):
    """RCCL prints a version banner on the C stdout at communicator creation (buffered: it would come out at exit,
    after the JSON line).  Inside this block fd 1 is stderr, and the C stdio buffers are flushed before fd 1 is
    given back, so stdout stays reserved for the ONE JSON line."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import ctypes
        sys.stdout.flush()
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def rccl_library():
    """Who answers the nccl* calls of this process: librccl, or the test double of tests/fake_nccl (LD_PRELOAD) -- a line
    produced over the double measures the host code's call sequence on one device, not a transport."""
    import ctypes
    try:
        return 'TEST DOUBLE tests/fake_nccl (LD_PRELOAD): not RCCL, not a measurement' if ctypes.CDLL(None).fake_nccl_present() == 1 else 'librccl'
    except (AttributeError, OSError):
        return 'librccl'


def setup_comm(e, engine, dist, rank, world, want_overlap):
    """RCCL communicator(s) of the data path.  The overlapped state all-gather drives two communicators
    on two streams; a self-test with a deadline runs that exact pattern first, and if any rank does not
    complete it every rank aborts and re-initialises without the overlap (in-line gather).  A second
    failure exits non-zero.  Never hangs: every wait has a deadline."""
    import torch
    overlap = want_overlap
    for attempt in range(2):
        uid = [engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ok = 1
        with stdout_to_stderr():
            try:
                e.comm_init(uid[0], overlap=overlap)
                e.comm_selftest(30000)
            except engine.MclError as ex:
                sys.stderr.write('rank %d: communicator self-test failed (%s)\n' % (rank, ex))
                ok = 0
        t = torch.tensor([ok], dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t[0]) == 1:
            ranks, has_overlap = e.comm_ranks()
            return ranks, has_overlap
        e.comm_shutdown(abort=True)
        if not overlap:
            break
        overlap = False
    sys.stderr.write('rank %d: no working RCCL communicator\n' % rank)
    sys.exit(3)


def worker(a, rank, world, local_rank):
    import numpy as np
    dist = None
    if world > 1:
        # torch first: its bundled HIP runtime / RCCL are then the single copies in the process and
        # libmcl_hip.so binds to them (loading the library first would map a second HIP runtime)
        import torch
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)

    from smarc_navigation_amd import engine, synth

    B = a.beams
    P = a.particles if a.scaling == 'weak' else a.total_particles // world
    if a.scaling == 'strong' and P * world != a.total_particles:
        sys.stderr.write('bench: --total-particles must be a multiple of the rank count\n')
        sys.exit(2)
    m = build_map(a.map)
    nblocks = max(1, int(math.ceil(MIN_TIMED_STEPS / float(a.steps))))
    nblocks = min(nblocks, max(1, (4000 - a.warmup) // a.steps - 1))  # mcl_mean_history keeps 4096 results
    total_steps = a.warmup + a.steps * (nblocks + 1)   # (+ one block at the end that carries the per-phase events)
    stream = synth.odom_stream(total_steps)
    ba = synth.beam_angles(B)

    e = engine.Engine(P, seed=5, device=local_rank, rank=rank, world=world, n_global=P * world,
                      global_offset=P * rank, **COV)
    rccl_ranks, has_overlap = 1, False
    if world > 1:
        rccl_ranks, has_overlap = setup_comm(e, engine, dist, rank, world, not a.no_overlap)
        if rccl_ranks != world:
            sys.stderr.write('rank %d: RCCL connected %d ranks, expected %d\n' % (rank, rccl_ranks, world))
            sys.exit(4)
    attach_map(e, m)
    ranges = make_ranges(engine, m, stream['truth'], ba, SIGMA, R_MAX, device=local_rank)
    e.init_particles()

    def barrier():
        e.sync()
        if dist is not None:
            dist.barrier()

    def run(k0, k1):
        for k in range(k0, k1):
            if rank == a.test_kill_rank and k == a.warmup + a.test_kill_step:
                os.kill(os.getpid(), 9)   # (test hook: a rank that vanishes in the middle of the timed block)
            e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                        ranges[k], ba, SIGMA * (math.sqrt(float(B)) if a.temper else 1.0), R_MAX)

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    run(0, a.warmup)
    barrier()
    # ---- the contract's timed region: EXACTLY --steps steps between two barrier + synchronize pairs
    t0 = time.perf_counter()
    run(a.warmup, a.warmup + a.steps)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    # ---- the same block repeated until >= MIN_TIMED_STEPS steps are timed: median and p95 per block
    block_ms = [1e3 * dt / a.steps]
    for b in range(1, nblocks):
        k0 = a.warmup + b * a.steps
        t0 = time.perf_counter()
        run(k0, k0 + a.steps)
        barrier()
        block_ms.append(1e3 * max_over_ranks(time.perf_counter() - t0) / a.steps)
    # ---- per-phase HIP events on ONE more block, after the timed ones (the events cost the stream a few microseconds per
    # region, so they stay out of every timed block; at the end of the run the filter is in its steady state, like
    # the median block and like the rocprofv3 average the roofline is checked against -- the first steps after the
    # warm-up still work on a wider cloud and a sweep launch takes 15 % longer there)
    e.timing_enable(True)
    k0 = a.warmup + nblocks * a.steps
    run(k0, k0 + a.steps)
    barrier()
    tim = e.timing_get()
    # every rank's share of the resample exchange (states sent to peers, lost slots), so that the first multi-GPU run
    # explains itself: which rank fed which, and how far the traffic is from the (world - 1) / world worst case
    ex_all = None
    if world > 1 and dist is not None:
        s_, l_ = e.exchange_stats()
        tl = [torch.zeros(2, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(tl, torch.tensor([float(s_), float(l_)], dtype=torch.float64))
        ex_all = [(float(x[0]), float(x[1])) for x in tl]
    e.timing_enable(False)
    # pose RMSE of the filter's mean (x, y) against the synthetic ground truth over every timed step
    n_timed = min(a.steps * (nblocks + 1), 4096)   # (the library keeps the last 4 096 results: mcl_mean_history)
    hist = e.mean_history(n_timed)
    truth = stream['truth'][total_steps - n_timed:total_steps]
    pose_rmse = float(np.sqrt(np.mean((hist[:, 0] - truth[:, 0]) ** 2 + (hist[:, 1] - truth[:, 1]) ** 2)))

    out = None
    if rank == 0:
        first_block_ms = 1e3 * dt / a.steps
        ms_per_step = pctl(block_ms, 50)   # the median block (the first block still holds the clock ramp)
        n_timed = a.steps * nblocks
        total_particles = P * world
        path_main = mbes_path(e, P)
        value = 1e3 / ms_per_step * (total_particles / 1048576.0)
        alg = alg_bytes(P, B, m['bytes'], world)
        kernels = kernel_table(tim, alg, a.steps)
        dom = max((k for k in kernels if k in alg), key=lambda k: tim[k][0])
        # the dominant LAUNCH: HIP events around that one kernel (MCL_K_MBES_MAIN), not the update's whole region
        if dom == 'update_mbes' and tim.get('mbes_main', (0, 0))[1]:
            dom_ms = tim['mbes_main'][0] / tim['mbes_main'][1]
            dom_time_source = ('HIP events (no system fence) around the one launch (%s), mean of the %d launches of the block after '
                               'the timed ones: the filter in its steady state.  rocprof_launch_us is the mean over ALL '
                               'dispatches of a profiled run, whose first steps after the init work on a wider cloud '
                               '(kernel stats: min / avg / max)' % (path_main['dominant_launch'], tim['mbes_main'][1]))
        else:
            dom_ms = tim[dom][0] / a.steps
            dom_time_source = 'HIP-event region of the phase'
        achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
        # the cloud the timed steps ran on: spread of the last step's posterior and its effective sample size
        _, _, c9 = e.last_mean_cov()
        q_fix, q_tot = e.fixed_weights()
        qf = q_fix.astype(np.float64) / float(q_tot)
        n_eff = float(1.0 / np.sum(qf * qf))
        cloud = {'sigma_x_m': round(float(np.sqrt(max(c9[0], 0.0))), 4), 'sigma_y_m': round(float(np.sqrt(max(c9[4], 0.0))), 4),
                 'n_eff': round(n_eff, 1), 'n_eff_frac': round(n_eff / P, 6),
                 'note': 'posterior spread after the last timed step and the effective sample size of its weights (this '
                         'shard): 512 beams at sigma = %.1f m per ping collapse the cloud to the resampling noise within '
                         'the warm-up -- extra.filter_tempered runs the same step on a filter that keeps a healthy spread (its dominant '
                         'launch takes ~20 %% longer: compare extra.filter_tempered.kernels.mbes_main with roofline.launch_us)' % SIGMA}
        # PMC-measured HBM traffic of the dominant kernel: collected offline (counters cannot be read inside
        # this run) by tools/pmc_summarise.py into profiles/<round>_traffic.json, attached ONLY when that file was
        # taken at the kernel sources this library was built from and at this workload
        traffic, traffic_source, pmc = None, None, {}
        src = source_hash()
        import glob
        for tpath in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic.json')), reverse=True):   # newest round first
            try:
                with open(tpath) as f:
                    tj = json.load(f)
            except (IOError, ValueError):
                continue
            rel = os.path.relpath(tpath, ROOT)
            if tj.get('source_hash') == src and P == 1048576 and B == 512 and world == 1:
                pmc = tj.get(m['kind'], {})
                traffic = pmc.get('traffic_bytes_per_launch')
                traffic_source = 'offline: %s (rocprofv3 --pmc passes, kernel sources %s)' % (rel, src)
            elif tj.get('source_hash') != src:
                traffic_source = 'none: %s is for kernel sources %s, this library is %s' % (rel, tj.get('source_hash'), src)
            break
        streaming = [k for k in ('predict', 'normalise', 'scan', 'resample', 'mean_cov') if k in kernels]
        out = {
            'metric': METRIC,
            'value': round(value, 3), 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms_per_step, 4), 'first_block_ms': round(first_block_ms, 4),
            'value_definition': 'steps/s of the MEDIAN of the timed blocks (each exactly --steps steps between barrier + '
                                'synchronize pairs, max over ranks; >= %d steps in all); first_block_value is the contract\'s '
                                'single block right after the warm-up (rounds 1-2 reported that one)' % MIN_TIMED_STEPS,
            'first_block_value': round(1e3 / first_block_ms * (total_particles / 1048576.0), 3),
            'higher_is_better': True, 'scaling': a.scaling,
            'vs_baseline': None, 'dtype': 'f64 state / f32 ray-cast', 'data': 'synthetic',
            'steps_per_s': round(1e3 / ms_per_step, 3), 'particles_total': total_particles, 'cloud': cloud,
            'rccl_ranks': rccl_ranks, 'overlapped_state_gather': bool(has_overlap), 'rccl_library': rccl_library(),
            'launcher': 'bench.py' if os.environ.get('MCL_BENCH_SPAWNED') == '1' else ('external' if world > 1 else 'none'),
            'timed': {'blocks': nblocks, 'steps_per_block': a.steps, 'steps_total': n_timed,
                      'ms_per_step_median': round(pctl(block_ms, 50), 4), 'ms_per_step_p95': round(pctl(block_ms, 95), 4),
                      'ms_per_step_min': round(min(block_ms), 4), 'ms_per_step_max': round(max(block_ms), 4),
                      'note': 'the timed block (exactly --steps steps between barrier+synchronize pairs, max over ranks) '
                              'is repeated until >= %d steps are timed; value / ms_per_step are the MEDIAN block, '
                              'first_block_ms the first one; the per-phase HIP events (kernels, roofline.launch_us) are '
                              'recorded on one further block after them' % MIN_TIMED_STEPS},
            'config': {'workload': '%d particles/GPU x %d beams, %s%s, predict+MBES update+normalise+systematic '
                                   'resample+mean/cov per step' % (P, B, m['desc'], (
                                       ' -- RECOGNISED AS A LATTICE by mesh_build: swept on its 708 x 708 node heights, no triangle record '
                                       'is read (an irregular TIN of the same size: value_tin / extra.mesh_tin)') if m['kind'] == 'mesh' else ''),
                       'particles_per_gpu': P, 'beams': B, 'map': m['kind'], 'parallelism': 'particle-shard x%d' % world},
            # (the dominant kernel is bound by vector issue -- `frac` stays the HBM figure SURVEY 8(d)
            #  defines, `frac_valu` is the share of the chip's vector-issue slots, from the offline PMC passes)
            'roofline': {'bound': 'valu_issue' if dom == 'update_mbes' else 'hbm', 'kernel': dom, 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': traffic,
                         'launch_us': round(dom_ms * 1e3, 2), 'time_source': dom_time_source,
                         'alg_bytes_per_launch': alg[dom],
                         'alg_bytes_what': '56 B per particle + 8 B per beam + the map bytes this path must read: %s' % m['bytes_what'],
                         'rocprof_launch_us': pmc.get('kernel_avg_us'),
                         'traffic_source': traffic_source, 'kernel_source_hash': src,
                         'rays_per_s': round(P * B / (dom_ms * 1e-3), 1),
                         'valu_insts_per_ray': (round(pmc['valu_insts_per_launch'] * 64.0 / (P * B), 1)
                                                if pmc.get('valu_insts_per_launch') else None),
                         'valu_lane_utilisation': (round(pmc['valu_lane_utilisation'], 3)
                                                   if pmc.get('valu_lane_utilisation') else None),
                         # share of the chip's vector-issue slots: a wave64 VALU instruction occupies a SIMD-32
                         # for 2 cycles (MI355X_MICROARCH.md "Wave scheduling"), 1024 SIMDs at 2.4 GHz
                         'valu_issue_frac': (round(pmc['valu_insts_per_launch'] / (dom_ms * 1e-3) / (1024 * 2.4e9 / 2.0), 3)
                                             if pmc.get('valu_insts_per_launch') else None),
                         'frac_valu': (round(pmc['valu_insts_per_launch'] / (dom_ms * 1e-3) / (1024 * 2.4e9 / 2.0), 3)
                                       if pmc.get('valu_insts_per_launch') else None),
                         'bound_note': 'valu_issue: the kernel moves ~61 MB and executes ~2e8 wave instructions per launch; `achieved` / '
                                       '`peak` / `frac` are the HBM figures SURVEY 8(d) asks for (they move only with the instruction '
                                       'count), frac_valu the share of the vector-issue slots (wave64 VALU = 2 cycles of a SIMD, 1 024 '
                                       'SIMDs at 2.4 GHz)',
                         'streaming_ms_per_step': round(sum(kernels[k]['ms_per_step'] for k in streaming), 5),
                         'mbes_path': path_main,
                         'note': 'the MBES update is bound by vector / LDS issue, not by HBM (SURVEY 8d): its compulsory '
                                 'traffic is 56 B per particle + the map; valu_* come from the offline PMC '
                                 'passes named in traffic_source (null when that file does not match this library), '
                                 'valu_insts_per_ray = wave instructions x 64 / (particles x beams); '
                                 'streaming kernels are listed in "kernels" with their own HBM fractions'},
            'kernels': kernels,
            'pose_rmse_m': round(pose_rmse, 4),
        }
        if world > 1:
            sent, lost = e.exchange_stats()
            ops, rounds = e.exchange_ops()
            nst = float(total_steps)
            out['exchange'] = {'rank0_p2p_ops_per_exchange': round(ops / max(rounds, 1), 2), 'p2p_ops_bound': 2 * (world - 1),
                               'phases_ms_per_step': {k: kernels[k]['ms_per_step'] for k in ('comm_records', 'pack', 'comm_p2p', 'comm_moments', 'comm') if k in kernels},
'mode': os.environ.get('MCL_EXCHANGE', 'p2p (O(n) per rank: hand-over records + point-to-point surplus copies)'),
                               'rank0_states_sent_per_step': round(sent / nst, 1), 'rank0_lost_slots_per_step': round(lost / nst, 1),
                               'rank0_bytes_sent_per_step': round(24.0 * sent / nst, 1),
                               'bytes_sent_per_step_by_rank': [round(24.0 * x[0] / nst, 1) for x in ex_all] if ex_all else None,
                               'lost_slots_per_step_by_rank': [round(x[1] / nst, 1) for x in ex_all] if ex_all else None,
                               'fraction_of_cloud_moved_per_step': round(sum(x[0] for x in ex_all) / nst / (P * world), 6) if ex_all else None,
                               'worst_case_fraction': round((world - 1.0) / world, 4),
                               'note': 'x, y, yaw of every surplus copy that fills a lost slot of ANOTHER rank (z, roll, pitch are '
                                       'the odometry\'s on every particle after predict); the all-gather exchange of rounds 1-2 '
                                       'moved 28 B x N_global per rank per step'}
    e.close()

    # ---- N = 1: CPU baseline, trajectory RMSE against the oracle, extra workload legs
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # all host cores (the oracle's particle loops are OpenMP-parallel) and, beside it, one thread
        ns = a.cpu_particles or (8192 if m['kind'] == 'grid' else 4096)
        cores = host_cores()
        one = cpu_baseline(m, stream, ranges, ba, COV, 1048576, ns, 1, 6.0)
        # all cores: >= 262 144 particles (a quarter of the metric's cloud) so that the trajectory it leaves is a
        # meaningful reference for "pose RMSE vs ref" -- about 25 s of CPU on 16 threads
        allc = cpu_baseline(m, stream, ranges, ba, COV, 1048576, max(ns * min(cores, 32), 262144), cores, 22.0, max_steps=6)
        allc['value_1thread'] = one['value']
        allc['sample_1thread'] = one['sample']
        # "pose RMSE vs ref": the same filter (same Philox draws, same stream and map) on the GPU at the
        # oracle's sample size, mean (x, y) trajectory against the oracle's over the steps it ran
        ref_xy, n_ref = allc.pop('_means'), allc.pop('_n')
        one.pop('_means'), one.pop('_n')
        if a.rmse_particles and a.rmse_particles != n_ref:
            big = cpu_baseline(m, stream, ranges, ba, COV, 1048576, a.rmse_particles, cores, 1e9, max_steps=6)
            ref_xy, n_ref = big.pop('_means'), big.pop('_n')
        g = engine.Engine(n_ref, seed=5, device=local_rank, **COV)
        attach_map(g, m)
        g.init_particles()
        for k in range(len(ref_xy)):
            g.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                        ranges[k], ba, SIGMA, R_MAX)
        g.sync()
        gh = g.mean_history(len(ref_xy))
        g.close()
        rm = float('%.3g' % np.sqrt(np.mean(np.sum((gh[:, :2] - ref_xy) ** 2, axis=1))))
        out['pose_rmse_vs_oracle_m'] = rm
        out['pose_rmse_vs_oracle_at'] = '%d particles x %d beams x %d steps (GPU filter vs oracle, same Philox draws)' % (
            n_ref, B, len(ref_xy))
        allc['pose_rmse_vs_oracle_m'] = rm
        allc['reference_python'] = REFERENCE_PYTHON
        out['cpu_baseline'] = allc
    if rank == 0 and world == 1 and not a.no_extra:
        extra = {}
        legs = []
        mesh = m if a.map == 'mesh' else build_map('mesh')
        if a.map != 'grid':
            legs.append(('grid', dict(m=build_map('grid'), P=1048576, B=512, steps=50, warmup=40)))
        # a filter that keeps a healthy spread: the 512 beams of a ping share ONE error budget (likelihood tempered by
        # 1 / B: sigma_eff = sigma sqrt(B) = 4.5 m), so the posterior stays decimetres wide instead of collapsing to
        # the resampling noise -- the same kernels, the same launches, lanes of a wave no longer walk the same triangles
        legs.append(('filter_tempered', dict(m=mesh, P=1048576, B=512, steps=30, warmup=10, sigma=SIGMA * math.sqrt(512.0))))
        if a.map != 'mesh-general':
            legs.append(('mesh_general', dict(m=build_map('mesh-general'), P=1048576, B=512, steps=10, warmup=2)))
        if a.map != 'mesh-soup':
            legs.append(('mesh_soup_irregular', dict(m=build_map('mesh-soup'), P=1048576, B=512, steps=10, warmup=2)))
        if a.map != 'mesh-adjacency':
            legs.append(('mesh_adjacency', dict(m=build_map('mesh-adjacency'), P=1048576, B=512, steps=50, warmup=40)))
        # (the surface legs run 40 steps before their 50 timed ones: like the headline's median block they time the filter's
        #  steady state -- rounds 1-5 timed steps 3 .. 22, on which the cloud is still contracting and a sweep launch takes
        #  15 % longer; every leg records its `steps` and `warmup`)
        # the path of a REAL survey mesh: an irregular height-field TIN through the adjacency walk (k_mbes_sweep<5,...>) --
        # as generated (row-major), in random input order (mesh_build's Morton pass must make the two alike), and on a
        # filter that keeps a healthy spread (the realistic deployment point)
        tin = m if a.map == 'mesh-tin' else build_map('mesh-tin')
        if a.map != 'mesh-tin':
            legs.append(('mesh_tin', dict(m=tin, P=1048576, B=512, steps=50, warmup=40)))
        if a.map != 'mesh-tin-shuffled':
            legs.append(('mesh_tin_shuffled', dict(m=build_map('mesh-tin-shuffled'), P=1048576, B=512, steps=50, warmup=40)))
        legs.append(('mesh_tin_tempered', dict(m=tin, P=1048576, B=512, steps=30, warmup=10, sigma=SIGMA * math.sqrt(512.0))))
        # ... and the same TIN with a data gap of ten triangles under the swath of the timed steps: every particle's slice runs
        # into it, the sweep hands the whole cloud over -- to the fan slice (mcl_host_update.h; round 5 / MCL_HANDOVER_SLICE=0:
        # to the ray traversal, 21 ms per step).  The worst case of a survey mesh with gaps, not its average.
        legs.append(('mesh_tin_hole_under_swath', dict(m=punch_hole(tin, 1.0, 10.0), P=1048576, B=512, steps=50, warmup=40)))
        # ... and with gaps EVERYWHERE (one per 6 x 6 m, 13 % of the triangles missing): every particle's slice crosses several
        # ... and with the OUTLINE of a real survey (border triangles missing at random, bays cut in from the sides: synth.mesh_ragged), the
        # track 24 m inside the southern outline: one side of every fan runs out over the outline and across the bays
        legs.append(('mesh_tin_ragged_outline', dict(m=dict(tin, tris=synth.mesh_ragged(tin['verts'], tin['tris']), desc=tin['desc'] + ', ragged outline with six bays (synth.mesh_ragged), the track 24 m inside it'),
                                                     P=1048576, B=512, steps=50, warmup=40, m2o=synth.rigid_matrix(100.0, -330.0, 0.0, 0.0, 0.0, 0.0))))
        # ... and the vehicle OFF that map, 6 m beyond the bounding box (a lawn-mower turn outside the surveyed area): every sensor
        # looks back in over the outline -- the walk starts where the fan plane runs onto the mesh
        legs.append(('mesh_tin_vehicle_off_the_map', dict(m=legs[-1][1]['m'], P=1048576, B=512, steps=50, warmup=40, m2o=synth.rigid_matrix(100.0, -360.0, 0.0, 0.0, 0.0, 0.0))))
        legs.append(('mesh_tin_gaps_everywhere', dict(m=punch_gaps(tin), P=1048576, B=512, steps=50, warmup=40)))
        # global-localisation regime: sigma = 50 m cloud that nothing collapses (no resample).  Particles are
        # initialised around the odom origin (auv_particle.py:24), so the map <- odom transform puts that
        # origin 250 m inside the map; 'cloud_wide_at_border' leaves it 64 m from the western border, where
        # a tenth of the cloud is off the map and its groups take the general kernel
        shift = synth.rigid_matrix(250.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        wide = dict(COV, init_cov=[2500.0, 2500.0, 0.0, 0.0, 0.0, 0.05])
        legs.append(('cloud_wide', dict(m=mesh, P=1048576, B=512, steps=10, warmup=3, resample=False, cov=wide, m2o=shift)))
        legs.append(('cloud_wide_at_border', dict(m=mesh, P=1048576, B=512, steps=10, warmup=3, resample=False, cov=wide)))
        legs.append(('cloud_converged_update_only', dict(m=mesh, P=1048576, B=512, steps=10, warmup=3, resample=False,
                                                         m2o=shift)))
        # the headline workload over seconds instead of tenths of a second: under this vector load the chip gives some clock
        # back after about a second (MI355X_MICROARCH.md, DVFS) -- what a long replay sees
        legs.append(('sustained_3000_steps', dict(m=mesh, P=1048576, B=512, steps=3000, warmup=20)))
        legs.append(('config2', dict(m=build_map('grid'), P=65536, B=256, steps=200, warmup=20)))
        legs.append(('config4_shard', dict(m=mesh, P=524288, B=512, steps=100, warmup=10)))
        legs.append(('config4_shard_rccl_1rank', dict(m=mesh, P=524288, B=512, steps=100, warmup=10, rccl_1rank=True)))   # (100 steps each: their DIFFERENCE -- the exchange at one rank, ~16 us -- is what the two are read for)
        legs.append(('config5_shard', dict(m=mesh, P=524288, B=512, steps=30, warmup=5,
                                           landmarks=(synth.landmark_map(4096, (-64.0, -354.0, 643.0, 353.0)), 16))))
        cpu_todo = []
        cpu_legs = () if a.no_cpu_baseline else ('grid', 'config2', 'config4_shard', 'config5_shard', 'mesh_tin')   # "at every config"
        for name, kw in legs:
            try:
                lm, lP, lB, lland = kw['m'], kw['P'], kw['B'], kw.get('landmarks')
                extra[name] = run_leg(engine, name, kw.pop('m'), kw.pop('P'), kw.pop('B'), kw.pop('steps'), kw.pop('warmup'),
                                      device=local_rank, **kw)
                if name in cpu_legs:
                    cpu_todo.append((name, lm, lP, lB, lland))
            except Exception as ex:  # a failing leg must not cost the headline line
                extra.setdefault(name, {})['error'] = '%s: %s' % (type(ex).__name__, ex)
        # (the oracle's legs AFTER every GPU leg: its OpenMP workers keep spinning on the granted cores for a while after a
        #  parallel region, and a host-latency-bound GPU leg right behind one -- config4_shard_rccl_1rank -- measured 0.43 ms
        #  instead of 0.26)
        for name, lm, lP, lB, lland in cpu_todo:
            try:
                extra[name]['cpu_baseline'] = cpu_leg_baseline(engine, lm, lP, lB, local_rank, landmarks=lland)
            except Exception as ex:
                extra[name]['cpu_baseline'] = {'error': '%s: %s' % (type(ex).__name__, ex)}
        out['extra'] = extra
        # the numbers of the other legs where a reader of `value` sees them: the headline cloud has
        # collapsed to the resampling noise; filter_tempered keeps a posterior decimetres wide on the same kernels
        ft = extra.get('filter_tempered', {})
        if ft.get('ms_per_step'):
            out['value_healthy_cloud'] = round(1e3 / ft['ms_per_step'], 3)
            out['value_healthy_cloud_what'] = ('steps/s of extra.filter_tempered: the same step on a filter whose posterior stays '
                                               'decimetres wide (likelihood tempered by 1 / beams)')
        # ... and the general height-field mesh: `value` is the LATTICE special case (a triangulated regular grid is recognised
        # and swept on its node heights, no triangle record read); an irregular TIN takes the adjacency walk
        for key, leg, what in (('value_tin', 'mesh_tin', 'an irregular height-field TIN (the adjacency sweep k_mbes_sweep<5,...>), collapsed cloud like `value`'),
                               ('value_tin_shuffled_input', 'mesh_tin_shuffled', 'the same TIN handed over in random vertex / triangle order'),
                               ('value_tin_healthy_cloud', 'mesh_tin_tempered', 'the TIN with a posterior that stays decimetres wide: the realistic deployment point'),
                               ('value_tin_gaps_everywhere', 'mesh_tin_gaps_everywhere', 'the TIN with a data gap per 6 x 6 m (13 % of its triangles missing): the walk crosses them by their rims'),
                               ('value_tin_ragged_outline', 'mesh_tin_ragged_outline', 'the TIN with the outline of a real survey (sawtooth border, bays), the track 24 m inside it')):
            lg = extra.get(leg, {})
            if lg.get('ms_per_step'):
                out[key] = round(1e3 / lg['ms_per_step'], 3)
                out[key + '_what'] = 'steps/s of extra.%s: %s' % (leg, what)
        out['extra_summary'] = {k: v.get('ms_per_step') for k, v in extra.items()}
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    a = parse()
    env_world = os.environ.get('WORLD_SIZE')
    if env_world is None and a.gpus > 1:
        sys.exit(launch(a, sys.argv[1:]))
    world = int(env_world) if env_world is not None else 1
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if os.environ.get('MCL_BENCH_ONE_DEVICE') == '1':
        # test hook (tests/test_gpu_bench_multirank.py): every rank on device 0 -- only possible with the librccl test
        # double preloaded (RCCL itself refuses two ranks on one device); the JSON line says which library answered
        local_rank = 0
    if world != a.gpus:
        sys.stderr.write('bench: WORLD_SIZE=%d but --gpus %d: refusing to measure a different rank count\n' % (world, a.gpus))
        sys.exit(2)
    if a.dry_run:
        sys.exit(dry_run(a, rank, world))
    sys.exit(worker(a, rank, world, local_rank))


if __name__ == '__main__':
    main()
