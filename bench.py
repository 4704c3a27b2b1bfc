#!/usr/bin/env python3
"""bench.py -- filter steps/sec of the MCL hot path (BASELINE.json metric).

One "step" = predict (IMU+DVL motion model + process noise) -> MBES update (per-particle,
per-beam ray-cast + Gaussian log-likelihood) -> weight normalisation -> systematic resample
(+ keep/lost/dupes reassign + resampling noise) -> mean/covariance, on synthetic streams
(SURVEY.md 8(d)), all inputs resident in HBM except the per-ping 512 ranges (2 KiB).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--map grid|mesh] [--particles P]

N > 1: launched by torch.distributed.run, one rank per GPU.  Particles shard by contiguous
global id (weak scaling: P particles per GPU); the data-path collectives are native RCCL calls
inside libmcl_hip.so, torch.distributed (gloo) only carries the RCCL unique id and the timing
barrier.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--particles', type=int, default=1048576, help='particles per GPU')
    ap.add_argument('--beams', type=int, default=512)
    ap.add_argument('--map', default='mesh', choices=['grid', 'mesh'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--mesh-general', action='store_true',
                    help='do not use the structured-mesh path (detected for triangulated regular grids)')
    ap.add_argument('--cpu-particles', type=int, default=0, help='oracle sample size (0 = auto)')
    return ap.parse_args()


def build_map(kind):
    from smarc_navigation_amd import synth
    if kind == 'grid':
        origin = (-64.0, -256.0)
        z = synth.bathymetry_grid(512, 512, 1.0, origin, seed=3)
        return dict(kind='grid', z=z, origin=origin, res=1.0, bytes=z.nbytes,
                    desc='512x512 fp32 height grid, 1 m cells')
    origin = (-64.0, -354.0)
    z = synth.bathymetry_grid(708, 708, 1.0, origin, seed=3)
    verts, tris = synth.mesh_from_grid(z, 1.0, origin)
    return dict(kind='mesh', z=z, origin=origin, res=1.0, verts=verts, tris=tris,
                bytes=verts.nbytes + tris.nbytes,
                desc='%d-triangle mesh (708x708 height field triangulated)' % tris.shape[0])


a_mesh_general = [False]  # --mesh-general: force the triangle-record traversal (arbitrary soups)


def attach_map(e, m):
    if m['kind'] == 'grid':
        e.set_map_grid(m['z'], m['origin'], m['res'])
    else:
        e.set_map_mesh(m['verts'], m['tris'], general=(a_mesh_general[0]))


def make_ranges(engine_mod, m, stream, n_steps, beam_angles, sigma, r_max, device=0):
    """Synthetic pings: expected ranges at the truth pose (one-particle engine on the GPU) + noise."""
    e = engine_mod.Engine(1, rng_mode=engine_mod.RNG_REPLAY, device=device)  # this rank's own GPU
    attach_map(e, m)
    rs = np.random.RandomState(4)
    out = np.zeros((n_steps, beam_angles.size), np.float32)
    for k in range(n_steps):
        e.set_particles(stream['truth'][k][:, None].copy())
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


def cpu_baseline(m, stream, ranges, beam_angles, sigma, r_max, cov, n_full, n_sample, threads, budget_s):
    """The oracle (C restatement; its particle loops run on `threads` host threads) on a bounded
    sample of the same workload."""
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
        lw, _ = orc.mbes_update(soa, np.identity(4), [0] * 6, amap, beam_angles, ranges[k], sigma, r_max,
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
        if el > budget_s or steps >= min(len(ranges), 20):
            break
    per_step = el / steps
    return dict(value=(n / float(n_full)) / per_step, unit='steps/s', cores=threads, kind='port',
                pose_rmse_m=round(float(np.sqrt(sq_err / steps)), 4), _means=np.array(means), _n=n,
                sample='%d particles x %d beams x %d steps of the same stream+map on %d host thread(s) (%.2f s/step), '
                       'scaled linearly to %d particles' % (n, beam_angles.size, steps, threads, per_step, n_full))


def main():
    a = parse()
    a_mesh_general[0] = a.mesh_general
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus and world > 1:
        a.gpus = world
    dist = None
    if world > 1:
        # torch first: its bundled HIP runtime / RCCL are then the single copies in the process and
        # libmcl_hip.so binds to them (loading the library first would map a second HIP runtime)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)

    from smarc_navigation_amd import engine, synth

    P, B = a.particles, a.beams
    sigma, r_max = 0.2, 100.0
    cov = dict(init_cov=[2.0, 2.0, 0.0, 0.0, 0.0, 0.05], process_cov=[1e-4, 1e-4, 0.0, 0.0, 0.0, 1e-6],
               resample_cov=[1e-3, 1e-3, 0.0, 0.0, 0.0, 1e-5])
    m = build_map(a.map)
    total_steps = a.steps + a.warmup
    stream = synth.odom_stream(total_steps)
    ba = synth.beam_angles(B)

    e = engine.Engine(P, seed=5, device=local_rank, rank=rank, world=world, n_global=P * world,
                      global_offset=P * rank, **cov)
    if world > 1:
        import torch
        uid = [engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        # RCCL prints a version banner on stdout at communicator creation: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            e.comm_init(uid[0])
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    attach_map(e, m)
    ranges = make_ranges(engine, m, stream, total_steps, ba, sigma, r_max, device=local_rank)
    e.init_particles()

    def barrier():
        e.sync()
        if dist is not None:
            dist.barrier()

    def run(k0, k1):
        for k in range(k0, k1):
            e.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                        ranges[k], ba, sigma, r_max)

    run(0, a.warmup)
    barrier()
    e.timing_enable(True)
    t0 = time.perf_counter()
    run(a.warmup, total_steps)
    barrier()
    dt = time.perf_counter() - t0
    tim = e.timing_get()
    e.timing_enable(False)
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    # pose RMSE of the filter's mean (x, y) against the synthetic ground truth over the timed steps
    hist = e.mean_history(a.steps)
    truth = stream['truth'][a.warmup:total_steps]
    pose_rmse = float(np.sqrt(np.mean((hist[:, 0] - truth[:, 0]) ** 2 + (hist[:, 1] - truth[:, 1]) ** 2)))

    if rank == 0:
        ms_per_step = 1e3 * dt / a.steps
        value = world * a.steps / dt * (P / 1048576.0)
        # per-kernel algorithmic HBM bytes per launch (DESIGN.md "Roofline accounting")
        alg = {
            'predict': 72.0 * P,
            'update_mbes': 56.0 * P + 8.0 * B + m['bytes'],
            'normalise': 24.0 * P,
            'scan': 28.0 * P,
            'resample': 12.0 * P * world + 96.0 * P,
            'mean_cov': 80.0 * P,
        }
        kernels = {}
        for name, (ms, cnt) in tim.items():
            if cnt == 0:
                continue
            avg = ms / cnt
            entry = dict(avg_ms=round(avg, 5), regions=int(cnt))
            if name in alg and avg > 0:
                per_launch = alg[name] * (a.steps / float(cnt)) if name != 'update_mbes' else alg[name]
                gbs = alg[name] * a.steps / (ms * 1e-3) / 1e9
                entry.update(alg_bytes_per_step=alg[name], achieved_gbs=round(gbs, 2),
                             hbm_frac=round(gbs / HBM_PEAK_GBS, 5))
            kernels[name] = entry
        # PMC-measured HBM traffic of the dominant kernel (offline rocprofv3 passes, profiles/r01_traffic.json)
        traffic, pmc = None, {}
        try:
            with open(os.path.join(ROOT, 'profiles', 'r01_traffic.json')) as f:
                pmc = json.load(f).get(m['kind'], {})
            if P == 1048576 and B == 512 and world == 1:
                traffic = pmc.get('traffic_bytes_per_launch')
        except (IOError, ValueError):
            pass
        dom = max((k for k in kernels if k in alg), key=lambda k: tim[k][0])
        dom_ms = tim[dom][0] / a.steps
        achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
        out = {
            'metric': 'filter steps/sec at 1M particles x 512 MBES beams; pose RMSE vs ref',
            'value': round(value, 3), 'unit': 'steps/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64 state / f32 ray-cast', 'data': 'synthetic',
            'config': {'workload': '%d particles/GPU x %d beams, %s, predict+MBES update+normalise+systematic '
                                   'resample+mean/cov per step' % (P, B, m['desc']),
                       'particles_per_gpu': P, 'beams': B, 'map': m['kind'], 'parallelism': 'particle-shard x%d' % world},
            'roofline': {'bound': 'hbm', 'kernel': dom, 'achieved': round(achieved, 3), 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 6), 'traffic': traffic,
                         'rays_per_s': round(P * B / (dom_ms * 1e-3), 1),
                         'valu_insts_per_ray': (round(pmc['valu_insts_per_launch'] * 64.0 / (P * B), 1)
                                                if pmc.get('valu_insts_per_launch') and traffic else None),
                         'valu_lane_utilisation': (round(pmc['valu_lane_utilisation'], 3)
                                                   if pmc.get('valu_lane_utilisation') and traffic else None),
                         # share of the chip's vector-issue slots: a wave64 VALU instruction occupies a SIMD-32
                         # for 2 cycles (MI355X_MICROARCH.md "Wave scheduling"), 1024 SIMDs at 2.4 GHz
                         'valu_issue_frac': (round(pmc['valu_insts_per_launch'] / (dom_ms * 1e-3) / (1024 * 2.4e9 / 2.0), 3)
                                             if pmc.get('valu_insts_per_launch') and traffic else None),
                         'note': 'the ray-cast is VALU-bound, not HBM-bound (SURVEY 8d): PMC SQ_INSTS_VALU wave-instructions '
                                 'per ray-lane, the fraction of lanes active in them and the share of vector-issue '
                                 'slots they fill (profiles/r01_traffic.json); streaming kernels are listed in "kernels" with their own HBM fractions'},
            'kernels': kernels,
            'pose_rmse_m': round(pose_rmse, 4),
        }
        if world == 1 and not a.no_cpu_baseline:
            # all host cores (the oracle's particle loops are OpenMP-parallel) and, beside it, one thread
            ns = a.cpu_particles or (8192 if m['kind'] == 'grid' else 4096)
            cores = host_cores()
            one = cpu_baseline(m, stream, ranges, ba, sigma, r_max, cov, 1048576, ns, 1, 8.0)
            allc = cpu_baseline(m, stream, ranges, ba, sigma, r_max, cov, 1048576, ns * min(cores, 32), cores, 10.0)
            allc['value_1thread'] = one['value']
            allc['sample_1thread'] = one['sample']
            # "pose RMSE vs ref": the same filter (same Philox draws, same stream and map) on the GPU at the
            # oracle's sample size, mean (x, y) trajectory against the oracle's over the steps it ran
            ref_xy, n_ref = allc.pop('_means'), allc.pop('_n')
            one.pop('_means'), one.pop('_n')
            g = engine.Engine(n_ref, seed=5, device=local_rank, **cov)
            attach_map(g, m)
            g.init_particles()
            for k in range(len(ref_xy)):
                g.step_mbes(stream['v'][k], stream['wz'][k], stream['q'][k], stream['z'][k], stream['dt'],
                            ranges[k], ba, sigma, r_max)
            g.sync()
            gh = g.mean_history(len(ref_xy))
            g.close()
            allc['pose_rmse_vs_oracle_m'] = float('%.3g' % np.sqrt(np.mean(np.sum((gh[:, :2] - ref_xy) ** 2, axis=1))))
            out['pose_rmse_vs_oracle_m'] = allc['pose_rmse_vs_oracle_m']
            out['cpu_baseline'] = allc
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    e.close()


if __name__ == '__main__':
    main()
