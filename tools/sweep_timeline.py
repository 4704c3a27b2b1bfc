# tools/sweep_timeline.py [map] [steps] -- run on the GPU box: builds a -DSWEEP_TIMELINE variant of the library (every wave of
# the fan sweep records start / barrier / end on the 100 MHz device clock + its hardware slot), runs the headline loop and
# prints, for the LAST sweep launch: how full the machine is over the launch (resident waves / 8 192 slots per decile of the
# launch), how long the tail is, the spread of wave lifetimes and of the wait for the partner wave at the barrier.
import os, sys, subprocess, tempfile, ctypes
ROOT = os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, ROOT)
import numpy as np
kind = sys.argv[1] if len(sys.argv) > 1 else 'mesh'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
if 'MCL_LIB' not in os.environ:
    tmp = tempfile.mkdtemp()
    variant = os.path.join(tmp, 'tl.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-Wno-unused-function',
                           '-Wno-bitwise-instead-of-logical', '-fno-slp-vectorize', '-DSWEEP_TIMELINE'] + os.environ.get('TL_FLAGS', '').split() +
                          ['-o', variant, os.path.join(ROOT, 'smarc_navigation_amd', 'csrc', 'mcl_api.hip'), '-L/opt/rocm/lib', '-lrccl',
                           '-Wl,-rpath,/opt/rocm/lib'])
    env = dict(os.environ, MCL_LIB=variant)
    sys.exit(subprocess.call([sys.executable, __file__] + sys.argv[1:], env=env))
import bench
from smarc_navigation_amd import engine, synth, _lib
m = bench.build_map(kind)
B, P = 512, 1 << 20
base = 200
stream = synth.odom_stream(base)
ba = synth.beam_angles(B)
ranges = bench.make_ranges(engine, m, stream['truth'], ba, bench.SIGMA, bench.R_MAX)
e = engine.Engine(P, seed=5, **bench.COV)
bench.attach_map(e, m)
e.init_particles()
for k in range(steps):
    j = k % base
    e.step_mbes(stream['v'][j], stream['wz'][j], stream['q'][j], stream['z'][j], stream['dt'], ranges[j], ba, bench.SIGMA, bench.R_MAX)
e.sync()
lib = ctypes.CDLL(os.environ['MCL_LIB'])
nw = 2 * P // 64
buf = np.zeros((nw, 6), dtype=np.uint64)
assert lib.mcl_debug_sweep_timeline(buf.ctypes.data_as(ctypes.c_void_p), nw) == 0
t0, tb, t1, hw, c0, c1 = (buf[:, i].astype(np.int64) for i in range(6))
o = t0.min()
s, b, f = (t0 - o) / 100.0, (tb - o) / 100.0, (t1 - o) / 100.0   # microseconds
T = f.max()
print('%s: %d waves, launch %.1f us (first wave start -> last wave end)' % (kind, nw, T))
life = f - s
print('wave lifetime us: mean %.1f  p5 %.1f  p50 %.1f  p95 %.1f  max %.1f' % (life.mean(), *np.percentile(life, [5, 50, 95]), life.max()))
wait = f - b
print('from reaching the barrier to the end (wait for the partner + epilogue) us: mean %.1f  p50 %.1f  p95 %.1f  max %.1f ; share of lifetime %.3f'
      % (wait.mean(), *np.percentile(wait, [50, 95]), wait.max(), wait.sum() / life.sum()))
print('slot-time used: %.3f of 8192 slots x launch' % (life.sum() / (8192 * T)))
# resident waves over time
grid = np.linspace(0, T, 41)
res = [(np.sum((s <= t) & (f > t))) for t in grid]
print('resident waves at t/T = 0, .025, ... :', ' '.join('%d' % r for r in res))
# when does the last wave START, and how many slots are empty after that
last_start = s.max()
print('last wave starts at %.1f us (%.3f of the launch); tail = %.1f us' % (last_start, last_start / T, T - last_start))
tail_t = np.linspace(last_start, T, 11)
print('resident waves over the tail:', ' '.join('%d' % np.sum((s <= t) & (f > t)) for t in tail_t))
# busy slot-time lost in the tail: sum over slots of (T - end) for the waves that are the last on their slot ~ integral of empties
empt = np.trapezoid([8192 - np.sum((s <= t) & (f > t)) for t in np.linspace(last_start, T, 201)], np.linspace(last_start, T, 201))
print('empty slot-time in the tail: %.0f slot-us = %.3f of the launch' % (empt, empt / (8192 * T)))
# lifetime against the index of the wave (are late workgroups heavier?)
q = np.array_split(np.arange(nw), 8)
print('mean lifetime by eighth of the grid:', ' '.join('%.1f' % life[i].mean() for i in q))
print('mean start time by eighth of the grid:', ' '.join('%.1f' % s[i].mean() for i in q))
xcc = (hw >> 32) & 0xf
print('waves per XCC:', np.bincount(xcc.astype(int), minlength=8))
print('last end per XCC us:', ' '.join('%.1f' % f[xcc == x].max() for x in range(8) if np.any(xcc == x)))
ghz = (c1 - c0) / np.maximum(t1 - t0, 1) / 10.0
print('shader clock over a wave lifetime (s_memtime ticks / s_memrealtime): mean %.3f GHz  p5 %.3f  p95 %.3f' % (ghz.mean(), *np.percentile(ghz, [5, 95])))
