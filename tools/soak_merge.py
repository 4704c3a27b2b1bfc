# tools/soak_merge.py <steps> -- run on the GPU box: the headline loop (1 M x 512, mesh and TIN) for <steps> fused steps, three times --
# the assembly merge loop twice and a -DSWEEP_MERGE_CXX=1 build once -- final state and mean history compared bit for bit.
import os, sys, subprocess, tempfile
ROOT = os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, ROOT)
import numpy as np
if len(sys.argv) > 2:
    import bench
    from smarc_navigation_amd import engine, synth
    out, steps, kind = sys.argv[1], int(sys.argv[2]), sys.argv[3]
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
    np.savez(out, state=e.get_particles(), hist=e.mean_history(min(steps, 4000)), path=np.array(e.mbes_last_path()))
else:
    steps = int(sys.argv[1])
    tmp = tempfile.mkdtemp()
    variant = os.path.join(tmp, 'cxx.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-Wno-unused-function',
                           '-Wno-bitwise-instead-of-logical', '-fno-slp-vectorize', '-DSWEEP_MERGE_CXX=1', '-o', variant,
                           os.path.join(ROOT, 'smarc_navigation_amd', 'csrc', 'mcl_api.hip'), '-L/opt/rocm/lib', '-lrccl', '-Wl,-rpath,/opt/rocm/lib'])
    for kind in ('mesh', 'mesh-tin'):
        res = {}
        for name, lib in (('asm1', None), ('asm2', None), ('cxx', variant)):
            env = dict(os.environ)
            if lib: env['MCL_LIB'] = lib
            out = os.path.join(tmp, name + '.npz')
            subprocess.check_call([sys.executable, __file__, out, str(steps), kind], env=env)
            res[name] = np.load(out)
        a, b, c = res['asm1'], res['asm2'], res['cxx']
        print(kind, steps, 'steps: asm run 1 == run 2:', np.array_equal(a['state'], b['state']) and np.array_equal(a['hist'], b['hist']),
              '| asm == compiler loop:', np.array_equal(a['state'], c['state']) and np.array_equal(a['hist'], c['hist']),
              '| finite:', bool(np.isfinite(a['state']).all()), '| path', a['path'], '| mean xy', a['hist'][-1][:2], flush=True)
