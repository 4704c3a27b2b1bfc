#!/usr/bin/env python3
"""One extra leg of bench.py alone (profiling runs): tools/experiments/leg.py config5|config4|config4_rccl|tin_hole|tin_ragged [steps]"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from smarc_navigation_amd import engine, synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'config5'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
mesh = bench.build_map('mesh')
kw = dict(P=524288, B=512, steps=steps, warmup=5)
if name == 'config5':
    kw['landmarks'] = (synth.landmark_map(4096, (-64.0, -354.0, 643.0, 353.0)), 16)
if name == 'config4_rccl':   # the sharded pipeline over a 1-rank RCCL communicator (bench.py: config4_shard_rccl_1rank)
    kw['rccl_1rank'] = True
if name == 'tin_hole':       # the irregular TIN with a data gap under the swath (bench.py: mesh_tin_hole_under_swath): k_mbes_sweep<6, ...>
    mesh = bench.punch_hole(bench.build_map('mesh-tin'), 1.0, 10.0)
    kw.update(P=1048576, warmup=40)
if name == 'tin_ragged':     # ... with a ragged outline and bays, the track 24 m inside it (bench.py: mesh_tin_ragged_outline)
    tin = bench.build_map('mesh-tin')
    mesh = dict(tin, tris=synth.mesh_ragged(tin['verts'], tin['tris']), desc=tin['desc'] + ', ragged outline')
    kw.update(P=1048576, warmup=40, m2o=synth.rigid_matrix(100.0, -330.0, 0.0, 0.0, 0.0, 0.0))
out = bench.run_leg(engine, name, mesh, kw.pop('P'), kw.pop('B'), kw.pop('steps'), kw.pop('warmup'), **kw)
print(json.dumps({k: out[k] for k in ('ms_per_step', 'kernels')}))
