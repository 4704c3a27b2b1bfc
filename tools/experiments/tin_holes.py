#!/usr/bin/env python3
"""What a hole costs the TIN sweep (round 6: crossed by its rim records; before: 48 x): the irregular TIN of the bench with the triangles around (1.0, Y) missing -- under the
swath of the timed steps (a slice that runs into a hole ends the walk: the particle is handed over; with a collapsed cloud
every particle of the ping is).  tools/experiments/tin_holes.py [Y ...]   (Y = none: the intact mesh)"""
import json
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
import bench  # noqa: E402
from smarc_navigation_amd import engine  # noqa: E402

for y in (sys.argv[1:] or ['none', '10', '30']):
    m = bench.build_map('mesh-tin')
    if y != 'none':
        m = bench.punch_hole(m, 1.0, float(y))
    # (both switches are read when the map / the handle is made) rims: mesh_build links the hole's rim and the walk crosses it;
    #  without: the sweep hands over, to the fan slice (handover_slice 1) or to the ray traversal (0: rounds 1-5)
    for rims, ho in ([('1', '1'), ('0', '1'), ('0', '0')] if y != 'none' else [('1', '1')]):
        os.environ['MCL_TIN_RIMS'] = rims
        os.environ['MCL_HANDOVER_SLICE'] = ho
        out = bench.run_leg(engine, 'tin_holes', m, 1048576, 512, 30, 40)
        print(json.dumps({'hole_at_y': y, 'rims': rims, 'handover_slice': ho, 'ms_per_step': out['ms_per_step'],
                          'to_slice': out['mbes_path']['particles_handed_to_fan_slice'], 'to_traversal': out['mbes_path']['particles_handed_to_traversal'],
                          'kernels': out['kernels']}), flush=True)
