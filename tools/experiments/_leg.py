import json, os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import bench
from smarc_navigation_amd import engine, synth
tin = bench.build_map('mesh-tin')
for box in ('1', '0'):
    os.environ['MCL_TIN_BOX_OUTLINE'] = box
    out = bench.run_leg(engine, 'x', tin, 1048576, 512, 50, 40)
    print('intact TIN, box outline', box, out['ms_per_step'], out['kernels'].get('mbes_main'), flush=True)
out = bench.run_leg(engine, 'x', bench.punch_gaps(tin), 1048576, 512, 50, 40)
print('gaps', out['ms_per_step'], out['kernels'].get('mbes_main'), flush=True)
out = bench.run_leg(engine, 'x', bench.punch_hole(tin, 1.0, 10.0), 1048576, 512, 50, 40)
print('hole', out['ms_per_step'], out['kernels'].get('mbes_main'), flush=True)
m = dict(tin, tris=synth.mesh_ragged(tin['verts'], tin['tris']), desc='ragged')
out = bench.run_leg(engine, 'x', m, 1048576, 512, 50, 40, m2o=synth.rigid_matrix(100.0, -330.0, 0.0, 0.0, 0.0, 0.0))
print('ragged', out['ms_per_step'], out['kernels'].get('mbes_main'), flush=True)
