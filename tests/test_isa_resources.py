"""Register / scratch budget of the hot-path kernels (tools/isa.sh: hipcc -S for gfx950 + kernel-resource-usage
remarks; no GPU needed).  A kernel of the filter step that spills to scratch pays HBM traffic the roofline accounting
does not know about: none of them may.  The table is also kept under profiles/ (r04_isa_resources.tsv)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ('k_mbes_sweep', 'k_mbes_fast', 'k_mbes_cast', 'k_mbes_pose', 'k_mbes_classify', 'k_predict', 'k_quantise_tiles',
       'k_quantise_shard', 'k_shift_scan', 'k_visit_scan', 'k_cdf_expand', 'k_resample_gather', 'k_pack_dupes', 'k_offspring_cdf', 'k_gps_logw', 'k_landmark_update',
       'k_mean_partial', 'k_cov_partial', 'k_max_slots')


@pytest.fixture(scope='module')
def table(tmp_path_factory):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    out = str(tmp_path_factory.mktemp('isa'))
    subprocess.check_call([os.path.join(ROOT, 'tools', 'isa.sh'), out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    rows = {}
    with open(os.path.join(out, 'resources.tsv')) as f:
        next(f)
        for line in f:
            name, sgpr, vgpr, scratch, lds, occ = line.rstrip('\n').split('\t')
            rows[name] = dict(vgpr=int(vgpr), scratch=int(scratch), lds=int(lds), occ=int(occ))
    with open(os.path.join(out, 'mcl.s')) as f:
        asm = f.read()
    return rows, asm


def test_no_hot_path_kernel_spills_to_scratch(table):
    rows, _ = table
    hot = {k: v for k, v in rows.items() if any(h in k for h in HOT)}
    assert len(hot) > 40   # every instantiation was seen
    spilling = {k: v['scratch'] for k, v in hot.items() if v['scratch'] > 0}
    assert not spilling, spilling


def test_headline_sweep_keeps_eight_waves_per_simd_and_uses_no_matrix_cores(table):
    rows, asm = table
    for surf in (2, 3):   # the two diagonals of a triangulated DEM: the same workload
        r = rows['void k_mbes_sweep<%d, false, false>' % surf]
        assert r['vgpr'] <= 64 and r['occ'] == 8 and r['scratch'] == 0, r
    # the general TIN's walk keeps the same occupancy, and so does its variant that goes on through holes and over the outline
    # (SURF 6; the rim search carries three registers beside the walk's state and recomputes the cut it kept: at 75 registers
    # and 6 waves per SIMD the launch was 11 % slower -- this walk waits for a dependent load on every step)
    for surf in (5, 6):
        r = rows['void k_mbes_sweep<%d, false, false>' % surf]
        assert r['vgpr'] <= 64 and r['occ'] == 8 and r['scratch'] == 0, r
    assert 'v_mfma' not in asm   # nothing on this path is a dense contraction (north_star)
    assert 'v_pk_fma_f32' not in asm and 'v_pk_add_f32' not in asm   # -fno-slp-vectorize: packed f32 holds the SIMD twice
