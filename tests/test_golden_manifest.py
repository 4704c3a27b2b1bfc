"""Regenerate-or-fail for the committed fixtures.

tests/golden/MANIFEST.json holds one content hash per fixture, written by the generator that made it
(oracle/ref_harness/manifest.py).  Three checks, all on the CPU:
  1. the committed fixtures ARE what the manifest says: no file without an entry, no entry without a file, no drift;
  2. the reference-derived fixtures (the five trajectories, resampler and particle KATs, DR, evaluation statistics, launch
     parameters) are re-made from /root/reference where it exists (this container: 45 s) and must come out bit for bit;
     the GPU box has no reference: there check 1 stands for them;
  3. a subset of the self-oracle MBES / landmark cases (the ones that take seconds: the mesh cases brute-force every
     triangle; the grid cases sample their rays at 5 mm and take minutes -- `python oracle/ref_harness/manifest.py
     --regen-check` runs them all) is re-made anywhere and must come out bit for bit: every case draws from its own seeded
     streams, so a case added or removed cannot move another."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle', 'ref_harness'))
import manifest  # noqa: E402

QUICK_MBES = ['landmarks_knn', 'mbes_mesh_regular', 'mbes_mesh_tin', 'mbes_mesh_sweep', 'mbes_tin_sweep']


def test_committed_fixtures_are_what_the_manifest_says():
    m = manifest.load()
    files = manifest.fixtures()
    assert len(files) >= 23
    assert sorted(m) == files, (sorted(set(files) ^ set(m)))
    assert manifest.verify() == []
    for name, e in m.items():   # every entry names a generator that exists
        assert os.path.exists(os.path.join(ROOT, e['generator'])), (name, e['generator'])


def test_self_oracle_cases_regenerate_bit_for_bit(tmp_path):
    manifest.regenerate(str(tmp_path), scripts=['gen_golden_mbes.py'], mbes_cases=QUICK_MBES)
    made = manifest.fixtures(str(tmp_path))
    assert made == sorted(c + '.npz' for c in QUICK_MBES)
    assert manifest.verify(str(tmp_path), against=manifest.GOLDEN) == []


@pytest.mark.skipif(not os.path.isdir('/root/reference'), reason='the reference only exists in the build container; the manifest check stands for these fixtures elsewhere')
def test_reference_derived_fixtures_regenerate_bit_for_bit(tmp_path):
    manifest.regenerate(str(tmp_path), scripts=['gen_golden.py', 'gen_golden_dr.py', 'gen_golden_stats.py', 'gen_launch_fixture.py'])
    made = manifest.fixtures(str(tmp_path))
    committed = manifest.load()
    assert made == sorted(n for n, e in committed.items() if e['needs_reference'])
    assert manifest.verify(str(tmp_path), against=manifest.GOLDEN) == []
