"""CPU test: libmcl_hip.so loads without a GPU, exports every symbol include/*.h declare, the
ctypes table covers all of them, and the product path fails loudly when no device is present."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    for hdr in ('mcl.h', 'mcl_dr.h', 'mcl_map.h'):
        src = open(os.path.join(ROOT, 'include', hdr)).read()
        src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
        names |= set(re.findall(r'\b(mcl_[a-z0-9_]+)\s*\(', src))
    return sorted(names)


def test_header_symbols_exported_and_bound():
    from smarc_navigation_amd import _lib
    names = _declared()
    assert len(names) >= 30
    lib = ctypes.CDLL(_lib.SO_PATH)
    for n in names:
        assert hasattr(lib, n), 'libmcl_hip.so does not export %s' % n
        assert n in _lib.SYMBOLS, 'ctypes table misses %s' % n
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.load().mcl_abi_version() == 4


def test_config_struct_matches_header_layout():
    from smarc_navigation_amd import _lib
    # 3*8 + 6*4 + 8 + 18*8 + 8 + 16*8
    assert ctypes.sizeof(_lib.Config) == 24 + 24 + 8 + 144 + 8 + 128
    assert ctypes.sizeof(_lib.Odom) == 8 * 10
    assert ctypes.sizeof(_lib.Timing) == 15 * 16   # MCL_K_COUNT entries of (double ms, int64 launches)


def test_no_silent_fallback_without_gpu(gpu_available):
    if gpu_available:
        pytest.skip('GPU present')
    from smarc_navigation_amd import engine
    with pytest.raises(engine.MclError) as ei:
        engine.Engine(16)
    assert ei.value.status == -2
    with pytest.raises(engine.MclError):
        engine.resample_indices([0.5, 0.5], 0.3)


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, 'smarc_navigation_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                txt = open(os.path.join(dirpath, f), errors='ignore').read()
                assert 'oracle' not in txt.replace('oracle/mcl_oracle.c', '').replace('the oracle', '').replace(
                    'fp64 oracle', '').lower() or f.endswith('.h') or f.endswith('.hip'), f
                assert 'import oracle' not in txt and 'from oracle' not in txt and 'mcl_oracle.h' not in txt, f


def test_matrix_from_tf_host_helper_matches_reference_golden():
    """mcl_matrix_from_tf needs no GPU: check it against the reference's matrix_from_tf output."""
    import numpy as np
    from smarc_navigation_amd import _lib
    from tests import helpers
    g = helpers.load('particle_kat')
    lib = _lib.load()
    t = np.ascontiguousarray(g['mt_in'][:3])
    q = np.ascontiguousarray(g['mt_in'][3:])
    m = np.zeros(16)
    assert lib.mcl_matrix_from_tf(t.ctypes.data, q.ctypes.data, m.ctypes.data) == 0
    np.testing.assert_allclose(m.reshape(4, 4), g['mt_M'], rtol=0, atol=1e-15)
