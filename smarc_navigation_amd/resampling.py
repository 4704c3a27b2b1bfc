"""Mirror of auv_particle_filter/scripts/resampling.py: same function names and argument meaning;
the work runs on the GPU through mcl_resample_indices (exact integer CDF).

The reference draws its uniforms from numpy's global legacy RNG inside each function
(resampling.py:74,103,157,194); so do these, so that seeding np.random reproduces the reference's
ancestor indices."""
import numpy as np
from numpy.random import random

from . import engine as _engine


def naive_resample(weights):
    """resampling.py:116-131: base positions i/N plus ONE np.random.uniform(0, 1/N) draw (= random()/N),
    first index whose cumulative weight is >= the position.  Returns a list like the reference."""
    w = np.asarray(weights, dtype=np.float64)
    return _engine.resample_indices(w, random(), _engine.NAIVE).tolist()


def systematic_resample(weights):
    w = np.asarray(weights, dtype=np.float64)
    return _engine.resample_indices(w, random(), _engine.SYSTEMATIC)


def stratified_resample(weights):
    w = np.asarray(weights, dtype=np.float64)
    return _engine.resample_indices(w, random(len(w)), _engine.STRATIFIED)


def multinomial_resample(weights):
    w = np.asarray(weights, dtype=np.float64)
    return _engine.resample_indices(w, random(len(w)), _engine.MULTINOMIAL)


def residual_resample(weights):
    w = np.asarray(weights, dtype=np.float64)
    n = len(w)
    k = int(np.floor(n * w).astype(int).sum())
    return _engine.resample_indices(w, random(max(n - k, 0)) if n - k > 0 else np.zeros(1), _engine.RESIDUAL)
