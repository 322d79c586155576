"""Shared definition of the golden-vector cases.

Used by make_golden.py (which runs the *reference* in the build container) and
by the parity tests (which never touch /root/reference).  Inputs are recipes
over professad_amd.synth; fixtures store the reference's outputs plus an input
checksum so generator drift is detected.
"""
import numpy as np

from professad_amd import synth

# name -> (shape, cell recipe, density recipe)
GRID_CASES = {
    'g16r': dict(shape=(16, 16, 16), cell=('cubic', 16), den=('random', 1)),
    'g16s': dict(shape=(16, 16, 16), cell=('cubic', 16), den=('smooth', 2)),
    'g17r': dict(shape=(17, 17, 17), cell=('cubic', 17), den=('random', 3)),
    'g18t': dict(shape=(18, 20, 16), cell=('tri', 0.55), den=('random', 4)),
    'g20t': dict(shape=(20, 20, 20), cell=('tri', 0.62), den=('smooth', 5)),
    'g32r': dict(shape=(32, 32, 32), cell=('cubic', 32), den=('random', 6)),
    'gmix': dict(shape=(16, 32, 64), cell=('tri', 1.1), den=('random', 7)),
}

# cases that store the full per-term potentials (small grids only)
PER_TERM_CASES = ['g16r', 'g16s', 'g17r', 'g18t']
# cases that store only the fused configuration sums (+ closure outputs)
FUSED_CASES = ['g16r', 'g17r', 'g20t', 'g32r', 'gmix']

SINGLE_TERMS = ['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'wt', 'perrot', 'sm', 'wgc98',
                'wgc99', 'lda_x', 'pz_c', 'pw_c', 'chachiyo_c', 'pbe_x', 'pbe_c', 'lkt', 'pg1', 'pgs', 'wts_exp', 'pgsl025', 'pgslr',
                'vwgtf1', 'vwgtf2']

# fused configurations of BASELINE.json (SURVEY §8d term sets)
CONFIGS = {
    'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'lda_x', 'pz_c'],
    'cfg2': ['ion_electron', 'hartree', 'wt', 'lda_x', 'pz_c'],
    'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe_x', 'pbe_c'],
}


def make_cell(recipe):
    kind, arg = recipe
    if kind == 'cubic':
        return synth.cubic_cell(arg)
    if kind == 'tri':
        return synth.triclinic_cell(arg)
    raise ValueError(kind)


def make_inputs(name):
    """-> box [3,3], den, vext, chi, n_elec (numpy fp64)."""
    c = GRID_CASES[name]
    shape = c['shape']
    box = make_cell(c['cell'])
    kind, seed = c['den']
    if kind == 'random':
        den = synth.random_density(shape, seed=seed)
    else:
        den = synth.smooth_density(shape, seed=seed)
    vext = synth.random_potential(shape, seed=100 + seed)
    rng = np.random.default_rng(1000 + seed)
    chi = np.sqrt(den) * (1.0 + 0.05 * rng.random(shape))      # un-normalised on purpose
    vol = abs(np.linalg.det(box))
    n_elec = float(np.floor(den.mean() * vol) + 0.3)           # non-integer, far from a rounding edge
    return box, den, vext, chi, n_elec


def checksum(*arrays):
    return float(sum(np.sum(np.asarray(a, dtype=np.float64) * np.cos(np.arange(a.size).reshape(a.shape)))
                     for a in arrays))


def probe_stats(a):
    """Size-independent summary of a big grid: sum, L2, 8 fixed probes."""
    flat = a.reshape(-1)
    idx = (np.arange(8) * 2654435761 % flat.size).astype(np.int64)
    return dict(sum=float(flat.sum()), l2=float(np.sqrt((flat * flat).sum())),
                probes=[float(x) for x in flat[idx]], probe_idx=[int(i) for i in idx])
