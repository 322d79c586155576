"""Deterministic synthetic inputs (cells, densities, external potentials).

These are the build's own generators: the golden-vector
script, the parity tests and bench.py all draw their inputs from here so that
a fixture only needs to store the seed/recipe and the reference's outputs.
Everything is numpy fp64; numpy's ``default_rng`` (PCG64) stream is stable
across numpy versions by policy.
"""
import numpy as np

# fcc-Al conventional cell: 4 atoms x 16.8 A^3 (reference tests/test_den_opt.py:45,
# crystal_tools 'fcc-c'); bohr per angstrom as in system.py:27-33 (CODATA-2018)
BOHR_PER_ANGSTROM = 1.0 / 0.529177210903
A_FCC_AL_BOHR = (4 * 16.8) ** (1.0 / 3.0) * BOHR_PER_ANGSTROM  # 7.6752...


def cubic_cell(n_or_len, per32=True):
    """Cubic box_vecs [3,3] (rows = lattice vectors, bohr).

    ``per32=True``: edge = a_fcc-Al * (N/32), i.e. one 4-atom conventional cell
    per 32^3 block of grid points."""
    a = A_FCC_AL_BOHR * (n_or_len / 32.0) if per32 else float(n_or_len)
    return np.eye(3) * a


def triclinic_cell(scale=1.0):
    """A fixed, well-conditioned triclinic cell (bohr)."""
    return scale * np.array([[7.9, 0.4, -0.3],
                             [0.7, 8.3, 0.5],
                             [-0.2, 0.6, 7.4]])


def random_density(shape, seed=1234, n0=0.03, amp=0.2):
    """n = n0 (1 + amp U[0,1)) -- full-spectrum content (exercises Nyquist)."""
    rng = np.random.default_rng(seed)
    return n0 * (1.0 + amp * rng.random(tuple(shape)))


def smooth_density(shape, seed=20240601, n0=0.03, amp=0.3, nwaves=8, kmax=3):
    """Smooth positive density: n0 (1 + amp * sum_m a_m cos(2 pi g_m . f + phi_m))."""
    rng = np.random.default_rng(seed)
    f = np.meshgrid(*[np.arange(s) / s for s in shape], indexing='ij')
    acc = np.zeros(tuple(shape))
    wsum = 0.0
    for _ in range(nwaves):
        g = rng.integers(-kmax, kmax + 1, size=3)
        a = rng.random()
        phi = 2 * np.pi * rng.random()
        acc += a * np.cos(2 * np.pi * (g[0] * f[0] + g[1] * f[1] + g[2] * f[2]) + phi)
        wsum += a
    return n0 * (1.0 + amp * acc / wsum)


def random_potential(shape, seed=4321, amp=0.5):
    """Smooth-ish external potential stand-in (Ha)."""
    rng = np.random.default_rng(seed)
    f = np.meshgrid(*[np.arange(s) / s for s in shape], indexing='ij')
    v = np.zeros(tuple(shape))
    for _ in range(6):
        g = rng.integers(-2, 3, size=3)
        v += rng.standard_normal() * np.cos(2 * np.pi * (g[0] * f[0] + g[1] * f[1] + g[2] * f[2])
                                            + 2 * np.pi * rng.random())
    return amp * v / 6.0 + 0.05 * rng.standard_normal(tuple(shape))


def tile_periodic(a32, n):
    """Tile a 32^3 periodic field (N/32)^3 times."""
    r = n // a32.shape[0]
    assert r * a32.shape[0] == n
    return np.tile(a32, (r, r, r))


def perturbed(den, box, n_elec, seed=20240601, rel=1e-3):
    """den * (1 + rel * low-|k| cosine mix), renormalised to n_elec (SURVEY §8d option A)."""
    shape = den.shape
    rng = np.random.default_rng(seed)
    f = np.meshgrid(*[np.arange(s) / s for s in shape], indexing='ij')
    acc = np.zeros(shape)
    for _ in range(8):
        g = rng.integers(-4, 5, size=3)
        acc += rng.random() * np.cos(2 * np.pi * (g[0] * f[0] + g[1] * f[1] + g[2] * f[2])
                                     + 2 * np.pi * rng.random())
    out = den * (1.0 + rel * acc)
    vol = abs(np.linalg.det(box))
    return out * (n_elec / (out.mean() * vol))


def bench_inputs(n, golden_dir=None, rank=0):
    """Inputs of the BASELINE workload on an n^3 grid (SURVEY.md §8d option A) -> box, chi, v_ext, N_e, description.

    n a multiple of 32 and the config-1 fixture present: the reference's converged fcc-Al 32^3 density tiled (n/32)^3
    times plus the seeded low-|k| perturbation; otherwise the full-spectrum random density (option B)."""
    import os
    fx = os.path.join(golden_dir, 'cfg1_fccAl_32.npz') if golden_dir else None
    if fx and os.path.exists(fx) and n % 32 == 0:
        d = np.load(fx)
        r = n // 32
        box = d['box'] * r
        n_elec = float(d['n_elec']) * r ** 3
        den = tile_periodic(d['den'], n)
        vext = tile_periodic(d['vext'], n)
        den = perturbed(den, box, n_elec, seed=20240601 + rank)
        src = 'converged fcc-Al 32^3 fixture tiled %d^3 + 1e-3 low-|k| perturbation' % r
    else:
        box = cubic_cell(n)
        den = random_density((n, n, n), seed=1234 + rank)
        vext = random_potential((n, n, n), seed=77)
        n_elec = float(round(den.mean() * abs(np.linalg.det(box))))
        src = 'n0(1+0.2U) random density'
    return box, np.sqrt(den), vext, n_elec, src
