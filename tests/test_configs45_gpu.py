"""BASELINE configs 4 and 5 at their full sizes on ONE GPU (the box of the -m gpu tier has one MI355X, 288 GB):

  config 4   512^3 fp64, WGC99 + PBE, slab-decomposed over 8 ranks  -> the 8-rank geometry emulated with 8 contexts
  config 5   1024^3 fp32, 8 ranks                                   -> the same emulation on the fp32 build, and the
                                                                       whole 1024^3 grid in one context (extensivity)

The emulation (tests/local_ranks.py) runs the very kernels, exchange-buffer layouts and stage order an 8-GPU job runs;
only the transport differs (device copies instead of RCCL).  Inputs are generated on the device (a 1024^3 grid would
need ~35 GB of host arrays through the numpy recipes); the comparisons are engine against engine and against the
32^3 fp64 result through periodic tiling, so no oracle run at these sizes is needed.
"""
import numpy as np
import pytest
import torch

from local_ranks import LocalRanks
from professad_amd import synth
from professad_amd.engine import Engine

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')
CFG3 = ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']
CFG2 = ['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'lda_x', 'pz_c']


def _device_inputs(shape, dtype, seed):
    """rough positive density (as chi), external potential, N_e -- generated on the GPU"""
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    den = 0.03 * (1.0 + 0.2 * torch.rand(shape, generator=g, device=DEV, dtype=dtype))
    chi = torch.sqrt(den) * (1.0 + 0.1 * torch.rand(shape, generator=g, device=DEV, dtype=dtype))
    del den
    vext = 0.05 * torch.randn(shape, generator=g, device=DEV, dtype=dtype)
    return chi, vext


def _compare(E, Er, mu, mur, g, gr, e_tol, g_tol):
    for k in Er:
        assert abs(E[k] - Er[k]) <= e_tol * max(abs(Er[k]), 1e-3), (k, E[k], Er[k])
    assert abs(mu - mur) <= e_tol * max(1.0, abs(mur))
    scale = float(gr.abs().max())
    err = 0.0
    for x in range(0, g.shape[0], 64):                      # in x slices: no third grid-sized temporary
        err = max(err, float((g[x:x + 64] - gr[x:x + 64]).abs().max()))
    assert err <= g_tol * scale, (err, scale)


def test_config4_eight_rank_slab_geometry_512_fp64():
    """512^3 fp64, IonElectron + Hartree + WGC99 + PBE: 8 emulated slab ranks against one engine, 1e-12"""
    n = 512
    shape = (n, n, n)
    box = torch.as_tensor(synth.triclinic_cell(n / 16.0))
    chi, vext = _device_inputs(shape, torch.double, 4)
    n_elec = float(np.floor(0.033 * abs(np.linalg.det(box.numpy()))) + 0.3)
    ref = Engine(shape, DEV).set_cell(box).set_terms(CFG3)
    Er, mur, gr = ref.energy_grad_chi(chi, n_elec, vext)
    assert ref.fast_path and int(ref.query(0)) == 23          # the fused pipeline, 23 transforms
    ref.close()
    loc = LocalRanks(shape, DEV, 8).set_cell(box).set_terms(CFG3)
    assert loc.st[0].nchunks == 4          # the automatic choice at this size: every exchange travels as four kz chunks
    E, mu, g = loc.closure(chi, n_elec, vext)
    assert int(loc.st[0].query(0)) == 23   # ... and the chunked sequence executes the same 23 transforms
    loc.close()
    _compare(E, Er, mu, mur, g, gr, 1e-12, 1e-12)


def test_config5_eight_rank_slab_geometry_1024_fp32():
    """1024^3 fp32 (the fp32 build), config-2 terms: 8 emulated slab ranks against one engine (fp32 round-off only)"""
    n = 1024
    shape = (n, n, n)
    box = torch.as_tensor(synth.cubic_cell(n))
    chi, vext = _device_inputs(shape, torch.float32, 5)
    n_elec = float(np.floor(0.033 * abs(np.linalg.det(box.numpy()))) + 0.3)
    ref = Engine(shape, DEV, dtype=torch.float32).set_cell(box).set_terms(CFG2)
    Er, mur, gr = ref.energy_grad_chi(chi, n_elec, vext)
    assert ref.fast_path
    ref.close()
    torch.cuda.empty_cache()
    loc = LocalRanks(shape, DEV, 8, dtype=torch.float32).set_cell(box).set_terms(CFG2)
    assert loc.st[0].nchunks == 4
    E, mu, g = loc.closure(chi, n_elec, vext)
    loc.close()
    _compare(E, Er, mu, mur, g, gr, 5e-6, 5e-4)


@pytest.mark.parametrize('terms', [CFG2, CFG3], ids=['cfg2', 'cfg3'])
def test_config5_whole_1024_grid_fp32_is_the_tiled_fp64_32_cube(terms):
    """Single context, 1024^3 fp32: a 32^3 state tiled 32^3 times on the 32x cell.  Every term is extensive, so the
    energy is 32768 x the fp64 32^3 energy and the potential is the tiled 32^3 potential -- the fp32 engine at the
    config-5 size pinned to the fp64 engine (itself pinned to the reference's goldens) without a reference run."""
    base, n = 32, 1024
    r = n // base
    box32 = synth.cubic_cell(base)
    den32 = synth.smooth_density((base,) * 3, seed=21) * (1 + 0.02 * np.random.default_rng(8).random((base,) * 3))
    vext32 = synth.random_potential((base,) * 3, seed=22)
    den32 *= 3.0 / (den32.mean() * abs(np.linalg.det(box32)))     # integer N_e: WGC99's rounding commutes with tiling
    small = Engine((base,) * 3, DEV).set_cell(torch.as_tensor(box32)).set_terms(terms)
    d32, v32 = torch.as_tensor(den32, device=DEV), torch.as_tensor(vext32, device=DEV)
    E32, pot32 = small.energy_potential(d32, v32)
    small.close()
    big = Engine((n,) * 3, DEV, dtype=torch.float32).set_cell(torch.as_tensor(synth.cubic_cell(n))).set_terms(terms)
    denN = d32.float().repeat(r, r, r)
    vN = v32.float().repeat(r, r, r)
    EN, potN = big.energy_potential(denN, vN)
    assert big.fast_path
    big.close()
    for k in E32:
        assert abs(EN[k] - r ** 3 * E32[k]) <= 5e-6 * max(abs(EN[k]), 1.0), (k, EN[k], r ** 3 * E32[k])
    scale = float(pot32.abs().max())
    for sl in ((slice(0, base),) * 3, (slice(n - base, n), slice(base, 2 * base), slice(n - base, n))):
        assert float((potN[sl].double() - pot32).abs().max()) <= 5e-4 * scale


def _al_table():
    """the al.gga recpot table (DATA of the reference's tests, committed as tests/golden/recpots.npz)"""
    import os
    import cases
    from professad_amd.ions import recpot_table
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(cases.__file__)), 'recpots.npz'))
    return recpot_table(g['al_raw'], float(g['al_kmax']))


@pytest.mark.parametrize('n', [256, 1024])
def test_config5_pme_potential_and_forces_at_scale_by_periodic_tiling(n):
    """Particle-mesh Ewald with the ion count of config 5: the fcc-Al conventional cell (4 ions, 32^3 grid) repeated
    (n/32)^3 times -- 131 072 ions on the 1024^3 grid.  The tiled system is exactly periodic, so its ionic potential is the
    tiled 32^3 potential and every ion feels the force of its image in the small cell: PME spreading (atomics), the
    b-factor multiply, both transforms and the force gather at full size, pinned without a reference run."""
    from professad_amd.ions import ion_electron_forces, ionic_potential
    base, order = 32, 10
    r = n // base
    tab = _al_table()
    frac32 = np.array([[0.0, 0.0, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5], [0.5, 0.5, 0.0]]) + 0.013      # off the grid points
    box32 = synth.cubic_cell(base)
    small = Engine((base,) * 3, DEV)
    v32 = ionic_potential(small, box32, [(frac32, tab)], pme_order=order)
    den32 = torch.as_tensor(synth.smooth_density((base,) * 3, seed=9, n0=0.03, amp=0.4), device=DEV)
    F32_ = ion_electron_forces(small, box32, den32, [(frac32, tab)], pme_order=order)[0]
    small.close()
    shifts = np.stack(np.meshgrid(np.arange(r), np.arange(r), np.arange(r), indexing='ij'), -1).reshape(-1, 1, 3)
    frac = ((frac32[None, :, :] + shifts) / r).reshape(-1, 3)
    assert frac.shape[0] == 4 * r ** 3
    big = Engine((n,) * 3, DEV)
    boxN = synth.cubic_cell(n)
    vN = ionic_potential(big, boxN, [(frac, tab)], pme_order=order)
    scale = float(v32.abs().max())
    for sl in ((slice(0, base),) * 3, (slice(n - base, n), slice(base, 2 * base), slice(n - base, n))):
        assert float((vN[sl] - v32).abs().max()) <= 1e-9 * scale
    del vN
    denN = den32.repeat(r, r, r)
    FN = ion_electron_forces(big, boxN, denN, [(frac, tab)], pme_order=order)[0].reshape(-1, 4, 3)
    big.close()
    assert np.abs(FN - F32_[None]).max() <= 1e-9 * max(np.abs(F32_).max(), 1e-3)


def test_config5_stress_of_the_1024_fp32_grid_is_intensive():
    """Config 5's stress at full size: an fp32 engine on the 1024^3 grid hands `stress` to its fp64 sibling on the widened
    density (professad_amd.engine.Engine.stress; 25 GB of fp64 workspace).  The tiled cell has the stress tensors of the
    32^3 cell, every term of config 2 -- pinned to the fp64 32^3 engine (itself pinned to the reference's get_stress goldens)."""
    base, n = 32, 1024
    r = n // base
    box32 = synth.cubic_cell(base)
    den32 = synth.smooth_density((base,) * 3, seed=21) * (1 + 0.02 * np.random.default_rng(8).random((base,) * 3))
    den32 *= 3.0 / (den32.mean() * abs(np.linalg.det(box32)))
    small = Engine((base,) * 3, DEV).set_cell(torch.as_tensor(box32)).set_terms(CFG2)
    d32 = torch.as_tensor(den32, device=DEV)
    s32 = small.stress(d32)
    small.close()
    big = Engine((n,) * 3, DEV, dtype=torch.float32).set_cell(torch.as_tensor(synth.cubic_cell(n))).set_terms(CFG2)
    sN = big.stress(d32.float().repeat(r, r, r))
    big.close()
    from professad_amd.engine import _ENGINES
    for key in [k for k in _ENGINES if k[0] == (n, n, n)]:          # the cached fp64 sibling: give its 25 GB back
        _ENGINES.pop(key).close()
    torch.cuda.empty_cache()
    for k in s32:
        scale = max(float(np.abs(s32[k]).max()), 1e-6)
        assert float(np.abs(sN[k] - s32[k]).max()) <= 2e-6 * scale, (k, sN[k], s32[k])      # the density was rounded to fp32 on the way
