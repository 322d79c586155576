"""GPU parity tests of the fp32 build (libofdft_hip_f32.so, BASELINE config 5).

The reference is fp64 only, so parity here is "fp32 engine against the reference's fp64 goldens / the fp64 oracle" at a
stated looser tolerance.  The inputs themselves are rounded to fp32 (relative 6e-8), energy sums are accumulated in
fp64 by the kernels; measured: energies within 1e-6 relative, potentials / gradients within 1e-4 of their maximum
(the vW term's Laplacian amplifies the FFT round-off by k^2).  Tolerances used below: 5e-6 and 5e-4.
"""
import os

import numpy as np
import pytest
import torch

import cases
from lbfgs_double import NumpyLbfgsBackend
from local_ranks import LocalRanks
from professad_amd import functionals as F
from professad_amd import synth
from professad_amd.engine import Engine
from professad_amd.optimize import HipLbfgsBackend, optimize_density

pytestmark = pytest.mark.gpu
GOLDEN = os.path.dirname(os.path.abspath(cases.__file__))
DEV = 'cuda:0'
E_RTOL = 5e-6
V_RTOL = 5e-4
F32 = torch.float32


def dev32(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=F32, device=DEV)


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    a, b = a.astype(np.complex128 if np.iscomplexobj(a) else np.float64), b.astype(np.complex128 if np.iscomplexobj(b) else np.float64)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


@pytest.mark.parametrize('shape', [(8, 8, 16), (16, 32, 64), (64, 64, 64), (256, 8, 32), (8, 1024, 16), (8, 8, 2048),
                                   (128, 128, 128), (18, 20, 16), (5, 6, 7), (17, 17, 17), (3, 5, 255), (255, 3, 5), (7, 129, 67),
                                   (48, 96, 120), (144, 160, 192), (240, 250, 270), (288, 48, 320), (384, 96, 480), (96, 480, 144)])
def test_rfftn_irfftn_match_numpy_f32(shape):
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(np.float32)
    eng = Engine(shape, DEV, dtype=F32)
    got = eng.rfftn(dev32(x))
    assert got.dtype == torch.complex64
    ref = np.fft.rfftn(x.astype(np.float64))
    assert relerr(got.cpu().numpy(), ref) < 2e-6
    yk = (rng.standard_normal(ref.shape) + 1j * rng.standard_normal(ref.shape)).astype(np.complex64)
    got_r = eng.irfftn(torch.as_tensor(yk, device=DEV)).cpu().numpy()
    assert relerr(got_r, np.fft.irfftn(yk.astype(np.complex128), s=shape, axes=(0, 1, 2))) < 2e-6
    mixed = (48, 96, 120, 144, 160, 192, 240, 250, 270, 288, 320, 384, 480)          # mixed-radix plans: in the fp32 build since round 3
    assert eng.fast_path == all((s & (s - 1)) == 0 or s in mixed for s in shape)
    with pytest.raises(TypeError):
        eng.rfftn(torch.zeros(shape, dtype=torch.double, device=DEV))       # an fp32 engine takes fp32 tensors only
    eng.close()


_CFG_TERMS = {
    'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'pz'],
    'cfg2': ['ion_electron', 'hartree', 'wt', 'pz'],
    'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe'],
}


@pytest.mark.parametrize('case', cases.FUSED_CASES)
@pytest.mark.parametrize('pipeline', [0, 1])
def test_f32_engine_against_reference_goldens(case, pipeline):
    """energy + potential and the optimize_density closure of configs 1-3 against the REFERENCE's fp64 outputs"""
    gold = np.load(os.path.join(GOLDEN, 'fused_%s.npz' % case))
    box, den, vext, chi, n_elec = cases.make_inputs(case)
    eng = Engine(den.shape, DEV, dtype=F32).set_cell(torch.as_tensor(box)).set_option(0, pipeline)
    for cfg, names in _CFG_TERMS.items():
        eng.set_terms(F.NativeTerms(names).names)
        E, v = eng.energy_potential(dev32(den), dev32(vext))
        Eref = float(gold['E_' + cfg])
        assert v.dtype == F32
        assert abs(sum(E.values()) - Eref) <= E_RTOL * max(1.0, abs(Eref)), (cfg, sum(E.values()), Eref)
        assert relerr(v.cpu().numpy(), gold['v_' + cfg]) < V_RTOL, cfg
        Et, mu, g = eng.energy_grad_chi(dev32(chi), n_elec, dev32(vext))
        Ecl = float(gold['Ec_' + cfg])
        assert abs(sum(Et.values()) - Ecl) <= E_RTOL * max(1.0, abs(Ecl))
        assert relerr(g.cpu().numpy(), gold['g_' + cfg]) < V_RTOL
    eng.close()


def test_f32_tracks_f64_engine_at_256():
    """config-5 style terms (config 2's, Wang-Teter + LDA) and config 3 on the bench density at 256^3"""
    n = 256
    shape = (n, n, n)
    box = torch.as_tensor(synth.cubic_cell(n))
    den = synth.smooth_density(shape, seed=3)
    vext = synth.random_potential(shape, seed=4)
    chi = np.sqrt(den)
    nel = float(round(den.mean() * abs(np.linalg.det(box.numpy()))))
    e64 = Engine(shape, DEV).set_cell(box)
    e32 = Engine(shape, DEV, dtype=F32).set_cell(box)
    for names in (['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'lda_x', 'pz_c'],
                  ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']):
        e64.set_terms(names)
        e32.set_terms(names)
        Ea, mua, ga = e64.energy_grad_chi(torch.as_tensor(chi, device=DEV), nel, torch.as_tensor(vext, device=DEV))
        Eb, mub, gb = e32.energy_grad_chi(dev32(chi), nel, dev32(vext))
        for k in Ea:
            assert abs(Ea[k] - Eb[k]) <= E_RTOL * max(abs(Ea[k]), 1e-3), (k, Ea[k], Eb[k])
        assert abs(mua - mub) < 1e-6
        assert relerr(gb.cpu().numpy(), ga.cpu().numpy()) < V_RTOL
        assert e32.fast_path
    # the per-geometry-step quantities of an fp32 engine are formed by the fp64 routines from the widened density
    s32 = e32.stress(dev32(den))
    s64 = e64.stress(torch.as_tensor(den.astype(np.float32).astype(np.float64), device=DEV))
    for k in s64:
        assert np.allclose(s32[k], s64[k], rtol=0, atol=1e-12 * max(1.0, np.abs(s64[k]).max()))
    e64.close()
    e32.close()


@pytest.mark.parametrize('n', [4096, 1001, 262147])
def test_f32_lbfgs_sweeps_match_numpy_double(n):
    """the device L-BFGS sweeps on fp32 vectors (dot products accumulated in fp64) against the numpy double statement of
    their contract, fed the same fp32-rounded vectors: same scripted sequence as the fp64 test"""
    rng = np.random.default_rng(n)
    hip, ref = HipLbfgsBackend(n, 8, DEV, F32), NumpyLbfgsBackend(n, 8)
    x = rng.standard_normal(n).astype(np.float32).astype(np.float64)
    xd = dev32(x)
    for it in range(12):
        g = (rng.standard_normal(n) * (1.0 + it)).astype(np.float32).astype(np.float64)
        gd = dev32(g)
        va, ka = hip.dots(gd)
        vb, kb = ref.dots(g)
        assert ka == kb and va.shape == vb.shape
        assert np.abs(va - vb).max() <= 2e-5 * max(1.0, np.abs(vb).max()), (it, np.abs(va - vb).max())
        push = it > 0 and it != 5
        hip.commit(push)
        ref.commit(push)
        k = len(ref.S)
        cs, cy, cg, t = rng.standard_normal(k), rng.standard_normal(k), -0.7, 0.1 + 0.01 * it
        sa = hip.update(cs, cy, cg, t, xd, gd)
        sb = ref.update(cs, cy, cg, t, x, g)
        assert abs(sa - sb) <= 2e-5 * max(1.0, abs(sb))
        assert np.abs(xd.cpu().numpy() - x).max() <= 2e-5 * max(1.0, np.abs(x).max())
    with pytest.raises(ValueError):          # an fp32 optimiser refuses fp64 vectors
        hip.dots(torch.zeros(n, dtype=torch.double, device=DEV))
    hip.close()


def test_f32_density_optimisation_reaches_the_f64_minimum():
    """config-1 cell: the fp32 inner loop lands on the fp64 energy to fp32 accuracy"""
    box, den, vext, chi, n_elec = cases.make_inputs(cases.FUSED_CASES[0])
    names = F.NativeTerms(['ion_electron', 'hartree', 'tf', 'vw', 'pz']).names
    vol = float(abs(np.linalg.det(box)))
    out = {}
    for dt in (torch.double, F32):
        eng = Engine(den.shape, DEV, dtype=dt).set_cell(torch.as_tensor(box)).set_terms(names)
        r = optimize_density(eng, n_elec, torch.as_tensor(vext, dtype=dt, device=DEV), volume=vol, ntol=1e-5, n_maxiter=200)
        assert r['converged']
        assert r['chi'].dtype == dt
        out[dt] = r['E_Ha']
        eng.close()
    assert abs(out[F32] - out[torch.double]) <= 2e-5 * abs(out[torch.double])


@pytest.mark.parametrize('ranks', [2, 8])
def test_f32_slab_decomposed_stages_match_single_engine(ranks):
    """the slab-decomposed stage protocol on the fp32 build (emulated ranks, byte-exact exchange of fp32 spectra)"""
    n = 64
    shape = (n, n, n)
    box = torch.as_tensor(synth.cubic_cell(n))
    den = synth.smooth_density(shape, seed=3)
    chi, vext = dev32(np.sqrt(den)), dev32(synth.random_potential(shape, seed=4))
    nel = float(round(den.mean() * abs(np.linalg.det(box.numpy()))))
    names = ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']
    one = Engine(shape, DEV, dtype=F32).set_cell(box).set_terms(names)
    E1, mu1, g1 = one.energy_grad_chi(chi, nel, vext)
    loc = LocalRanks(shape, DEV, ranks, dtype=F32).set_cell(box).set_terms(names)
    E2, mu2, g2 = loc.closure(chi, nel, vext)
    for k in E1:         # fp32 round-off differs with the summation / transform order: per term, not on the cancelling total
        assert abs(E1[k] - E2[k]) <= 2e-6 * max(abs(E1[k]), 1e-3), (k, E1[k], E2[k])
    assert abs(mu1 - mu2) < 1e-6
    err = relerr(g2.cpu().numpy(), g1.cpu().numpy())
    assert err < V_RTOL, err
    one.close()
    loc.close()


@pytest.mark.parametrize('shape', [(48, 96, 120), (144, 160, 192), (240, 250, 270), (320, 384, 96), (250, 270, 240)])
def test_f32_mixed_radix_extents_run_the_fused_pipelines(shape):
    """round-2 verdict, missing 4: extents with factors 3 and 5 in the fp32 build.  The mixed-radix plans (fft_radix.h) now
    compile for fp32 too: such grids take the z-fused pipeline instead of chirp-z + unfused.  Against the fp32 chirp-z path
    (option 9 off: independent transforms and pipeline), the x-fused-only and unfused pipelines, and the fp64 engine."""
    box = torch.as_tensor(cases.make_cell(('tri', shape[0] / 24.0)))
    den = synth.random_density(shape, seed=71)
    vext = synth.random_potential(shape, seed=72)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(73).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box.numpy()))) + 0.3)
    e32 = Engine(shape, DEV, dtype=F32).set_cell(box)
    e64 = Engine(shape, DEV).set_cell(box)
    assert e32.fast_path and e64.fast_path
    sets = [(F.NativeTerms(names).names, None) for names in _CFG_TERMS.values()]
    sets.append((('hartree', 'vw', 'gga_k', 'pbe_x', 'pbe_c'), {'ggak_kind': 1.0, 'ggak_beta': 0.25, 'ggak_lambda': 0.4, 'ggak_sigma': 0.2}))
    for names, params in sets:
        e64.set_terms(names, params)
        E64, mu64, g64 = e64.energy_grad_chi(torch.as_tensor(chi, device=DEV), n_elec, torch.as_tensor(vext, device=DEV))
        e32.set_terms(names, params)
        launches = {}
        for key, (mixed, mode) in {'chirp': (0, 0), 'zf': (1, 0), 'xf': (1, 2), 'un': (1, 1)}.items():
            e32.set_option(9, mixed).set_option(0, mode)
            assert e32.fast_path == bool(mixed)
            E, mu, g = e32.energy_grad_chi(dev32(chi), n_elec, dev32(vext))
            launches[key] = int(e32.query(4))
            for k in E64:
                assert abs(E[k] - E64[k]) <= E_RTOL * max(abs(E64[k]), 1e-3), (names, key, k, E[k], E64[k])
            assert abs(mu - mu64) < 5e-6 * max(1.0, abs(mu64)), (names, key)
            assert relerr(g.cpu().numpy(), g64.cpu().numpy()) < V_RTOL, (names, key)
        assert launches['zf'] < launches['un']             # the fused pipeline really ran
        e32.set_option(9, 1).set_option(0, 0)
    e32.close()
    e64.close()


@pytest.mark.parametrize('shape', [(64, 32, 128), (32, 32, 1024), (8, 64, 32)])
def test_f32_pipelines_agree(shape):
    """fp32 build: z-fused (default), x-fused-only and unfused pipelines on one input (rows of 1024 take the 8-point
    lanes of the z kernels), against each other and against the fp64 engine"""
    box = torch.as_tensor(cases.make_cell(('tri', 1.7)))
    den = synth.random_density(shape, seed=31)
    vext = synth.random_potential(shape, seed=32)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(33).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box.numpy()))) + 0.3)
    e32 = Engine(shape, DEV, dtype=F32).set_cell(box)
    e64 = Engine(shape, DEV).set_cell(box)
    for cfg, names in _CFG_TERMS.items():
        terms = F.NativeTerms(names).names
        e64.set_terms(terms)
        E64, mu64, g64 = e64.energy_grad_chi(torch.as_tensor(chi, device=DEV), n_elec, torch.as_tensor(vext, device=DEV))
        e32.set_terms(terms)
        for mode in (0, 1, 2):
            e32.set_option(0, mode)
            E, mu, g = e32.energy_grad_chi(dev32(chi), n_elec, dev32(vext))
            for k in E64:
                assert abs(E[k] - E64[k]) <= E_RTOL * max(abs(E64[k]), 1e-3), (cfg, mode, k, E[k], E64[k])
            assert abs(mu - mu64) < 5e-6 * max(1.0, abs(mu64))
            assert relerr(g.cpu().numpy(), g64.cpu().numpy()) < V_RTOL, (cfg, mode)
        e32.set_option(0, 0)
    e32.close()
    e64.close()


_TERM_SETS = [
    (['lda_x', 'pw_c'], {}), (['lda_x', 'chachiyo_c'], {}), (['pz_c'], {}),
    (['tf', 'vw', 'wt_nl'], {'wt_alpha': 5 / 6 + 0.2, 'wt_beta': 5 / 6 - 0.2}),          # WGC98-style exponents (alpha != beta)
    (['tf', 'vw', 'wt_nl'], {'wt_alpha': 1.0, 'wt_beta': 1.0}),                          # Perrot
    (['vw', 'gga_k'], {'ggak_kind': 0.0}),                                               # Luo-Karasiev-Trickey
    (['vw', 'gga_k'], {'ggak_kind': 1.0, 'ggak_mu': 40 / 27}),                           # Pauli-Gaussian PGS
    (['vw', 'gga_k'], {'ggak_kind': 1.0, 'ggak_mu': 40 / 27, 'ggak_beta': 0.25}),        # PGSL0.25 (Laplacian member, unfused)
    (['vw', 'vwgtf'], {'vwgtf_kind': 1.0}), (['vw', 'vwgtf'], {'vwgtf_kind': 2.0}),
    (['hartree', 'pbe_x'], {}), (['pbe_c'], {}),
]


@pytest.mark.parametrize('idx', range(len(_TERM_SETS)))
def test_f32_every_term_tracks_f64_on_a_rough_density(idx):
    """every term family of the engine, fp32 against fp64 on a random (full-spectrum) density: catches formulas that are
    fine in fp64 but cancel catastrophically in fp32 (as the Lindhard function did)"""
    names, params = _TERM_SETS[idx]
    shape = (32, 16, 64)
    box = torch.as_tensor(cases.make_cell(('tri', 1.7)))
    den = synth.random_density(shape, seed=51)
    e64 = Engine(shape, DEV).set_cell(box).set_terms(names, params)
    e32 = Engine(shape, DEV, dtype=F32).set_cell(box).set_terms(names, params)
    E64, v64 = e64.energy_potential(torch.as_tensor(den, device=DEV))
    E32, v32 = e32.energy_potential(dev32(den))
    for k in E64:
        assert abs(E32[k] - E64[k]) <= E_RTOL * max(abs(E64[k]), 1e-3), (names, params, k, E32[k], E64[k])
    assert relerr(v32.cpu().numpy(), v64.cpu().numpy()) < V_RTOL, (names, params)
    e64.close()
    e32.close()


def test_f32_engine_hands_ion_routines_to_the_fp64_sibling():
    """ionic potential / forces / stress / ion-ion asked of an fp32 engine: computed by the fp64 routines (potential
    narrowed to fp32, densities widened), identical to asking the fp64 engine directly"""
    from professad_amd.ions import ion_electron_forces, ion_electron_stress, ion_ion, ionic_potential
    shape = (32, 32, 32)
    box = cases.make_cell(('tri', 1.3))
    den = synth.random_density(shape, seed=7)
    ks = np.linspace(0.0, 12.0, 300)
    tab = (ks, -4 * np.pi * 3.0 / (ks ** 2 + 1.5) * np.exp(-0.05 * ks ** 2), 3)
    frac = np.array([[0.1, 0.2, 0.3], [0.6, 0.55, 0.8]])
    e32 = Engine(shape, DEV, dtype=F32)
    e64 = Engine(shape, DEV)
    v32 = ionic_potential(e32, box, [(frac, tab)], pme_order=4)
    v64 = ionic_potential(e64, box, [(frac, tab)], pme_order=4)
    # (the PME spreading adds with atomics: two runs may differ in the last bits, so no bitwise comparison here)
    assert v32.dtype == F32 and relerr(v32.cpu().numpy(), v64.cpu().numpy()) < 2e-7
    d32 = dev32(den)
    d64 = d32.double()
    assert np.allclose(ion_electron_forces(e32, box, d32, [(frac, tab)], pme_order=4)[0],
                       ion_electron_forces(e64, box, d64, [(frac, tab)], pme_order=4)[0], rtol=1e-11, atol=1e-14)
    assert np.allclose(ion_electron_stress(e32, box, d32, [(frac, tab)]), ion_electron_stress(e64, box, d64, [(frac, tab)]),
                       rtol=1e-11, atol=1e-16)
    a, b = ion_ion(e32, box, frac, [3.0, 3.0]), ion_ion(e64, box, frac, [3.0, 3.0])
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    e32.close()
    e64.close()


@pytest.mark.parametrize('n', [16, 32, 64])
@pytest.mark.parametrize('terms', [['ion_electron', 'hartree', 'wt', 'pbe'], ['ion_electron', 'hartree', 'wgc99', 'pbe'],
                                   ['ion_electron', 'hartree', 'tf', 'lkt', 'pz'], ['ion_electron', 'hartree', 'wt', 'pz']],
                         ids=['wt_pbe', 'wgc_pbe', 'lkt', 'wt_lda'])
def test_f32_persistent_kernel_matches_the_f32_staged_pipeline(n, terms):
    """the persistent small-grid kernel in the fp32 build, every phase of it (gradient / flux / divergence, WGC99 triples): closure
    and density-input entry against the staged fp32 pipeline (fp32 round-off) and the fp64 engine (the fp32 tolerances)"""
    from professad_amd import _native as N
    shape = (n, n, n)
    rng = np.random.default_rng(7)
    box = torch.as_tensor(synth.triclinic_cell(n / 4.0))
    den = synth.smooth_density(shape, seed=3, amp=0.5) * (1 + 0.1 * rng.random(shape))
    vext = synth.random_potential(shape, seed=4)
    chi = np.sqrt(den)
    names = F.NativeTerms(terms).names
    nel = 9.0
    e64 = Engine(shape, DEV).set_cell(box).set_terms(names).set_option(N.OPT_RESIDENT, 0)
    st = Engine(shape, DEV, dtype=F32).set_cell(box).set_terms(names).set_option(N.OPT_RESIDENT, 0)
    rs = Engine(shape, DEV, dtype=F32).set_cell(box).set_terms(names)
    Er, mur, gr = e64.energy_grad_chi(torch.as_tensor(chi, device=DEV), nel, torch.as_tensor(vext, device=DEV))
    Ea, mua, ga = st.energy_grad_chi(dev32(chi), nel, dev32(vext))
    Eb, mub, gb = rs.energy_grad_chi(dev32(chi), nel, dev32(vext))
    served = int(rs.query(N.Q_RESIDENT_EVALS))
    assert served == (0 if (n == 64 and 'wgc99' in terms) else 1)             # (fp32 serves gradient terms at 64^3 too)
    for k in Er:
        assert abs(Eb[k] - Er[k]) <= E_RTOL * max(abs(Er[k]), 1e-3), (k, Eb[k], Er[k])
    assert relerr(gb.cpu().numpy(), gr.cpu().numpy()) < V_RTOL
    assert relerr(gb.cpu().numpy(), ga.cpu().numpy()) < 1e-4
    dn = dev32(den * (nel / (den.mean() * abs(np.linalg.det(box.numpy())))))
    Ea, va = st.energy_potential(dn, dev32(vext))
    Eb, vb = rs.energy_potential(dn, dev32(vext))
    assert relerr(vb.cpu().numpy(), va.cpu().numpy()) < 1e-4
    for k in Ea:
        assert abs(Ea[k] - Eb[k]) <= E_RTOL * max(abs(Ea[k]), 1e-3)
    for e in (e64, st, rs):
        e.close()
