"""GPU parity tests: the HIP engine (through the C ABI) against
  (1) the reference's own outputs committed as golden fixtures,
  (2) the pinned CPU oracle on seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at larger sizes.
Tolerances: fp64 everywhere; energies within 1e-10 relative (north-star bar: 1e-8 Ha/atom),
potentials within 5e-10 of the max magnitude."""
import os

import numpy as np
import pytest
import torch

import cases
from oracle import closed_form as cf
from professad_amd import functionals as F
from professad_amd import synth
from professad_amd.engine import Engine, engine_for

pytestmark = pytest.mark.gpu
GOLDEN = os.path.dirname(os.path.abspath(cases.__file__))
DEV = 'cuda:0'
E_RTOL = 1e-10
V_RTOL = 5e-10


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.double, device=DEV)


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300))


def load(name):
    return np.load(os.path.join(GOLDEN, name))


# ------------------------------------------------------------------------------- FFT building block
FFT_SHAPES = [(8, 8, 16), (16, 16, 16), (16, 32, 64), (64, 64, 64), (32, 16, 128), (8, 128, 32), (256, 8, 32),
              (128, 128, 128), (512, 8, 16), (8, 1024, 16), (8, 8, 1024), (8, 8, 2048),
              (17, 17, 17), (18, 20, 16), (5, 6, 7), (20, 20, 20), (9, 8, 12),
              # extents with factors 3 and 5 served by mixed-radix plans (fft_radix.h), every x / y / z plan at least once
              (48, 96, 120), (120, 48, 96), (144, 160, 48), (96, 144, 160), (192, 240, 250), (250, 192, 240), (240, 250, 192),
              (270, 288, 144), (288, 270, 320), (320, 48, 270), (384, 48, 288), (48, 384, 384), (480, 96, 320), (96, 480, 480),
              (64, 240, 48), (250, 16, 270)]


@pytest.mark.parametrize('shape', FFT_SHAPES)
def test_rfftn_irfftn_match_numpy(shape):
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape)
    eng = Engine(shape, DEV)
    got = eng.rfftn(dev(x)).cpu().numpy()
    ref = np.fft.rfftn(x)
    assert relerr(got, ref) < 1e-13
    # non-Hermitian spectrum: imaginary parts at kz=0 / Nyquist must be ignored exactly like irfftn
    yk = rng.standard_normal(ref.shape) + 1j * rng.standard_normal(ref.shape)
    got_r = eng.irfftn(torch.as_tensor(yk, device=DEV)).cpu().numpy()
    ref_r = np.fft.irfftn(yk, s=shape, axes=(0, 1, 2))
    assert relerr(got_r, ref_r) < 1e-13
    mixed = (48, 96, 120, 144, 160, 192, 240, 250, 270, 288, 320, 384, 480)          # extents with a mixed-radix plan
    assert eng.fast_path == all((s & (s - 1)) == 0 or s in mixed for s in shape)
    eng.close()


# ------------------------------------------------------------------------------- golden: per term
_REF_NAME = {
    'hartree': F.Hartree, 'tf': F.ThomasFermi, 'vw': F.Weizsaecker,
    'wt_nl': lambda b, d: F.non_local_KEF(b, d, 5 / 6, 5 / 6),
    'wt': F.WangTeter, 'perrot': F.Perrot, 'sm': F.SmargiassiMadden, 'wgc98': F.WangGovindCarter98,
    'wgc99': F.WangGovindCarter99(), 'lda_x': F.lda_exchange, 'pz_c': F.perdew_zunger_correlation,
    'pw_c': F.perdew_wang_correlation, 'chachiyo_c': F.chachiyo_correlation,
    'pbe_x': F.pbe_exchange, 'pbe_c': F.pbe_correlation,
    'lkt': F.LuoKarasievTrickey, 'pg1': F.PauliGaussian((1.0, 0.0, 0.0, 0.0)), 'pgs': F.PauliGaussian((40 / 27, 0.0, 0.0, 0.0)),
    'wts_exp': F.WangTeterStyleFunctional((5 / 6, 5 / 6, torch.exp)),
    'pgsl025': F.PauliGaussian(), 'pgslr': F.PauliGaussian((40 / 27, 0.25, 0.4, 0.2)),
    'vwgtf1': F.vWGTF1, 'vwgtf2': F.vWGTF2,
}


@pytest.mark.parametrize('case', cases.PER_TERM_CASES)
def test_terms_match_reference_golden(case):
    gold = load('terms_%s.npz' % case)
    box, den, vext, chi, n_elec = cases.make_inputs(case)
    tb, td, tv = dev(box), dev(den), dev(vext)
    for nm in cases.SINGLE_TERMS:
        f = (lambda b, d: F.IonElectron(b, d, tv)) if nm == 'ion_electron' else _REF_NAME[nm]
        E = float(f(tb, td))
        v = F.get_functional_derivative(tb, td, f).cpu().numpy()
        Eref, vref = float(gold['E_' + nm]), gold['v_' + nm]
        assert abs(E - Eref) <= E_RTOL * max(1.0, abs(Eref)), (nm, E, Eref)
        assert relerr(v, vref) < V_RTOL, (nm, relerr(v, vref))


# ------------------------------------------------------------------------------- golden: fused + closure
_CFG_TERMS = {
    'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'pz'],
    'cfg2': ['ion_electron', 'hartree', 'wt', 'pz'],
    'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe'],
}


@pytest.mark.parametrize('case', cases.FUSED_CASES)
def test_fused_configs_and_closure_match_reference_golden(case):
    gold = load('fused_%s.npz' % case)
    box, den, vext, chi, n_elec = cases.make_inputs(case)
    tb, td, tv, tc = dev(box), dev(den), dev(vext), dev(chi)
    for cfg, names in _CFG_TERMS.items():
        fused = F.NativeTerms(names)
        d = td.clone().requires_grad_()
        E = fused(tb, d, tv)
        E.backward()
        dV = abs(np.linalg.det(box)) / den.size
        Eref = float(gold['E_' + cfg])
        assert abs(float(E) - Eref) <= E_RTOL * max(1.0, abs(Eref))
        assert relerr(d.grad.cpu().numpy() / dV, gold['v_' + cfg]) < V_RTOL
        # `potentials=` hook form
        assert relerr(fused.potential(tb, td, tv).cpu().numpy(), gold['v_' + cfg]) < V_RTOL
        # the optimize_density closure
        eng = engine_for(den.shape, DEV).set_cell(tb).set_terms(fused.names)
        Et, mu, g = eng.energy_grad_chi(tc, n_elec, tv)
        Ecl = float(gold['Ec_' + cfg])
        assert abs(sum(Et.values()) - Ecl) <= E_RTOL * max(1.0, abs(Ecl))
        assert relerr(g.cpu().numpy(), gold['g_' + cfg]) < V_RTOL


@pytest.mark.parametrize('fname', ['big_scalars.json', 'huge_scalars.json'])
def test_big_scalars_match_reference_golden(fname):
    """64^3 / 128^3 (big) and the full 256^3 bench size (huge): energy and statistics / probes of dE/dn of the fused
    configurations against the reference itself run at that size"""
    import json
    path = os.path.join(GOLDEN, fname)
    if not os.path.exists(path):
        pytest.skip(fname + ' not generated')
    big = json.load(open(path))
    for key, rec in big.items():
        cfg, n = key.split('_')
        n = int(n)
        shape = (n, n, n)
        box = synth.cubic_cell(n)
        den = synth.random_density(shape, seed=1234)
        vext = synth.random_potential(shape, seed=77)
        assert abs(cases.checksum(den[:8, :8, :8]) - rec['input_checksum']) < 1e-9
        fused = F.NativeTerms(_CFG_TERMS[cfg])
        v = fused.potential(dev(box), dev(den), dev(vext)).cpu().numpy()
        E = sum(fused.last_energies.values())
        assert abs(E - rec['E']) <= E_RTOL * abs(rec['E']), (key, E, rec['E'])
        st = cases.probe_stats(v)
        assert abs(st['sum'] - rec['pot']['sum']) <= 1e-9 * abs(rec['pot']['l2']) * np.sqrt(v.size)
        assert abs(st['l2'] - rec['pot']['l2']) <= 1e-10 * rec['pot']['l2']
        assert np.allclose(st['probes'], rec['pot']['probes'], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize('n', [64, 128, 256])
def test_bench_workload_energy_mu_and_gradient_match_the_reference_pin(n):
    """the TIMED workload of bench.py (its own input recipe) through the closure call: E, mu and chi.grad (L2 norm, sum, eight
    probes) against the reference's closure on these inputs (tests/golden/bench_scalars.json; system.py:830-853); the same
    check bench.py runs on its timed call (reference_check: exit code 3 on a mismatch)"""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    with open(os.path.join(GOLDEN, 'bench_scalars.json')) as fh:
        ref = json.load(fh)['cfg3_%d' % n]
    box, chi, vext, n_elec, _ = bench.make_inputs(n)
    assert abs(cases.checksum(chi[:8, :8, :8], vext[:8, :8, :8]) - ref['input_checksum']) < 1e-12 and n_elec == ref['n_elec']
    eng = Engine((n, n, n), DEV).set_cell(dev(box)).set_terms(bench.CFG3)
    Et, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
    st = bench.grad_stats(g)
    chk = bench.reference_check(n, 'cfg3', 'f64', sum(Et.values()), mu, st)
    assert chk['ok'] and chk['rel_dE'] < 1e-10 and chk['grad_rel_dl2'] < 1e-9 and chk['grad_probe_max_rel'] < 1e-9, chk
    # a gradient that is off by 1e-6 of its scale does not pass
    bad = dict(st, probes=[p * (1 + 1e-6) for p in st['probes']])
    assert not bench.reference_check(n, 'cfg3', 'f64', sum(Et.values()), mu, bad)['ok']
    eng.close()


@pytest.mark.parametrize('shape,cell', [((256, 16, 32), ('cubic', 32)), ((256, 32, 16), ('tri', 1.1)), ((512, 8, 16), ('cubic', 16))])
def test_folded_wgc99_table_reads_change_nothing(shape, cell):
    """round 5: on cells with orthogonal axes the cross-wave x pass reads the WGC99 table entry of x > n0 / 2 at n0 - x (|k| is
    even along a line; functionals.py:968-972 depends on |k| only).  Same numbers as with every k-point reading its own entry
    (OFDFT_OPT_WGC_FOLD = 0) -- to the rounding of |k|^2 of the two representatives -- and as the oracle; a triclinic cell never folds"""
    from professad_amd import _native as N
    box = cases.make_cell(cell) if cell[0] == 'tri' else np.diag([7.6 * s / 32.0 * (1.0 + 0.1 * i) for i, s in enumerate(shape)])
    den = synth.smooth_density(shape, seed=21) * (1 + 0.05 * np.random.default_rng(5).random(shape))
    vext = synth.random_potential(shape, seed=22)
    chi = np.sqrt(den)
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box))) + 0.5)
    names = F.NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe']).names
    out = {}
    for fold in (1, 0):
        eng = Engine(shape, DEV).set_cell(dev(box)).set_terms(names).set_option(N.OPT_WGC_FOLD, fold)
        out[fold] = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
        eng.close()
    (Ea, mua, ga), (Eb, mub, gb) = out[1], out[0]
    tol = 0.0 if cell[0] == 'tri' else 1e-13
    assert abs(sum(Ea.values()) - sum(Eb.values())) <= tol * abs(sum(Eb.values())) and abs(mua - mub) <= tol * abs(mub)
    assert float((ga - gb).abs().max()) <= tol * float(gb.abs().max())
    if shape[0] == 256:
        ev = cf.Evaluator(cf.Grid(box, shape))
        Ec, go, muo = ev.closure(cases.CONFIGS['cfg3'], chi, n_elec, vext)
        assert abs(sum(Ea.values()) - Ec) <= E_RTOL * max(1.0, abs(Ec)) and relerr(ga.cpu().numpy(), go) < V_RTOL


@pytest.mark.parametrize('dt', [torch.double, torch.float32])
def test_views_that_are_not_16_byte_aligned_give_the_same_numbers(dt):
    """chi handed over as a contiguous VIEW into a larger tensor (offsets of 1..3 elements: 4- / 8- / 12-byte alignment): the
    streaming kernels' 16-byte accesses (round 5) fall back to pairs, everything else takes the pointer as it is"""
    shape = (64, 64, 64)
    box = dev(synth.cubic_cell(64))
    chi_h = np.sqrt(synth.smooth_density(shape, seed=3))
    vext = torch.as_tensor(synth.random_potential(shape, seed=4), dtype=dt, device=DEV)
    n = int(np.prod(shape))
    big = torch.zeros(n + 8, dtype=dt, device=DEV)
    ref = None
    for off in (0, 1, 2, 3):
        chi = big[off:off + n].view(shape)
        chi.copy_(torch.as_tensor(chi_h, dtype=dt))
        from professad_amd import _native as N
        eng = Engine(shape, DEV, dtype=dt).set_cell(box).set_terms(['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c'])
        eng.set_option(N.OPT_RESIDENT, 0)
        E, mu, g = eng.energy_grad_chi(chi, 96.0, vext)
        eng.close()
        cur = (sum(E.values()), mu, g.double())
        if ref is None:
            ref = cur
        tol = 1e-13 if dt == torch.double else 1e-6
        assert abs(cur[0] - ref[0]) <= tol * abs(ref[0]) and abs(cur[1] - ref[1]) <= tol * abs(ref[1]), (off, cur[0], ref[0])
        assert float((cur[2] - ref[2]).abs().max()) <= tol * float(ref[2].abs().max()), off


# ------------------------------------------------------------------------------- oracle on seeded inputs
@pytest.mark.parametrize('shape,cell', [((64, 64, 64), ('cubic', 64)), ((32, 64, 16), ('tri', 1.3)),
                                        ((24, 20, 18), ('tri', 0.8)), ((33, 32, 31), ('cubic', 32))])
def test_configs_match_oracle_on_seeded_inputs(shape, cell):
    box = cases.make_cell(cell)
    den = synth.smooth_density(shape, seed=11) * (1 + 0.05 * np.random.default_rng(5).random(shape))
    vext = synth.random_potential(shape, seed=12)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(6).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box))) + 0.3)
    ev = cf.Evaluator(cf.Grid(box, shape))
    for cfg, names in _CFG_TERMS.items():
        fused = F.NativeTerms(names)
        v = fused.potential(dev(box), dev(den), dev(vext)).cpu().numpy()
        Eo, Es, vo = ev.terms(cases.CONFIGS[cfg], den, vext)
        assert abs(sum(fused.last_energies.values()) - Eo) <= E_RTOL * max(1.0, abs(Eo))
        assert relerr(v, vo) < V_RTOL
        eng = engine_for(shape, DEV).set_cell(dev(box)).set_terms(fused.names)
        Et, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
        Ec, go, muo = ev.closure(cases.CONFIGS[cfg], chi, n_elec, vext)
        assert abs(sum(Et.values()) - Ec) <= E_RTOL * max(1.0, abs(Ec))
        assert abs(mu - muo) <= 1e-9 * max(1.0, abs(muo))
        assert relerr(g.cpu().numpy(), go) < V_RTOL


# ------------------------------------------------------------------------------- size-independent properties
@pytest.mark.parametrize('n', [128, 256, 512])
def test_periodic_tiling_gives_extensive_energy_and_tiled_potential(n):
    """A 32^3 state tiled (n/32)^3 times on the (n/32)x cell: every term is extensive, so E scales by the
    tile count and the potential is the tiled 32^3 potential (needs no reference on the GPU box)."""
    base = 32
    box32 = synth.cubic_cell(base)
    den32 = synth.smooth_density((base,) * 3, seed=21) * (1 + 0.02 * np.random.default_rng(8).random((base,) * 3))
    vext32 = synth.random_potential((base,) * 3, seed=22)
    # make the 32^3 electron count an exact integer so that WGC99's rounding commutes with tiling
    vol32 = abs(np.linalg.det(box32))
    den32 *= 3.0 / (den32.mean() * vol32)
    r = n // base
    fused = F.NativeTerms(_CFG_TERMS['cfg3'])
    v32 = fused.potential(dev(box32), dev(den32), dev(vext32)).cpu().numpy()
    E32 = dict(fused.last_energies)
    vN = fused.potential(dev(synth.cubic_cell(n)), dev(synth.tile_periodic(den32, n)),
                         dev(synth.tile_periodic(vext32, n)))
    EN = dict(fused.last_energies)
    for k in E32:
        assert abs(EN[k] - r ** 3 * E32[k]) <= 2e-10 * max(1.0, abs(EN[k])), (k, EN[k], r ** 3 * E32[k])
    assert relerr(vN[:base, :base, :base].cpu().numpy(), v32) < V_RTOL
    assert relerr(vN[-base:, base:2 * base, -base:].cpu().numpy(), v32) < V_RTOL
    # the stress is intensive: the tiled cell has the stress tensors of the 32^3 cell (every term, incl. WGC99)
    bits = F.NativeTerms(['hartree', 'wgc99', 'pbe']).names
    s32 = Engine((base,) * 3, DEV).set_cell(dev(box32)).set_terms(bits)
    sN = engine_for((n,) * 3, DEV).set_cell(dev(synth.cubic_cell(n))).set_terms(bits)
    sig32, sigN = s32.stress(dev(den32)), sN.stress(dev(synth.tile_periodic(den32, n)))
    for k in sig32:
        assert np.abs(sigN[k] - sig32[k]).max() <= 1e-9 * max(1e-6, np.abs(sig32[k]).max()), k
    s32.close()


def test_round_trip_and_linearity_256():
    shape = (256, 256, 256)
    eng = engine_for(shape, DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    a = torch.randn(shape, dtype=torch.double, device=DEV, generator=g)
    b = torch.randn(shape, dtype=torch.double, device=DEV, generator=g)
    ak = eng.rfftn(a)
    assert float((eng.irfftn(ak) - a).abs().max()) < 1e-12
    lin = eng.rfftn(a + 2 * b) - (ak + 2 * eng.rfftn(b))
    assert float(lin.abs().max()) < 1e-9 * float(ak.abs().max())
    # Parseval on the half spectrum
    w = torch.full((129,), 2.0, dtype=torch.double, device=DEV)
    w[0] = w[-1] = 1.0
    lhs = float((a * a).sum())
    rhs = float(((ak.real ** 2 + ak.imag ** 2) * w).sum()) / a.numel()
    assert abs(lhs - rhs) < 1e-11 * lhs


# ------------------------------------------------------------------------------- edge cases / errors
def test_zero_density_point_vw_guard_and_errors():
    box, den, vext, chi, n_elec = cases.make_inputs('g16r')
    tb = dev(box)
    d0 = den.copy()
    d0[3, 4, 5] = 0.0
    v = F.get_functional_derivative(tb, dev(d0), F.Weizsaecker).cpu().numpy()
    assert np.isfinite(v).all() and v[3, 4, 5] == 0.0              # functionals.py:242-243 guard
    with pytest.raises(RuntimeError):
        engine_for(den.shape, DEV).set_cell(tb).set_terms(['ion_electron']).energy_potential(dev(den), None)
    with pytest.raises(TypeError):
        F.Hartree(tb, dev(den).float())
    with pytest.raises(ValueError):
        engine_for(den.shape, DEV).energy_potential(dev(den)[:8])
    tbg = tb.clone().requires_grad_()                 # lattice-vector gradients are served by the analytic stress
    assert torch.autograd.grad(F.Hartree(tbg, dev(den)), tbg)[0].shape == (3, 3)
    with pytest.raises(RuntimeError):
        engine_for(den.shape, DEV).set_cell(torch.zeros(3, 3, dtype=torch.double))


def test_reference_system_protocol_sum_of_terms():
    """Mimics System.__compute_energy's dispatch (system.py:759-772) over native terms."""
    box, den, vext, chi, n_elec = cases.make_inputs('g16r')
    gold = load('fused_g16r.npz')
    tb, tv = dev(box), dev(vext)
    terms = [F.IonIon, F.IonElectron, F.Hartree, F.WangGovindCarter99().forward, F.PerdewBurkeErnzerhof]
    d = dev(den).requires_grad_()
    E = torch.zeros((1,), dtype=torch.double, device=DEV)
    for f in terms:
        if f.__qualname__ == 'IonElectron':
            E = E + f(tb, d, tv)
        elif f.__qualname__ == 'IonIon':
            continue
        else:
            E = E + f(tb, d)
    E.backward()
    assert abs(float(E) - float(gold['E_cfg3'])) <= E_RTOL * abs(float(gold['E_cfg3']))
    dV = abs(np.linalg.det(box)) / den.size
    assert relerr(d.grad.cpu().numpy() / dV, gold['v_cfg3']) < V_RTOL


@pytest.mark.parametrize('shape', [(64, 32, 128), (16, 8, 16), (8, 64, 32), (32, 32, 1024)])
def test_all_pipelines_agree(shape):
    """z-fused (default), x-fused-only and unfused pipelines of the engine on one input, energy_potential and
    closure forms."""
    box = cases.make_cell(('tri', 1.7))
    den = synth.random_density(shape, seed=31)
    vext = synth.random_potential(shape, seed=32)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(33).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box))) + 0.3)
    eng = Engine(shape, DEV).set_cell(dev(box))
    for cfg, names in _CFG_TERMS.items():
        eng.set_terms(F.NativeTerms(names).names)
        res = {}
        for mode in (0, 1, 2):
            eng.set_option(0, mode)
            E, v = eng.energy_potential(dev(den), dev(vext))
            Ec, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
            res[mode] = (E, v.cpu().numpy(), Ec, mu, g.cpu().numpy(), eng.query(0))
        for mode in (0, 2):
            for k in res[1][0]:
                assert abs(res[mode][0][k] - res[1][0][k]) <= 1e-12 * max(1.0, abs(res[1][0][k])), (cfg, mode, k)
                assert abs(res[mode][2][k] - res[1][2][k]) <= 1e-12 * max(1.0, abs(res[1][2][k])), (cfg, mode, k)
            assert relerr(res[mode][1], res[1][1]) < 1e-12, (cfg, mode)
            assert relerr(res[mode][4], res[1][4]) < 1e-12, (cfg, mode)
            assert abs(res[mode][3] - res[1][3]) < 1e-12 * max(1.0, abs(res[1][3]))
            assert res[mode][5] == res[1][5]          # same number of 3-D FFTs
        # x-chunked form of the z-fused pipeline (Infinity-Cache reuse): same kernels on x ranges, same reduction order
        eng.set_option(0, 0)
        # the split combine (WGC99 part as its own kernel on the side stream) adds the same numbers in another order
        eng.set_option(4, 0)
        E0, v0 = eng.energy_potential(dev(den), dev(vext))
        Ec0, mu0, g0 = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
        for k in E0:
            assert abs(E0[k] - res[0][0][k]) <= 1e-13 * max(1.0, abs(E0[k])) and abs(Ec0[k] - res[0][2][k]) <= 1e-13 * max(1.0, abs(Ec0[k]))
        assert relerr(v0.cpu().numpy(), res[0][1]) < 1e-13 and relerr(g0.cpu().numpy(), res[0][4]) < 1e-13
        res[0] = (E0, v0.cpu().numpy(), Ec0, mu0, g0.cpu().numpy(), eng.query(0))
        eng.set_option(3, 31)                          # every stage pair chunked (x ranges and kz-block ranges)
        for nch in (1, 2, 8):
            eng.set_option(2, nch)
            E, v = eng.energy_potential(dev(den), dev(vext))
            Ec, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
            assert all(E[k] == res[0][0][k] and Ec[k] == res[0][2][k] for k in E), (cfg, nch)
            assert np.array_equal(v.cpu().numpy(), res[0][1]) and np.array_equal(g.cpu().numpy(), res[0][4]), (cfg, nch)
            assert mu == res[0][3] and eng.query(0) == res[0][5]
        eng.set_option(2, 0)
        eng.set_option(3, 2)
        eng.set_option(4, 1)
    eng.close()


@pytest.mark.parametrize('optimizer', ['fused', 'torch'])
def test_density_optimisation_reaches_reference_ground_state(optimizer):
    """Config 1 end to end (reference tests/test_den_opt.py path): from the uniform density the native closure +
    the from-scratch fixed-step L-BFGS reach the state the reference's System.optimize_density converged to
    (fixture made by running the reference: E = 2.40469334875 Ha in 17 outer iterations).  The fixed-step L-BFGS
    trajectory amplifies 1e-15 differences ~50x per inner iteration early on (measured), so iteration counts may
    differ by one or two; the minimum may not."""
    from professad_amd.optimize import optimize_density
    d = np.load(os.path.join(GOLDEN, 'cfg1_fccAl_32.npz'))
    box, vext, den_ref, n_elec = d['box'], d['vext'], d['den'], float(d['n_elec'])
    eng = Engine((32, 32, 32), DEV).set_cell(dev(box)).set_terms(F.NativeTerms(_CFG_TERMS['cfg1']).names)
    res = optimize_density(eng, n_elec, dev(vext), volume=abs(np.linalg.det(box)), optimizer=optimizer)
    assert res['converged'] and abs(res['iterations'] - 17) <= 3
    assert abs(res['E_Ha'] - float(d['E_Ha'])) < 2e-8
    assert relerr(res['den'].cpu().numpy(), den_ref) < 1e-4
    # first outer iteration (before the chaotic amplification sets in) matches the reference's log to all digits shown
    assert abs(res['history'][0][1] - 68.191536) < 1e-6 and abs(res['history'][0][3] - 0.593563) < 1e-6
    # Later rows cannot be pinned the same way: two closures that agree to 1e-13 (the staged pipeline and the persistent kernel)
    # give 65.989312 and 65.986863 eV in row 2 where the reference's log has 65.989145 (tools/opt_rows_probe.py) -- the fixed-
    # step trajectory amplifies round-off by ~50x per inner iteration; both optimisers walk identical rows on the same closure.
    # The optimiser's own semantics are pinned row by row on the CPU, with the oracle closure (tests/test_optimizer_cpu.py).
    assert abs(res['history'][1][1] - 65.989145) < 5e-3
    # ... with the STAGED closure (the bench's pipeline; persistent kernel off) row 2 lands within 1.7e-4 eV of the reference's
    # log: pinned at 5e-4 (round-2 verdict, item 9)
    from professad_amd import _native as N
    eng.set_option(N.OPT_RESIDENT, 0)
    res2 = optimize_density(eng, n_elec, dev(vext), volume=abs(np.linalg.det(box)), optimizer=optimizer)
    assert res2['converged'] and abs(res2['iterations'] - 17) <= 3 and abs(res2['E_Ha'] - float(d['E_Ha'])) < 2e-8
    assert abs(res2['history'][0][1] - 68.191536) < 1e-6
    assert abs(res2['history'][1][1] - 65.989145) < 5e-4, res2['history'][1]
    eng.close()


@pytest.mark.parametrize('n', [53, 64])
def test_exact_single_orbital_cases(n):
    """The reference's exact cases (tests/test_den_opt.py:13-40): one electron with IonElectron + Weizsaecker is the single-orbital
    Schroedinger problem -- the hydrogen atom (Coulomb recpot; E -> -0.5 Ha, the reference asserts 2 places) and the harmonic
    oscillator v = k r^2 / 2 (E = 3/2 sqrt(k), 5 places).  53^3 is the grid the reference's System.ecut2shape gives its 20-bohr box
    (prime: chirp-z transforms, unfused pipeline; fixture = the reference's own run, tests/golden/exact_cases.npz), 64^3 the next
    fused extent (persistent small-grid kernel)."""
    from professad_amd.ions import ionic_potential, recpot_table
    from professad_amd.optimize import optimize_density
    g = load('exact_cases.npz')
    L, k = 20.0, float(g['qho_k'])
    box = L * np.eye(3)
    eng = Engine((n, n, n), DEV).set_cell(dev(box)).set_terms(['ion_electron', 'vw'])
    # harmonic oscillator
    f = np.arange(n) / n
    x, y, z = np.meshgrid(L * f, L * f, L * f, indexing='ij')
    pot = 0.5 * k * ((x - L / 2) ** 2 + (y - L / 2) ** 2 + (z - L / 2) ** 2)
    res = optimize_density(eng, 1.0, dev(pot), volume=L ** 3, ntol=1e-4)
    assert res['converged'] and abs(res['E_Ha'] - 1.5 * np.sqrt(k)) < 5e-6
    if n == 53:
        assert abs(res['E_Ha'] - float(g['qho_E_Ha'])) < 2e-6 and abs(res['iterations'] - int(g['qho_iterations'])) <= 6
    # hydrogen atom: the ionic potential from the recpot table (equal to the reference's on its grid), then the ground state
    tab = recpot_table(g['h_raw'], float(g['h_kmax']))
    vext = ionic_potential(eng, box, [(np.array([[0.5, 0.5, 0.5]]), tab)])
    if n == 53:
        assert relerr(vext.cpu().numpy(), g['h_vext']) < 1e-10
    res = optimize_density(eng, 1.0, vext, volume=L ** 3, ntol=1e-4)
    assert res['converged'] and abs(res['E_Ha'] + 0.5) < 5e-3
    if n == 53:
        assert abs(res['E_Ha'] - float(g['h_E_Ha'])) < 1e-5
    eng.close()


def test_ionic_potential_matches_reference_golden():
    """v_ext from ion positions (exact and PME structure factors) against the reference's lattice_sum outputs."""
    from professad_amd.ions import ionic_potential, recpot_table
    g = load('ions.npz')
    tab = recpot_table(g['recpot_raw'], float(g['recpot_kmax']))
    assert tab[2] == 3
    for tag, shape, orders in (('a', (32, 32, 32), (None, 4, 10)), ('b', (16, 20, 24), (None, 6))):
        eng = Engine(shape, DEV)
        for o in orders:
            v = ionic_potential(eng, g[tag + '_box'], [(g[tag + '_frac'], tab)], pme_order=o).cpu().numpy()
            ref = g[tag + '_v_exact'] if o is None else g['%s_v_pme%d' % (tag, o)]
            assert relerr(v, ref) < 1e-11, (tag, o, relerr(v, ref))
        # two "species" (the ion list split in two) accumulate to the same potential
        fr = g[tag + '_frac']
        v2 = ionic_potential(eng, g[tag + '_box'], [(fr[:2], tab), (fr[2:], tab)], pme_order=orders[-1]).cpu().numpy()
        assert relerr(v2, g['%s_v_pme%d' % (tag, orders[-1])]) < 1e-11
        with pytest.raises(RuntimeError):
            ionic_potential(eng, g[tag + '_box'], [(fr, tab)], pme_order=3)
        eng.close()
    # config 1 end to end: ions -> v_ext equals the potential the reference's System used
    c1 = load('cfg1_fccAl_32.npz')
    eng = Engine((32, 32, 32), DEV)
    v = ionic_potential(eng, c1['box'], [(g['a_frac'], tab)]).cpu().numpy()
    assert relerr(v, c1['vext']) < 1e-10
    eng.close()


def test_ion_electron_forces_match_reference_golden():
    """-dU/dR on every ion (exact and PME) against the reference's autograd forces; plus a finite-difference check
    of the forces against the engine's own ionic potential at a size the goldens do not cover."""
    from professad_amd.ions import ion_electron_forces, ionic_potential, recpot_table
    g = load('ions.npz')
    tab = recpot_table(g['recpot_raw'], float(g['recpot_kmax']))
    for tag, shape, order, dk in (('a', (32, 32, 32), 10, dict(seed=8, n0=0.03, amp=0.5)),
                                  ('b', (16, 20, 24), 6, dict(seed=7, n0=0.05, amp=0.5))):
        eng = Engine(shape, DEV)
        den = torch.as_tensor(synth.smooth_density(shape, **dk), device=DEV)
        for o in (None, order):
            F = ion_electron_forces(eng, g[tag + '_box'], den, [(g[tag + '_frac'], tab)], pme_order=o)[0]
            ref = g[tag + '_force_exact'] if o is None else g['%s_force_pme%d' % (tag, o)]
            assert np.abs(F - ref).max() < 1e-12, (tag, o, np.abs(F - ref).max())
        eng.close()
    # 64^3, triclinic: central finite differences of U = dV sum n v_ext[R]
    shape = (64, 64, 64)
    box = synth.triclinic_cell(1.0)
    frac = np.array([[0.03, 0.11, 0.52], [0.48, 0.57, 0.02], [0.71, 0.33, 0.80]])
    den = torch.as_tensor(synth.smooth_density(shape, seed=3, n0=0.03, amp=0.4), device=DEV)
    eng = Engine(shape, DEV)
    dV = abs(np.linalg.det(box)) / np.prod(shape)
    inv = np.linalg.inv(box)
    for o in (None, 8):
        F = ion_electron_forces(eng, box, den, [(frac, tab)], pme_order=o)[0]
        h = 1e-4
        for a, j in ((0, 0), (1, 2), (2, 1)):
            U = []
            for sgn in (+1, -1):
                cart = frac @ box
                cart[a, j] += sgn * h
                v = ionic_potential(eng, box, [(cart @ inv, tab)], pme_order=o)
                U.append(float((v * den).sum()) * dV)
            fd = -(U[0] - U[1]) / (2 * h)
            assert abs(fd - F[a, j]) < 2e-7 * max(1.0, abs(fd)), (o, a, j, fd, F[a, j])
    eng.close()


@pytest.mark.parametrize('n', [4096, 1001, 262147])
def test_lbfgs_sweeps_match_numpy_double(n):
    """ofdft_lbfgs_dots / _commit / _update against the numpy statement of their contract: a scripted sequence with
    pushes, a rejected pair and history wrap-around; even and odd vector lengths"""
    from lbfgs_double import NumpyLbfgsBackend
    from professad_amd.optimize import HipLbfgsBackend
    rng = np.random.default_rng(n)
    hip, ref = HipLbfgsBackend(n, 8, DEV), NumpyLbfgsBackend(n, 8)
    x = rng.standard_normal(n)
    xd = dev(x.copy())
    for it in range(12):
        g = rng.standard_normal(n) * (1.0 + it)
        gd = dev(g)
        va, ka = hip.dots(gd)
        vb, kb = ref.dots(g)
        assert ka == kb and va.shape == vb.shape
        assert np.abs(va - vb).max() <= 1e-12 * max(1.0, np.abs(vb).max()), (it, np.abs(va - vb).max())
        push = it > 0 and it != 5          # iteration 5: the candidate is discarded
        hip.commit(push)
        ref.commit(push)
        k = len(ref.S)
        cs, cy, cg, t = rng.standard_normal(k), rng.standard_normal(k), -0.7, 0.1 + 0.01 * it
        sa = hip.update(cs, cy, cg, t, xd, gd)
        sb = ref.update(cs, cy, cg, t, x, g)
        assert abs(sa - sb) <= 1e-12 * max(1.0, abs(sb))
        assert np.abs(xd.cpu().numpy() - x).max() <= 1e-13 * max(1.0, np.abs(x).max())
    with pytest.raises(RuntimeError):       # protocol errors are reported, not ignored
        hip.dots(gd)
        hip.update(cs, cy, cg, t, xd, gd)
    hip.close()


_STRESS_BITS = {'hartree': ['hartree'], 'tf': ['tf'], 'vw': ['vw'], 'wt_nl': ['wt_nl'], 'lda_x': ['lda_x'], 'pz_c': ['pz_c'],
                'pw_c': ['pw_c'], 'chachiyo_c': ['chachiyo_c'], 'pbe_x': ['pbe_x'], 'pbe_c': ['pbe_c'],
                'wgc99': ['tf', 'vw', 'wgc99_nl'], 'lkt': ['vw', 'gga_k']}


@pytest.mark.parametrize('case', ['g16r', 'gmix', 'g18t'])
def test_stress_matches_reference_get_stress(case):
    """ofdft_stress per term against the reference's autograd get_stress (power-of-two cubic / triclinic grids and a
    generic-path grid), one term at a time and all terms in one call"""
    g = load('stress.npz')
    box, den, vext, chi, n_elec = cases.make_inputs(case)
    eng = Engine(den.shape, DEV).set_cell(dev(box))
    for name, bits in list(_STRESS_BITS.items()) + [('pgs', ['vw', 'gga_k']), ('vwgtf1', ['vw', 'vwgtf']), ('vwgtf2', ['vw', 'vwgtf'])]:
        sig = eng.set_terms(bits, {'ggak_kind': 1.0 if name == 'pgs' else 0.0,
                                   'vwgtf_kind': 2.0 if name == 'vwgtf2' else 1.0}).stress(dev(den))
        tot = sum(sig[b] for b in bits)
        ref = g['%s_%s' % (case, name)]
        assert np.abs(tot - ref).max() <= 2e-10 * np.abs(ref).max(), (case, name, np.abs(tot - ref).max() / np.abs(ref).max())
    allbits = ['hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']
    sig = eng.set_terms(allbits).stress(dev(den))
    ref = g[case + '_hartree'] + g[case + '_wgc99'] + g[case + '_pbe_x'] + g[case + '_pbe_c']
    assert np.abs(sum(sig.values()) - ref).max() <= 2e-10 * np.abs(ref).max()
    eng.close()


def test_ion_electron_stress_matches_reference():
    from professad_amd.ions import ion_electron_stress, recpot_table
    g, ions = load('stress.npz'), load('ions.npz')
    tab = recpot_table(ions['recpot_raw'], float(ions['recpot_kmax']))
    for tag, shape, order, dk in (('a', (32, 32, 32), 10, dict(seed=8, n0=0.03, amp=0.5)),
                                  ('b', (16, 20, 24), 6, dict(seed=7, n0=0.05, amp=0.5))):
        eng = Engine(shape, DEV)
        den = dev(synth.smooth_density(shape, **dk))
        for o in (None, order):
            s = ion_electron_stress(eng, ions[tag + '_box'], den, [(ions[tag + '_frac'], tab)], pme_order=o)
            ref = g['%s_ion_electron_%s' % (tag, 'exact' if o is None else 'pme%d' % o)]
            assert np.abs(s - ref).max() <= 1e-10 * np.abs(ref).max(), (tag, o)
        eng.close()


def test_get_stress_protocol_on_native_terms():
    """the reference's get_stress recipe (functional_tools.py:94-99: box_vecs.requires_grad, den * vol.detach() / vol,
    autograd.grad w.r.t. box_vecs, stress = dEdcell^T B / vol) applied to the native drop-in terms"""
    g = load('stress.npz')
    box, den, vext, chi, n_elec = cases.make_inputs('gmix')

    def get_stress(f):
        b = dev(box).clone().requires_grad_(True)
        vol = torch.abs(torch.linalg.det(b))
        E = f(b, dev(den) * vol.detach() / vol)
        dEdcell = torch.autograd.grad(E, b)[0].T
        return (dEdcell @ b.detach() / vol.detach()).cpu().numpy()

    for f, ref in ((F.Hartree, g['gmix_hartree']), (F.WangGovindCarter99(), g['gmix_wgc99']),
                   (F.NativeTerms(['pbe']), g['gmix_pbe_x'] + g['gmix_pbe_c'])):
        s = get_stress(f)
        assert np.abs(s - ref).max() <= 2e-10 * np.abs(ref).max()


def test_ion_ion_known_answers_forces_and_stress():
    """ofdft_ion_ion against the reference's known-answer energies (tests/test_ion_utils.py:12-147 data), and its forces /
    stress against the oracle's analytic forms (themselves checked by finite differences in the CPU suite)"""
    import json
    from oracle import ionion as ii
    from professad_amd.ions import ion_ion
    doc = json.load(open(os.path.join(GOLDEN, 'ion_ion_known_answers.json')))
    eng = Engine((16, 16, 16), DEV)
    E = {}
    for c in doc['cases']:
        box = np.array(c['box'], dtype=np.float64)
        frac = np.array(c['frac'], dtype=np.float64) if c['frac'] is not None else np.array(c['cart'], dtype=np.float64) @ np.linalg.inv(box)
        z = np.array(c['charges'], dtype=np.float64)
        E[c['name']], F, S = ion_ion(eng, box, frac, z, Rc=12 * c['h_max'])
        if c['expected'] is not None:
            assert abs(E[c['name']] - c['expected']) / len(z) < 1e-10, (c['name'], E[c['name']])
        if c['name'] in ('Si', 'SiO2'):
            Rc = 12 * c['h_max']
            Rd = float(np.sqrt((1.0 / np.sqrt(np.sum(np.linalg.inv(box.T) ** 2, axis=1))).max() * Rc / 3))
            Fo, So = ii.forces_stress(box, frac @ box, z, Rc, Rd)
            assert np.abs(F - Fo).max() < 1e-11 and np.abs(S - So).max() < 1e-12, c['name']
    assert abs(4 * E['NaCl_fcc'] - E['NaCl_two'] - doc['madelung']) < 1e-10
    # the reference's default parameters (Rc = None) on its FD-stress cell
    box = np.array([[6.5, -0.13, 0.25], [-0.33, 7.21, 0.24], [0.55, 0.04, 6.78]])
    frac = np.array([[0, 0, 0], [0.35, 0.65, 0.45]])
    Ed, Fd, Sd = ion_ion(eng, box, frac, [1.0, 1.0])
    Rc, Rd = ii.heuristics(box)
    Fo, So = ii.forces_stress(box, frac @ box, np.array([1.0, 1.0]), Rc, Rd)
    assert abs(Ed - ii.energy(box, frac @ box, np.array([1.0, 1.0]), Rc, Rd)) < 1e-11
    assert np.abs(Fd - Fo).max() < 1e-11 and np.abs(Sd - So).max() < 1e-12
    eng.close()


def test_pauli_gaussian_members_and_all_pipelines_for_gga_kinetic():
    """the kinetic GGA through every engine pipeline, together with PBE (shared gradient / divergence)"""
    gold = load('terms_g16r.npz')
    box, den, vext, chi, n_elec = cases.make_inputs('g16r')
    eng = Engine(den.shape, DEV).set_cell(dev(box)).set_terms(['vw', 'gga_k', 'pbe_x', 'pbe_c'], {'ggak_kind': 0.0})
    Eref = float(gold['E_lkt']) + float(gold['E_pbe_x']) + float(gold['E_pbe_c'])
    vref = gold['v_lkt'] + gold['v_pbe_x'] + gold['v_pbe_c']
    for mode in (0, 1, 2):
        eng.set_option(0, mode)
        E, v = eng.energy_potential(dev(den))
        assert abs(sum(E.values()) - Eref) <= E_RTOL * abs(Eref), mode
        assert relerr(v.cpu().numpy(), vref) < V_RTOL, mode
    eng.close()


@pytest.mark.parametrize('case', ['g16r', 'g16s'])
@pytest.mark.parametrize('member', ['pgsl025', 'pgslr'])
def test_laplacian_dependent_pauli_gaussian_through_every_pipeline(case, member):
    """PGSL0.25 (the reference's default PauliGaussian) and PGSLr: the z-fused split-derivative chain carries lap n and
    lap(df/dL) as one more spectrum each way; the unfused pipeline and the unsplit form (which falls back to it) agree;
    all against the reference's goldens (tests/tools_for_tests.py:86-118 is the closed form)"""
    gold = load('terms_%s.npz' % case)
    box, den, vext, chi, n_elec = cases.make_inputs(case)
    params = {'ggak_kind': 1.0, 'ggak_mu': 40 / 27, 'ggak_beta': 0.25}
    if member == 'pgslr':
        params.update(ggak_lambda=0.4, ggak_sigma=0.2)
    Eref, vref = float(gold['E_' + member]), gold['v_' + member]
    eng = Engine(den.shape, DEV).set_cell(dev(box)).set_terms(['vw', 'gga_k'], params)
    counts = {}
    for mode, gsplit in ((0, 1), (1, 1), (0, 0), (2, 1)):
        eng.set_option(0, mode).set_option(6, gsplit)
        E, v = eng.energy_potential(dev(den))
        assert abs(sum(E.values()) - Eref) <= E_RTOL * abs(Eref), (mode, gsplit)
        assert relerr(v.cpu().numpy(), vref) < V_RTOL, (mode, gsplit)
        counts[(mode, gsplit)] = int(eng.query(4))
    assert counts[(0, 1)] < counts[(1, 1)]            # the fused chain really ran (fewer launches than the unfused pipeline)
    # together with PBE and Hartree in one evaluation (shared gradient chain), closure form
    eng.set_option(0, 0).set_option(6, 1)
    eng.set_terms(['hartree', 'vw', 'gga_k', 'pbe_x', 'pbe_c'], params)
    E, v = eng.energy_potential(dev(den))
    Eref2 = Eref + float(gold['E_hartree']) + float(gold['E_pbe_x']) + float(gold['E_pbe_c'])
    vref2 = vref + gold['v_hartree'] + gold['v_pbe_x'] + gold['v_pbe_c']
    assert abs(sum(E.values()) - Eref2) <= E_RTOL * abs(Eref2)
    assert relerr(v.cpu().numpy(), vref2) < V_RTOL
    eng.close()


@pytest.mark.parametrize('case', ['g16r', 'gmix', 'g18t'])
def test_stress_of_laplacian_dependent_pauli_gaussian_and_wt_style(case):
    """PGSL0.25 / PGSLr (derived Hessian term, oracle/stress.py::pauli_gaussian) and the Wang-Teter style functional with
    f = exp (tools_for_tests.py:310-364) against the reference's get_stress"""
    g = load('stress.npz')
    box, den, vext, chi, n_elec = cases.make_inputs(case)
    eng = Engine(den.shape, DEV).set_cell(dev(box))
    for member, params in (('pgsl025', {'ggak_kind': 1.0, 'ggak_mu': 40 / 27, 'ggak_beta': 0.25}),
                           ('pgslr', {'ggak_kind': 1.0, 'ggak_mu': 40 / 27, 'ggak_beta': 0.25, 'ggak_lambda': 0.4, 'ggak_sigma': 0.2})):
        sig = eng.set_terms(['vw', 'gga_k'], params).stress(dev(den))
        tot = sig['vw'] + sig['gga_k']
        ref = g['%s_%s' % (case, member)]
        assert np.abs(tot - ref).max() <= 2e-10 * np.abs(ref).max(), (member, np.abs(tot - ref).max(), np.abs(ref).max())
    sig = eng.set_terms(['tf', 'vw', 'wt_nl'], {'wts_kind': 1.0}).stress(dev(den))      # native weights f - f' X, f'
    ref = g['%s_wts_exp' % case]
    tot = sig['tf'] + sig['vw'] + sig['wt_nl']
    assert np.abs(tot - ref).max() <= 2e-10 * np.abs(ref).max(), np.abs(tot - ref).max()
    eng.close()

    def get_stress(f):        # the reference's recipe, functional_tools.py:94-99
        b = dev(box).clone().requires_grad_(True)
        vol = torch.abs(torch.linalg.det(b))
        E = f(b, dev(den) * vol.detach() / vol)
        dEdcell = torch.autograd.grad(E, b)[0].T
        return (dEdcell @ b.detach() / vol.detach()).cpu().numpy()
    for f, key in ((F.WangTeterStyleFunctional((5 / 6, 5 / 6, torch.exp)), 'wts_exp'), (F.PauliGaussian(), 'pgsl025')):
        s = get_stress(f)
        ref = g['%s_%s' % (case, key)]
        assert np.abs(s - ref).max() <= 2e-10 * np.abs(ref).max(), key


@pytest.mark.parametrize('shape', [(17, 18, 15), (5, 7, 9), (33, 35, 31), (3, 5, 255), (255, 3, 5), (2, 257, 6), (7, 129, 67)])
def test_generic_extent_paths_agree(shape):
    """non power-of-two grids: the chirp-z (Bluestein) line transforms against the plain DFT kernels and numpy (odd and even
    row counts -- the z passes transform rows in pairs --, every padded length 16 ... 1024, remainder planes of the layout)"""
    rng = np.random.default_rng(4)
    x = rng.standard_normal(shape)
    eng = Engine(shape, DEV)
    ref = np.fft.rfftn(x)
    res = {}
    for opt in (1, 0):
        eng.set_option(5, opt)
        yk = eng.rfftn(dev(x))
        assert relerr(yk.cpu().numpy(), ref) < 1e-13, opt
        res[opt] = eng.irfftn(yk).cpu().numpy()
        assert relerr(res[opt], x) < 1e-13, opt
    eng.close()


def test_fcc_aluminium_end_to_end_against_profess4_value():
    """The reference's own end-to-end anchor (tests/test_match_profess4.py:12-24): fcc-Al primitive cell, 18^3 grid,
    IonIon + IonElectron + Hartree + WangTeter + PBE, density optimised to ntol = 1e-7 -> -57.183329401794985 eV
    (PROFESS 4.0, atol 1e-4).  Here every piece is native: ionic potential from the recpot table, the closure on a
    non power-of-two grid (chirp-z transforms), the device L-BFGS, and the ion-ion sum."""
    from professad_amd.ions import ion_ion, ionic_potential, recpot_table
    from professad_amd.optimize import EV_PER_HA, optimize_density
    g = load('ions.npz')
    tab = recpot_table(g['recpot_raw'], float(g['recpot_kmax']))
    box = 4.050 / 0.529177210903 * np.array([[0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]])
    frac = np.zeros((1, 3))
    shape = (18, 18, 18)
    eng = Engine(shape, DEV).set_cell(dev(box))
    vext = ionic_potential(eng, box, [(frac, tab)])
    eng.set_terms(F.NativeTerms(['ion_electron', 'hartree', 'wt', 'pbe']).names)
    res = optimize_density(eng, float(tab[2]), vext, volume=abs(np.linalg.det(box)), ntol=1e-7)
    assert res['converged']
    E_ii, _, _ = ion_ion(eng, box, frac, [float(tab[2])])
    E_eV = (res['E_Ha'] + E_ii) * EV_PER_HA
    assert abs(E_eV - -57.183329401794985) < 1e-4, E_eV
    eng.close()


def test_optimisers_agree_and_convergence_measures_are_consistent():
    """The reference's tests/test_den_opt.py:43-75 on the native path: fcc-Al primitive cell with Hartree + LKT + PBE -- the
    fixed-step L-BFGS and the two-point gradient descent (n_method='TPGD') reach the same energy (3 decimal places in eV); and
    at the converged state max |dE/dchi| from the closure equals the formula from dE/dn (rtol 1e-10)."""
    from professad_amd.ions import ionic_potential, recpot_table
    from professad_amd.optimize import EV_PER_HA, optimize_density
    g = load('ions.npz')
    tab = recpot_table(g['recpot_raw'], float(g['recpot_kmax']))
    box = 4.050 / 0.529177210903 * np.array([[0.5, 0.5, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5]])
    vol = abs(np.linalg.det(box))
    shape = (20, 20, 20)
    eng = Engine(shape, DEV).set_cell(dev(box))
    vext = ionic_potential(eng, box, [(np.zeros((1, 3)), tab)])
    eng.set_terms(F.NativeTerms(['ion_electron', 'hartree', 'tf', 'lkt', 'pbe']).names, {'ggak_kind': 0})
    n_el = float(tab[2])
    r1 = optimize_density(eng, n_el, vext, volume=vol, ntol=1e-4)
    r2 = optimize_density(eng, n_el, vext, volume=vol, ntol=1e-4, n_conv_cond_count=5, n_method='TPGD')
    assert r1['converged'] and r2['converged']
    assert abs(r1['E_Ha'] - r2['E_Ha']) * EV_PER_HA < 5e-4
    with pytest.raises(ValueError):
        optimize_density(eng, n_el, vext, volume=vol, n_method='SD')
    # convergence measure: chi.grad / dV of the closure against 2 c chi (dE/dn - mu) from the potential entry (system.py:377-447)
    chi, den = r1['chi'], r1['den']
    _, mu, grad = eng.energy_grad_chi(chi, n_el, vext)
    _, dEdn = eng.energy_potential(den, vext)
    dV = vol / np.prod(shape)
    c = n_el / (float((chi * chi).sum()) * dV)
    mu2 = float((dEdn * den).sum()) * dV / n_el
    want = c * 2.0 * chi * (dEdn - mu2)
    assert abs(mu - mu2) < 1e-10 * abs(mu)
    assert abs(float(grad.abs().max()) / dV - float(want.abs().max())) <= 1e-10 * float(want.abs().max())
    # conv_target='euler' (system.py:377-412,873-874): max |mu - dE/dn| at the density of the step's last closure call, formed
    # from that call's gradient without another evaluation -- equals the direct evaluation at the same chi
    resid = torch.where(chi != 0, grad / (2.0 * c * dV * chi), torch.zeros_like(chi))
    assert abs(float(resid.abs().max()) - float((mu2 - dEdn).abs().max())) <= 1e-9 * float((mu2 - dEdn).abs().max())
    r3 = optimize_density(eng, n_el, vext, volume=vol, ntol=2e-5, conv_target='euler')
    assert r3['converged'] and abs(r3['E_Ha'] - r1['E_Ha']) * EV_PER_HA < 5e-4
    eng.close()


def test_native_lbfgs_recursion_walks_the_numpy_recursions_path():
    """ofdft_lbfgs_direction (curvature test, commit, Gram blocks, two-loop recursion on coefficients inside the library) against the
    numpy statement of the same host logic on the same device sweeps: a non-quadratic test function, history wrap-around, rejected
    pairs, 60 outer steps -- identical losses and iterates to round-off"""
    from professad_amd.optimize import HipLbfgsBackend, VectorFreeLBFGS

    class NumpyRecursion(HipLbfgsBackend):          # the same sweeps, but VectorFreeLBFGS finds no `direction` and keeps its numpy path
        def __getattribute__(self, name):
            if name == 'direction':
                raise AttributeError(name)
            return super().__getattribute__(name)
    n = 4096
    rng = np.random.default_rng(3)
    A = rng.standard_normal((64, n)) / 8.0
    At, b = dev(A), dev(rng.standard_normal(n))
    diag = dev(np.linspace(0.05, 3.0, n))

    def make(x):
        def closure():
            xx = x.detach().clone().requires_grad_(True)
            f = 0.5 * ((At @ xx) ** 2).sum() + 0.5 * (diag * xx * xx).sum() - b @ xx + 0.05 * (xx ** 4).sum()
            f.backward()
            return float(f.detach()), xx.grad.detach()
        return closure
    xa = torch.zeros(n, dtype=torch.double, device=DEV)
    xb = torch.zeros(n, dtype=torch.double, device=DEV)
    oa = VectorFreeLBFGS(xa, HipLbfgsBackend(n, 8, DEV))
    ob = VectorFreeLBFGS(xb, NumpyRecursion(n, 8, DEV))
    assert hasattr(oa.b, 'direction') and not hasattr(ob.b, 'direction')
    for step in range(60):
        la, lb = oa.step(make(xa)), ob.step(make(xb))
        assert abs(la - lb) <= 1e-10 * max(1.0, abs(lb)), step
    assert oa.total_iter == ob.total_iter and oa.func_evals == ob.func_evals
    assert float((xa - xb).abs().max()) <= 1e-8 * float(xb.abs().max())


def test_bcc_lithium_end_to_end_against_profess4_value():
    """second anchor of tests/test_match_profess4.py:26-37: bcc-Li, 18^3, IonIon + IonElectron + Hartree + SmargiassiMadden
    + PBE -> -14.741886997024537 eV (PROFESS 4.0, atol 1e-4)"""
    from professad_amd.ions import ion_ion, ionic_potential, recpot_table
    from professad_amd.optimize import EV_PER_HA, optimize_density
    g = load('recpots.npz')
    tab = recpot_table(g['li_raw'], float(g['li_kmax']))
    assert tab[2] == 1
    box = 3.48 / 0.529177210903 * np.eye(3)
    frac = np.array([[0.0, 0.0, 0.0], [0.5, 0.5, 0.5]])
    eng = Engine((18, 18, 18), DEV).set_cell(dev(box))
    vext = ionic_potential(eng, box, [(frac, tab)])
    eng.set_terms(['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'pbe_x', 'pbe_c'], {'wt_alpha': 0.5, 'wt_beta': 0.5})
    res = optimize_density(eng, 2.0, vext, volume=abs(np.linalg.det(box)), ntol=1e-7)
    assert res['converged']
    E_ii, _, _ = ion_ion(eng, box, frac, [1.0, 1.0])
    assert abs((res['E_Ha'] + E_ii) * EV_PER_HA - -14.741886997024537) < 1e-4
    eng.close()


def test_total_forces_are_energy_derivatives_at_the_ground_state():
    """The reference's force test (tests/test_forces.py:11-42) run natively: Li2 in a triclinic cell on the odd grid its
    ecut2shape gives, IonIon + IonElectron + Hartree + WangTeter + PBE.  Forces from the converged density (analytic
    ion-electron + ion-ion) against central differences of the re-optimised total energy (Hellmann-Feynman), atol 1e-4 eV/A."""
    from professad_amd.ions import ion_electron_forces, ion_ion, ionic_potential, recpot_table
    from professad_amd.optimize import EV_PER_HA, optimize_density
    A = 0.529177210903
    g = load('recpots.npz')
    tab = recpot_table(g['li_raw'], float(g['li_kmax']))
    box = np.array([[3.54, -0.13, 0.25], [-0.33, 3.82, 0.24], [0.55, 0.04, 3.45]]) / A
    kcut = np.sqrt(2 * 1600 / EV_PER_HA)
    shape = tuple(int(1 + 2 * np.ceil(kcut / (2 * np.pi / np.sqrt((box ** 2).sum(1)[i])))) for i in range(3))   # system.py:75-89
    assert all(s % 2 == 1 for s in shape)
    frac0 = np.array([[0.0, 0.0, 0.0], [0.35, 0.65, 0.45]])
    inv = np.linalg.inv(box)
    vol = abs(np.linalg.det(box))
    eng = Engine(shape, DEV).set_cell(dev(box)).set_terms(F.NativeTerms(['ion_electron', 'hartree', 'wt', 'pbe']).names)

    def ground_state(frac, chi0=None):
        vext = ionic_potential(eng, box, [(frac, tab)])
        res = optimize_density(eng, 2.0, vext, chi0=chi0, volume=vol, ntol=1e-8)
        assert res['converged']
        return res, res['E_Ha'] + ion_ion(eng, box, frac, [1.0, 1.0])[0]

    res0, E0 = ground_state(frac0)
    F_ie = ion_electron_forces(eng, box, res0['den'], [(frac0, tab)])[0]
    F_tot = (F_ie + ion_ion(eng, box, frac0, [1.0, 1.0])[1]) * EV_PER_HA / A           # eV / Angstrom
    eps = 1e-4 / A
    for ion, i in ((0, 0), (1, 1), (1, 2)):
        E = []
        for sgn in (+1, -1):
            cart = frac0 @ box
            cart[ion, i] += sgn * eps
            E.append(ground_state(cart @ inv, chi0=res0['chi'])[1])
        fd = -(E[0] - E[1]) / (2 * eps) * EV_PER_HA / A
        assert abs(fd - F_tot[ion, i]) < 1e-4, (ion, i, fd, F_tot[ion, i])
    eng.close()


def test_total_stress_is_the_strain_derivative_at_the_ground_state():
    """Native counterpart of the reference's stress checks (tests/test_stress.py, tests/test_ion_utils.py:149-180): total
    stress = density-functional terms + ion-electron + ion-ion at the converged density, against central differences of
    the re-optimised total energy under symmetric strains at fixed fractional coordinates."""
    from professad_amd.ions import ion_electron_stress, ion_ion, ionic_potential, recpot_table
    from professad_amd.optimize import EV_PER_HA, optimize_density
    A = 0.529177210903
    g = load('recpots.npz')
    tab = recpot_table(g['li_raw'], float(g['li_kmax']))
    box0 = np.array([[3.54, -0.13, 0.25], [-0.33, 3.82, 0.24], [0.55, 0.04, 3.45]]) / A
    kcut = np.sqrt(2 * 1000 / EV_PER_HA)
    shape = tuple(int(1 + 2 * np.ceil(kcut / (2 * np.pi / np.sqrt((box0 ** 2).sum(1)[i])))) for i in range(3))
    frac = np.array([[0.0, 0.0, 0.0], [0.35, 0.65, 0.45]])
    names = F.NativeTerms(['ion_electron', 'hartree', 'wt', 'pbe']).names
    eng = Engine(shape, DEV)

    def ground_state(box, chi0=None):
        eng.set_cell(dev(box)).set_terms(names)
        vext = ionic_potential(eng, box, [(frac, tab)])
        res = optimize_density(eng, 2.0, vext, chi0=chi0, volume=abs(np.linalg.det(box)), ntol=1e-9)
        assert res['converged']
        return res, res['E_Ha'] + ion_ion(eng, box, frac, [1.0, 1.0])[0]

    res0, E0 = ground_state(box0)
    eng.set_cell(dev(box0)).set_terms(names)
    sig = sum(eng.stress(res0['den']).values()) + ion_electron_stress(eng, box0, res0['den'], [(frac, tab)]) \
        + ion_ion(eng, box0, frac, [1.0, 1.0])[2]
    vol = abs(np.linalg.det(box0))
    h = 2e-4
    for i, j in ((0, 0), (2, 2), (0, 1), (1, 2)):
        eps = np.zeros((3, 3))
        eps[i, j] += 0.5 * h
        eps[j, i] += 0.5 * h
        Ep = ground_state(box0 + box0 @ eps, chi0=res0['chi'])[1]
        Em = ground_state(box0 - box0 @ eps, chi0=res0['chi'])[1]
        fd = (Ep - Em) / (2 * h * vol)
        assert abs(fd - sig[i, j]) < 2e-7, (i, j, fd, sig[i, j])
    # pressure = -tr(sigma) / 3 (functional_tools.py:104-131)
    assert abs(sig[0, 1] - sig[1, 0]) < 1e-14
    eng.close()


def test_pme_and_exact_structure_factors_give_the_same_physics():
    """tests/test_particle_mesh_ewald.py:68-89 natively: order-20 particle-mesh Ewald against the exact structure factor
    through the whole chain -- optimised energy, density, ion-electron forces and stress (np.allclose defaults) for the
    reference's bcc-Li cell; and non-zero forces for displaced ions (where the spline error is not symmetry-cancelled)."""
    from professad_amd.ions import ion_electron_forces, ion_electron_stress, ionic_potential, recpot_table
    from professad_amd.optimize import optimize_density
    g = load('recpots.npz')
    tab = recpot_table(g['li_raw'], float(g['li_kmax']))
    box = 6.96 * np.eye(3)
    eng = Engine((25, 25, 25), DEV).set_cell(dev(box)).set_terms(F.NativeTerms(['ion_electron', 'hartree', 'wt', 'pbe']).names)
    for frac, tols in ((np.array([[0.0, 0.0, 0.0], [0.5, 0.5, 0.5]]), dict()),
                       (np.array([[0.0, 0.0, 0.0], [0.47, 0.52, 0.5]]), dict(rtol=2e-4, atol=1e-7))):
        out = {}
        for order in (None, 20):
            vext = ionic_potential(eng, box, [(frac, tab)], pme_order=order)
            res = optimize_density(eng, 2.0, vext, volume=6.96 ** 3, ntol=1e-9)
            assert res['converged']
            out[order] = (res['E_Ha'], res['den'].cpu().numpy(),
                          ion_electron_forces(eng, box, res['den'], [(frac, tab)], pme_order=order)[0],
                          ion_electron_stress(eng, box, res['den'], [(frac, tab)], pme_order=order))
        for a, b in zip(out[None], out[20]):
            assert np.allclose(a, b, **tols)
    assert np.abs(out[None][2]).max() > 1e-3
    eng.close()


# ------------------------------------------------------------------------------- hipGraph replay of the closure
@pytest.mark.parametrize('shape,cfg', [((32, 32, 32), 'cfg1'), ((64, 64, 64), 'cfg3'), ((16, 32, 64), 'cfg2'),
                                       ((256, 32, 64), 'cfg3'),         # (256-point x lines: the cross-wave x pass inside a captured graph)
                                       ((15, 17, 19), 'cfg3'), ((53, 53, 53), 'cfg2')])      # (odd extents: the host-free chirp-z closure, round 4)
def test_graph_replay_is_bitwise_the_kernel_by_kernel_path(shape, cfg):
    """ofdft_energy_grad_chi captures a hipGraph on the second call with the same arguments and replays it afterwards:
    same bits as launching kernel by kernel, tracks in-place changes of chi, and is dropped when the configuration changes"""
    from professad_amd import _native as N
    rng = np.random.default_rng(11)
    box = dev(synth.cubic_cell(shape[0]))
    den = synth.smooth_density(shape, seed=3)
    vext = dev(synth.random_potential(shape, seed=4))
    chi = dev(np.sqrt(den))
    nel = 7.0
    names = F.NativeTerms(_CFG_TERMS[cfg]).names
    # (graph-eligible grids keep the WGC99 part inside the combine kernel: same setting for the comparison engine)
    # (the persistent small-grid kernel would serve 32^3 cfg1 instead: off, this test is about the replay)
    plain = Engine(shape, DEV).set_cell(box).set_terms(names).set_option(N.OPT_GRAPH, 0).set_option(4, 0).set_option(N.OPT_RESIDENT, 0)
    eng = Engine(shape, DEV).set_cell(box).set_terms(names).set_option(N.OPT_RESIDENT, 0)
    for rep in range(5):
        if rep == 3:
            chi.mul_(1.0 + 0.05 * torch.as_tensor(rng.random(shape), device=DEV))      # new data behind the same pointer
        Ea, mua, ga = plain.energy_grad_chi(chi, nel, vext)
        Eb, mub, gb = eng.energy_grad_chi(chi, nel, vext)
        assert Ea == Eb and mua == mub and torch.equal(ga, gb), rep
        del ga, gb          # let the allocator hand the same gradient buffer out again
    assert plain.query(N.Q_GRAPH_REPLAYS) == 0
    assert eng.query(N.Q_GRAPH_REPLAYS) >= 2
    assert eng.query(N.Q_LAUNCH_COUNT) == plain.query(N.Q_LAUNCH_COUNT) and eng.query(N.Q_FFT_COUNT) == plain.query(N.Q_FFT_COUNT)
    # another term set: the captured graphs no longer apply
    other = F.NativeTerms(['hartree', 'tf', 'vw', 'pz']).names
    Ea, mua, ga = plain.set_terms(other).energy_grad_chi(chi, nel, None)
    Eb, mub, gb = eng.set_terms(other).energy_grad_chi(chi, nel, None)
    assert Ea == Eb and mua == mub and torch.equal(ga, gb)
    plain.close()
    eng.close()


# ------------------------------------------------------------------------------- persistent small-grid kernel
@pytest.mark.parametrize('fault', ['does_not_fit', 'barrier_timeout'])
def test_resident_kernel_that_cannot_run_falls_back_to_the_staged_pipeline(fault):
    """round-2 advice: the persistent kernel needs its N workgroups co-resident.  (a) the occupancy check says it does not fit
    -> never launched, the call is served by the graph replay / the staged pipeline; (b) a launch whose grid barriers cannot
    complete (test hook: one workgroup short) -> ANY workgroup's time-out is published through a shared word, the call still
    returns the right numbers (re-run on the staged path) and the kernel is switched off for the context"""
    from professad_amd import _native as N
    n = 16
    shape = (n, n, n)
    box = dev(synth.triclinic_cell(n / 4.0))
    den = synth.smooth_density(shape, seed=3, amp=0.5)
    chi, vext = dev(np.sqrt(den)), dev(synth.random_potential(shape, seed=4))
    names = F.NativeTerms(['ion_electron', 'hartree', 'wt', 'pbe']).names
    staged = Engine(shape, DEV).set_cell(box).set_terms(names).set_option(N.OPT_RESIDENT, 0)
    Ea, mua, ga = staged.energy_grad_chi(chi, 9.0, vext)
    eng = Engine(shape, DEV).set_cell(box).set_terms(names).set_option(N.OPT_RESIDENT, 1)
    if fault == 'barrier_timeout':
        E0, mu0, g0 = eng.energy_grad_chi(chi, 9.0, vext)                 # a healthy launch first (the counters run on)
        assert eng.query(N.Q_RESIDENT_EVALS) == 1
    eng.set_option(N.OPT_TEST_FAULT, 2 if fault == 'does_not_fit' else 1)
    served_before = eng.query(N.Q_RESIDENT_EVALS)
    for rep in range(3):            # the faulty call itself, then calls on the switched-off context (graph capture + replay)
        Eb, mub, gb = eng.energy_grad_chi(chi, 9.0, vext)
        for k in Ea:
            assert abs(Ea[k] - Eb[k]) <= 1e-12 * max(abs(Ea[k]), 1e-2), (rep, k)
        assert abs(mua - mub) <= 1e-12 * max(1.0, abs(mua)) and float((ga - gb).abs().max()) <= 1e-12 * float(ga.abs().max())
    assert eng.query(N.Q_RESIDENT_EVALS) == served_before
    assert eng.query(N.Q_RESIDENT_FALLBACKS) == (1 if fault == 'barrier_timeout' else 0)
    E2, v2 = eng.energy_potential(chi * chi, vext)                        # the density entry takes the staged path as well
    E1, v1 = staged.energy_potential(chi * chi, vext)
    assert all(abs(E1[k] - E2[k]) <= 1e-12 * max(abs(E1[k]), 1e-2) for k in E1) and float((v1 - v2).abs().max()) <= 1e-12 * float(v1.abs().max())
    staged.close()
    eng.close()


_RES_TERMS = {'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'pz'], 'cfg2': ['ion_electron', 'hartree', 'wt', 'pz'],
              'local': ['tf', 'lda_x', 'pw_c'], 'vw_only': ['vw']}


@pytest.mark.parametrize('n', [16, 32, 64])
@pytest.mark.parametrize('cfg', ['cfg1', 'cfg2', 'local', 'vw_only', 'wt_ab', 'gtf', 'wt_pbe', 'lkt_pbe', 'pbe_only', 'pg1', 'cfg3', 'wgc_lda'])
def test_resident_kernel_matches_the_staged_pipeline(n, cfg):
    """cubic 16^3 / 32^3 / 64^3, term sets without gradient-dependent or WGC99 parts: the closure evaluation as ONE persistent
    kernel (csrc/resident.hip) against the staged pipeline (itself pinned to the oracle and the reference's goldens),
    energies / mu / chi.grad to 1e-12; repeated calls (the barrier counter runs on), new data, the untimed form"""
    from professad_amd import _native as N
    shape = (n, n, n)
    rng = np.random.default_rng(5)
    box = dev(synth.triclinic_cell(n / 4.0))
    den = synth.smooth_density(shape, seed=3, amp=0.5) * (1 + 0.1 * rng.random(shape))
    vext = dev(synth.random_potential(shape, seed=4))
    chi = dev(np.sqrt(den) * np.where(rng.random(shape) < 0.3, -1.0, 1.0))        # chi may change sign: n = c chi^2
    nel = 9.0
    params = None
    if cfg == 'wt_ab':
        names, params = ['ion_electron', 'tf', 'vw', 'wt_nl', 'lda_x', 'pz_c'], {'wt_alpha': 0.7, 'wt_beta': 0.9}
    elif cfg == 'gtf':
        names, params = ['ion_electron', 'hartree', 'vwgtf', 'lda_x', 'pw_c'], {'vwgtf_kind': 2}
    elif cfg == 'wt_pbe':                                   # the reference's standard term set (tests/test_den_opt.py:59)
        names = F.NativeTerms(['ion_electron', 'hartree', 'wt', 'pbe']).names
    elif cfg == 'lkt_pbe':                                  # tests/test_den_opt.py:44
        names, params = F.NativeTerms(['ion_electron', 'hartree', 'tf', 'lkt', 'pbe']).names, {'ggak_kind': 0}
    elif cfg == 'pbe_only':                                 # a gradient term without Hartree: the chi^2 spectrum feeds the gradient only
        names = ['tf', 'pbe_x', 'pbe_c']
    elif cfg == 'pg1':                                      # Pauli-Gaussian, mu-only member
        names, params = ['ion_electron', 'hartree', 'vw', 'gga_k', 'lda_x', 'pz_c'], {'ggak_kind': 1, 'ggak_mu': 1.0}
    elif cfg == 'cfg3':                                     # the bench's term set: WGC99 + PBE
        names = F.NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe']).names
    elif cfg == 'wgc_lda':                                  # WGC99 with other exponents, no gradient term
        names, params = F.NativeTerms(['ion_electron', 'wgc99', 'pz']).names, {'wgc_alpha': 1.1, 'wgc_beta': 0.7, 'wgc_gamma': 3.2}
    elif cfg in _RES_TERMS:
        names = F.NativeTerms(_RES_TERMS[cfg]).names if cfg in ('cfg1', 'cfg2') else _RES_TERMS[cfg]
    staged = Engine(shape, DEV).set_cell(box).set_terms(names, params).set_option(N.OPT_RESIDENT, 0)
    res = Engine(shape, DEV).set_cell(box).set_terms(names, params).set_option(N.OPT_RESIDENT, 1)       # timed form first
    ve = vext if 'ion_electron' in names else None
    for rep in range(4):
        if rep == 2:
            chi = chi * (1.0 + 0.05 * torch.as_tensor(rng.random(shape), device=DEV))
        if rep == 3:
            res.set_option(N.OPT_RESIDENT, 2)
        Ea, mua, ga = staged.energy_grad_chi(chi, nel, ve)
        Eb, mub, gb = res.energy_grad_chi(chi, nel, ve)
        for k in Ea:
            assert abs(Ea[k] - Eb[k]) <= 1e-12 * max(abs(Ea[k]), 1e-2), (rep, k, Ea[k], Eb[k])
        assert abs(mua - mub) <= 1e-12 * max(1.0, abs(mua))
        assert float((ga - gb).abs().max()) <= 1e-12 * float(ga.abs().max()), rep
    served = not (n == 64 and cfg in ('wt_pbe', 'lkt_pbe', 'pbe_only', 'pg1', 'cfg3', 'wgc_lda'))     # 64^3 with a gradient or WGC99 term: staged pipeline
    assert res.query(N.Q_RESIDENT_EVALS) == (4 if served else 0) and staged.query(N.Q_RESIDENT_EVALS) == 0
    Ec, muc, _ = res.energy_grad_chi(chi, nel, ve, want_grad=False)                  # energy only
    assert Ec == Eb and muc == mub
    # the density-input entry (ofdft_energy_potential: what the drop-in terms of professad_amd.functionals call)
    den_t = chi * chi * (nel / float((chi * chi).sum() * abs(np.linalg.det(box.cpu().numpy())) / n ** 3))
    before = res.query(N.Q_RESIDENT_EVALS)
    Ea, va = staged.energy_potential(den_t, ve)
    Eb, vb = res.energy_potential(den_t, ve)
    assert res.query(N.Q_RESIDENT_EVALS) == before + (1 if served else 0)
    for k in Ea:
        assert abs(Ea[k] - Eb[k]) <= 1e-12 * max(abs(Ea[k]), 1e-2), (k, Ea[k], Eb[k])
    assert float((va - vb).abs().max()) <= 1e-12 * float(va.abs().max())
    Ec, vc = res.energy_potential(den_t, ve, want_potential=False)
    assert vc is None and Ec == Eb
    staged.close()
    res.close()


def test_resident_kernel_leaves_other_grids_and_terms_to_the_staged_pipeline():
    from professad_amd import _native as N
    chi = dev(np.sqrt(synth.smooth_density((32, 32, 32), seed=3)))
    eng = Engine((32, 32, 32), DEV).set_cell(dev(synth.cubic_cell(32)))
    eng.set_terms(['hartree', 'vw', 'gga_k', 'tf'], {'ggak_kind': 1, 'ggak_mu': 40 / 27, 'ggak_beta': 0.25}).energy_grad_chi(chi, 5.0, None)
    assert eng.query(N.Q_RESIDENT_EVALS) == 0                      # Laplacian-dependent Pauli-Gaussian member: staged pipeline
    eng.set_terms(['hartree', 'tf']).energy_grad_chi(chi, 5.0, None)
    assert eng.query(N.Q_RESIDENT_EVALS) == 1
    eng.close()
    chi = dev(np.sqrt(synth.smooth_density((16, 32, 32), seed=3)))
    eng = Engine((16, 32, 32), DEV).set_cell(dev(synth.cubic_cell(32))).set_terms(['hartree', 'tf'])
    eng.energy_grad_chi(chi, 5.0, None)
    assert eng.query(N.Q_RESIDENT_EVALS) == 0
    eng.close()


@pytest.mark.parametrize('shape,cfg', [((32, 32, 32), 'cfg1'), ((32, 32, 32), 'cfg3'), ((48, 32, 64), 'cfg2')])
def test_calls_are_enqueued_on_the_callers_stream(shape, cfg):
    """the engine works on torch's CURRENT stream (SURVEY 8b threading rule): inputs produced on a side stream just before the call,
    outputs consumed on it right after -- persistent kernel, graph replay and the staged pipeline alike"""
    box = dev(synth.cubic_cell(shape[0]))
    names = F.NativeTerms(_CFG_TERMS[cfg]).names
    base = dev(np.sqrt(synth.smooth_density(shape, seed=3)))
    vext = dev(synth.random_potential(shape, seed=4))
    eng = Engine(shape, DEV).set_cell(box).set_terms(names)
    want = []
    for k in range(3):                                  # default stream
        E, mu, g = eng.energy_grad_chi(base * (1.0 + 0.01 * k), 6.0, vext)
        want.append((E, mu, g.clone()))
    side = torch.cuda.Stream(device=DEV)
    side.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(side):
        for k in range(3):
            chi = base * (1.0 + 0.01 * k)               # produced on the side stream, no synchronisation before the call
            E, mu, g = eng.energy_grad_chi(chi, 6.0, vext)
            gsum = (g - want[k][2]).abs().max()         # consumed on the side stream
            assert E == want[k][0] and mu == want[k][1] and float(gsum) == 0.0, k
    torch.cuda.current_stream(DEV).wait_stream(side)
    eng.close()


def test_contexts_give_their_memory_back():
    """create / evaluate / destroy in a loop (persistent kernel, graph replay, staged, L-BFGS handle): device memory returns"""
    from professad_amd.optimize import HipLbfgsBackend
    shape = (32, 32, 32)
    chi = dev(np.sqrt(synth.smooth_density(shape, seed=3)))
    vext = dev(synth.random_potential(shape, seed=4))
    box = dev(synth.cubic_cell(32))

    def cycle():
        for cfg in ('cfg1', 'cfg3'):
            eng = Engine(shape, DEV).set_cell(box).set_terms(F.NativeTerms(_CFG_TERMS[cfg]).names)
            for _ in range(3):
                eng.energy_grad_chi(chi, 6.0, vext)
            eng.energy_potential(chi * chi, vext)
            eng.close()
        b = HipLbfgsBackend(chi.numel(), 8, DEV)
        b.dots(chi.view(-1))
        b.close()
    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info(DEV)[0]
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info(DEV)[0]
    assert free0 - free1 < 64 * 1024 * 1024, (free0, free1)


def test_missing_vext_is_an_error_not_a_fault():
    """IonElectron without v_ext: every pipeline refuses the call (a null row pointer in a kernel would be a GPU fault)"""
    shape = (16, 16, 16)
    eng = Engine(shape, DEV).set_cell(dev(synth.cubic_cell(16))).set_terms(['ion_electron', 'hartree', 'tf'])
    den = dev(synth.smooth_density(shape, seed=3))
    for mode in (0, 1, 2):
        eng.set_option(0, mode)
        with pytest.raises(RuntimeError, match='needs vext'):
            eng.energy_potential(den, None)
        with pytest.raises(RuntimeError, match='needs vext'):
            eng.energy_grad_chi(torch.sqrt(den), 4.0, None)
    E, v = eng.set_option(0, 0).energy_potential(den, den, want_potential=False)      # energy only: no output array
    assert v is None and E['hartree'] > 0.0
    eng.close()


# ------------------------------------------------------------------------------- lean transcendentals (csrc/fastmath.h)
def test_lean_transcendentals_are_accurate_to_a_few_ulp():
    """the 1/x, log, exp and n^(-1/6) family the fused kernels use instead of the library functions, against numpy
    in extended precision over the ranges a density / PBE argument spans (relative error <= 2e-15)"""
    eng = engine_for((16, 16, 16), DEV)
    rng = np.random.default_rng(5)
    x = np.concatenate([10.0 ** rng.uniform(-30, 30, 200000), 1.0 + rng.uniform(-1e-3, 1e-3, 50000),
                        rng.uniform(0.5, 2.0, 50000), 10.0 ** rng.uniform(-300, 300, 20000)])
    xl = x.astype(np.longdouble)
    checks = [('rcp', x, 1.0 / xl, 2e-16 * 4), ('rsixth', x, xl ** (np.longdouble(-1) / 6), 1e-15),
              ('cbrt', x[:300000], np.cbrt(xl[:300000]), 1e-15), ('inv', x[:300000], 1.0 / xl[:300000], 2e-15)]
    for kind, xin, want, tol in checks:
        got = eng.debug_math(kind, dev(xin)).cpu().numpy().astype(np.longdouble)
        err = float(np.max(np.abs(got - want) / np.abs(want)))
        assert err < tol, (kind, err)
    # log: relative to max(|log x|, tiny) -- near x = 1 the result must still be relatively accurate (f = m - 1 is exact)
    got = eng.debug_math('log', dev(x)).cpu().numpy().astype(np.longdouble)
    want = np.log(xl)
    err = float(np.max(np.abs(got - want) / np.maximum(np.abs(want), 1e-300)))
    assert err < 2e-15, err
    xe = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-1, 1, 100000), rng.uniform(-1e-8, 1e-8, 1000)])
    got = eng.debug_math('exp', dev(xe)).cpu().numpy().astype(np.longdouble)
    want = np.exp(xe.astype(np.longdouble))
    err = float(np.max(np.abs(got - want) / want))
    assert err < 2e-15, err


@pytest.mark.parametrize('shape', [(8, 8, 16), (16, 32, 64), (32, 16, 32), (64, 32, 128), (128, 8, 16), (256, 64, 32), (512, 16, 32),
                                   (256, 48, 16), (512, 8, 48), (1024, 8, 16), (256, 120, 48)])      # (cross-wave kernel: thin / mixed-radix neighbours, 1 024 points)
def test_wave_local_x_pass_matches_the_group_parallel_kernel(shape):
    """the two fused x-pass kernels (xwave.h: a line of every spectrum in one wavefront, mix in registers; fft_kernels.h:
    one wave group per spectrum, mix through LDS) on every x extent and every mix functor of configs 1-3 + GGA kinetic"""
    box = dev(cases.make_cell(('tri', shape[0] / 16.0)))
    den = synth.random_density(shape, seed=61)
    vext = dev(synth.random_potential(shape, seed=62))
    s5 = np.sqrt(5.0)
    term_sets = [(_CFG_TERMS['cfg3'], None), (_CFG_TERMS['cfg2'], None), (['pbe'], None),
                 (['hartree', 'tf', 'vw', 'wt_nl', 'gga_k'], {'wt_alpha': (5 + s5) / 6, 'wt_beta': (5 - s5) / 6})]     # WGC98 exponents + LKT
    for ts, params in term_sets:
        names = F.NativeTerms(ts).names
        out = {}
        for xw in (2, 1, 0, 5, 7):      # wave-local, default (cross-wave at 256 / 512 points), group-parallel, cross-wave wherever it exists, round-3 choice
            for gsplit in (1, 0):
                eng = Engine(shape, DEV).set_cell(box).set_terms(names, params)
                eng.set_option(8, xw).set_option(6, gsplit)
                E, v = eng.energy_potential(dev(den), vext)
                out[(xw, gsplit)] = (E, v.cpu().numpy(), int(eng.query(0)))
                eng.close()
        Er, vr, nf = out[(0, 1)]
        for key, (E, v, n) in out.items():
            for k in Er:
                assert abs(E[k] - Er[k]) <= 1e-12 * max(1.0, abs(Er[k])), (ts, key, k, E[k], Er[k])
            assert relerr(v, vr) < 1e-12, (ts, key)
            assert n == nf


@pytest.mark.parametrize('case', ['g16r', 'g17r', 'g18t'])
def test_stabilised_wang_teter_style_functional_in_one_engine_call(case):
    """WangTeterStyleFunctional with f = exp (functionals.py:728-782) as ONE native evaluation (two combine passes) through
    every pipeline -- g17r / g18t are non power-of-two grids (unfused pipeline) -- against the reference's golden, the
    closure form against the same potential, and a custom f (torch composition) against the native form"""
    gold = load('terms_%s.npz' % case)
    box, den, vext, chi, n_elec = cases.make_inputs(case)
    Eref, vref = float(gold['E_wts_exp']), gold['v_wts_exp']
    eng = Engine(den.shape, DEV).set_cell(dev(box)).set_terms(['tf', 'vw', 'wt_nl'], {'wts_kind': 1.0})
    for mode in (0, 1, 2):
        eng.set_option(0, mode)
        E, v = eng.energy_potential(dev(den))
        assert abs(sum(E.values()) - Eref) <= E_RTOL * abs(Eref), mode
        assert relerr(v.cpu().numpy(), vref) < V_RTOL, mode
        assert E['wt_nl'] == 0.0 and E['tf'] > 0.0
    eng.set_option(0, 0)
    # closure: n = N_e chi^2 / int chi^2; the gradient is 2 c chi (v - mu) dV with THIS functional's potential
    vol = abs(np.linalg.det(box))
    c = n_elec / (np.mean(chi * chi) * vol)
    Ed, vd = eng.energy_potential(dev(c * chi * chi))
    Ec, mu, g = eng.energy_grad_chi(dev(chi), n_elec)
    assert abs(sum(Ec.values()) - sum(Ed.values())) <= 1e-12 * abs(sum(Ed.values()))
    vdn = vd.cpu().numpy()
    mu_ref = float(np.sum(vdn * c * chi * chi) * vol / chi.size / n_elec)
    gref = 2 * c * chi * (vdn - mu_ref) * vol / chi.size
    assert abs(mu - mu_ref) <= 1e-11 * abs(mu_ref) and relerr(g.cpu().numpy(), gref) < 1e-10
    eng.close()
    # the drop-in class: enumerated f -> native single call; anything else -> torch composition of native energies
    f_native = F.WangTeterStyleFunctional((5 / 6, 5 / 6, torch.exp))
    f_custom = F.WangTeterStyleFunctional((5 / 6, 5 / 6, lambda x: torch.exp(x * 1.0000000001) ** 1.0))
    assert f_native._kind == 1.0 and F.WangTeterStyleFunctional()._kind == 0.0
    tb = dev(box)
    for f in (f_native, f_custom):
        E = float(f(tb, dev(den)))
        v = F.get_functional_derivative(tb, dev(den), f).cpu().numpy()
        tol = 1e-8 if f is f_custom else 1.0
        assert abs(E - Eref) <= max(E_RTOL, 1e-9 if f is f_custom else 0) * abs(Eref) * (10 if f is f_custom else 1)
        assert relerr(v, vref) < (1e-8 if f is f_custom else V_RTOL), tol


@pytest.mark.parametrize('shape', [(48, 96, 120), (96, 120, 48), (144, 160, 192), (240, 250, 270), (270, 240, 250), (250, 270, 240),
                                   (288, 48, 320), (320, 384, 96), (480, 96, 144), (96, 48, 480), (64, 240, 256), (256, 128, 250)])
def test_mixed_radix_extents_run_the_fused_pipelines(shape):
    """grids whose extents have factors 3 and 5 (the reference's ecut2shape never returns a power of two): the
    mixed-radix line transforms put them on the z-fused pipeline.  Checked against the chirp-z + unfused path (option 9 off:
    independent transforms AND independent pipeline), with the x-fused-only and unfused pipelines on the same
    transforms in between, energy_potential and closure forms, configs 1-3 + the q-dependent Pauli-Gaussian."""
    box = cases.make_cell(('tri', shape[0] / 24.0))
    den = synth.random_density(shape, seed=71)
    vext = synth.random_potential(shape, seed=72)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(73).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box))) + 0.3)
    eng = Engine(shape, DEV).set_cell(dev(box))
    assert eng.fast_path
    sets = [(F.NativeTerms(names).names, None) for names in _CFG_TERMS.values()]
    sets.append((('hartree', 'vw', 'gga_k', 'pbe_x', 'pbe_c'), {'ggak_kind': 1.0, 'ggak_beta': 0.25, 'ggak_lambda': 0.4, 'ggak_sigma': 0.2}))
    for names, params in sets:
        eng.set_terms(names, params)
        res = {}
        for key, (mixed, mode) in {'ref': (0, 0), 'zf': (1, 0), 'xf': (1, 2), 'un': (1, 1)}.items():
            eng.set_option(9, mixed).set_option(0, mode)
            assert eng.fast_path == bool(mixed)
            E, v = eng.energy_potential(dev(den), dev(vext))
            Ec, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
            res[key] = (E, v.cpu().numpy(), Ec, mu, g.cpu().numpy(), int(eng.query(4)))
        ref = res['ref']
        for key in ('zf', 'xf', 'un'):
            r = res[key]
            for k in ref[0]:
                assert abs(r[0][k] - ref[0][k]) <= 1e-11 * max(1.0, abs(ref[0][k])), (names, key, k, r[0][k], ref[0][k])
                assert abs(r[2][k] - ref[2][k]) <= 1e-11 * max(1.0, abs(ref[2][k])), (names, key, k)
            assert relerr(r[1], ref[1]) < 1e-10, (names, key, relerr(r[1], ref[1]))
            assert relerr(r[4], ref[4]) < 1e-10, (names, key)
            assert abs(r[3] - ref[3]) < 1e-11 * max(1.0, abs(ref[3]))
        assert res['zf'][5] < res['un'][5]               # the fused pipeline really ran (fewer launches)
    eng.set_option(9, 1).set_option(0, 0)
    eng.close()


def test_odd_grid_fused_chirpz_x_pass_matches_the_three_pass_form_and_the_oracle():
    """extents as the reference's System.ecut2shape gives them (system.py:74-89: always odd; here 27 x 35 x 33, triclinic): the
    chirp-z path with forward-x / multiply / inverse-x in one kernel (OFDFT_OPT_BS_FUSED, default) against its three-pass form
    with separate multiply kernels, and config 3 against the pinned CPU oracle"""
    shape = (27, 35, 33)
    box = cases.make_cell(('tri', 1.3))
    den = synth.random_density(shape, seed=91)
    vext = synth.random_potential(shape, seed=92)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(93).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box))) + 0.3)
    eng = Engine(shape, DEV).set_cell(dev(box))
    assert not eng.fast_path
    sets = [(F.NativeTerms(names).names, None) for names in _CFG_TERMS.values()]
    sets.append((('hartree', 'vw', 'wt_nl', 'pbe_x', 'pbe_c'), {'wt_alpha': 0.7, 'wt_beta': 0.9}))
    sets.append((('vw', 'pbe_x'), None))                      # gradient without Hartree
    sets.append((('hartree', 'vw', 'gga_k', 'pbe_x', 'pbe_c'), {'ggak_kind': 1.0, 'ggak_beta': 0.25, 'ggak_lambda': 0.4, 'ggak_sigma': 0.2}))
    for names, params in sets:
        eng.set_terms(names, params)
        res = {}
        for fused in (1, 0):
            eng.set_option(15, fused)
            E, v = eng.energy_potential(dev(den), dev(vext))
            Ec, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
            res[fused] = (E, v.cpu().numpy(), Ec, mu, g.cpu().numpy(), int(eng.query(4)))
        a, b = res[1], res[0]
        for k in b[0]:
            assert abs(a[0][k] - b[0][k]) <= 1e-11 * max(1.0, abs(b[0][k])), (names, k, a[0][k], b[0][k])
            assert abs(a[2][k] - b[2][k]) <= 1e-11 * max(1.0, abs(b[2][k])), (names, k)
        assert relerr(a[1], b[1]) < 1e-10 and relerr(a[4], b[4]) < 1e-10 and abs(a[3] - b[3]) < 1e-11 * max(1.0, abs(b[3])), names
        if 'gga_k' not in names:
            assert a[5] < b[5], (names, a[5], b[5])       # the fused x pass really ran (fewer launches)
    eng.set_option(15, 1)
    ev = cf.Evaluator(cf.Grid(box, shape))
    Eo, go, muo = ev.closure(['ion_electron', 'hartree', 'wgc99', 'pbe_x', 'pbe_c'], chi, n_elec, vext)
    eng.set_terms(F.NativeTerms(_CFG_TERMS['cfg3']).names)
    E, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
    assert abs(sum(E.values()) - Eo) <= E_RTOL * abs(Eo)
    assert abs(mu - muo) <= 1e-9 * max(1.0, abs(muo))
    assert relerr(g.cpu().numpy(), go) < V_RTOL
    eng.close()


def test_mixed_radix_grid_matches_the_oracle():
    """config 3 on a 48 x 96 x 120 triclinic grid against the pinned CPU oracle (closed forms on numpy FFTs)"""
    shape = (48, 96, 120)
    box = cases.make_cell(('tri', 2.0))
    den = synth.random_density(shape, seed=81)
    vext = synth.random_potential(shape, seed=82)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(83).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box))) + 0.3)
    ev = cf.Evaluator(cf.Grid(box, shape))
    Eo, go, muo = ev.closure(['ion_electron', 'hartree', 'wgc99', 'pbe_x', 'pbe_c'], chi, n_elec, vext)
    eng = Engine(shape, DEV).set_cell(dev(box)).set_terms(F.NativeTerms(_CFG_TERMS['cfg3']).names)
    assert eng.fast_path
    E, mu, g = eng.energy_grad_chi(dev(chi), n_elec, dev(vext))
    assert abs(sum(E.values()) - Eo) <= E_RTOL * abs(Eo)
    assert abs(mu - muo) <= 1e-9 * max(1.0, abs(muo))
    assert relerr(g.cpu().numpy(), go) < V_RTOL
    eng.close()


def test_set_cell_sees_a_box_array_that_was_changed_in_place():
    """round-3 advice: Engine.set_cell compares a host array by its bytes, not by object identity -- a geometry / stress loop that
    rescales its box array in place and hands it over again must get the new cell"""
    shape = (16, 16, 16)
    box = np.ascontiguousarray(cases.make_cell(('tri', 1.0)))
    den = dev(synth.random_density(shape, seed=71))
    eng = engine_for(shape, DEV)
    eng.set_cell(box).set_terms(('hartree', 'tf'))
    E1, _ = eng.energy_potential(den)
    box *= 1.1                                                   # the same ndarray object
    eng.set_cell(box)
    E2, _ = eng.energy_potential(den)
    fresh = engine_for(shape, DEV).set_cell(box.copy()).set_terms(('hartree', 'tf'))
    E3, _ = fresh.energy_potential(den)
    assert abs(sum(E2.values()) - sum(E3.values())) <= 1e-13 * abs(sum(E3.values()))
    assert abs(sum(E2.values()) - sum(E1.values())) > 1e-3 * abs(sum(E1.values()))
