"""CPU-side checks: the C-ABI library builds/loads and exports every declared symbol; host logic."""
import ctypes
import os
import re

import pytest
import torch

from professad_amd import _native as N
from professad_amd import functionals as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('dtype', [N.F64, N.F32])
def test_library_exports_every_declared_symbol(dtype):
    """both precisions (libofdft_hip.so, libofdft_hip_f32.so) carry the whole ABI of include/ofdft_hip.h"""
    lib = N.load(dtype)
    header = open(os.path.join(ROOT, 'include', 'ofdft_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(ofdft_[a-z_]+)\s*\(', header)))
    assert declared == sorted(N.EXPORTS)
    for sym in declared:
        assert hasattr(lib, sym), sym


def test_header_constants_match_python_mirror():
    header = open(os.path.join(ROOT, 'include', 'ofdft_hip.h')).read()
    bits = dict(re.findall(r'#define OFDFT_([A-Z0-9_]+)\s+\(1u << (\d+)\)', header))
    for nm, bit in N.TERM_BITS.items():
        assert int(bits[nm.upper()]) == bit.bit_length() - 1
    assert N.TERM_ORDER == sorted(N.TERM_BITS, key=lambda k: N.TERM_BITS[k])
    assert int(re.search(r'#define OFDFT_NTERMS\s+(\d+)', header).group(1)) == N.NTERMS == len(N.TERM_ORDER)


def test_error_paths_without_gpu():
    lib = N.load()
    ctx = ctypes.c_void_p(0)
    # invalid extents are rejected before any device work
    assert lib.ofdft_create(ctypes.byref(ctx), 1, 8, 8, N.F64, 0) == N.EINVAL
    assert b'extents' in lib.ofdft_last_error(None)
    # a context's precision is its library's: each build refuses the other dtype and names the library to use
    assert lib.ofdft_create(ctypes.byref(ctx), 8, 8, 16, N.F32, 0) == N.EINVAL
    assert b'libofdft_hip_f32.so' in lib.ofdft_last_error(None)
    lib32 = N.load(N.F32)
    assert lib32.ofdft_create(ctypes.byref(ctx), 8, 8, 16, N.F64, 0) == N.EINVAL
    assert b'libofdft_hip.so' in lib32.ofdft_last_error(None)
    assert lib32.ofdft_stress(None, None, None, None) == N.EINVAL       # per-geometry-step routines: fp64 library only
    assert lib.ofdft_set_cell(None, None) == N.EINVAL
    if not torch.cuda.is_available():
        rc = lib.ofdft_create(ctypes.byref(ctx), 16, 16, 16, N.F64, 0)
        assert rc == N.EHIP and not ctx.value


@pytest.mark.skipif(torch.cuda.is_available(), reason='checks the no-GPU failure mode')
def test_product_path_fails_loudly_without_gpu():
    box = torch.eye(3, dtype=torch.double) * 5
    den = torch.full((16, 16, 16), 0.03, dtype=torch.double)
    with pytest.raises(N.NativeLibraryError):
        F.Hartree(box, den)


def test_reference_protocol_names():
    assert F.IonElectron.__qualname__ == 'IonElectron' and F.IonElectron.__name__ == 'IonElectron'
    assert F.IonIon.__qualname__ == 'IonIon'
    assert F.WangTeter.__qualname__ == 'WangTeter'
    w = F.WangGovindCarter99()
    assert w.__qualname__ == 'WangGovindCarter99' and (w.alpha, w.gamma) == ((5 + 5 ** 0.5) / 6, 2.7)
    fused = F.NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe'])
    assert fused.__qualname__ == 'IonElectron' and fused.needs_vext
    assert fused.names == ('ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c')
    assert F.NativeTerms(['wt', 'pz']).__qualname__ == 'NativeTerms'
    with pytest.raises(KeyError):
        F.NativeTerms(['nope'])


def test_recpot_reader(tmp_path):
    """The recpot parser on a synthetic file in the reference's format (ion_utils.py:20-81)."""
    import numpy as np
    from professad_amd.ions import BOHR, POT_CONV, read_recpot
    z, n = 3, 30
    kmax_file = 8.0
    ks = np.linspace(0.0, kmax_file * BOHR, n)
    dk = ks[1] - ks[0]
    v = np.empty(n)
    v[1:] = -4 * np.pi * z / ks[1:] ** 2 * np.exp(-ks[1:] ** 2 / 4)
    v[0] = v[1] + 4 * np.pi * z / dk ** 2          # makes (v1 - v0) dk^2 / (-4 pi) = z exactly
    p = tmp_path / 'x.recpot'
    with open(p, 'w') as f:
        f.write('START COMMENT\nsynthetic\nEND COMMENT\n3     5\n%.10f\n' % kmax_file)
        raw = v / POT_CONV
        for i in range(0, n, 3):
            f.write(' '.join('%.16e' % x for x in raw[i:i + 3]) + '\n')
        f.write('1000\n')
    k2, v2, z2 = read_recpot(str(p))
    assert z2 == z and np.allclose(k2, ks, rtol=1e-12)
    assert np.allclose(v2[1:], v[1:] + 4 * np.pi * z / ks[1:] ** 2, rtol=1e-10) and abs(v2[0] - v[0]) < 1e-8 * abs(v[0])


def test_fast_extent_helper_rounds_the_reference_rule_up():
    """ecut2shape_fast: the reference's odd extents (system.py:74-89) rounded up per axis to fused-pipeline extents"""
    import torch
    import numpy as np
    from professad_amd.engine import ecut2shape_fast, fused_extents, next_fast_extent
    assert next_fast_extent(241) == 250 and next_fast_extent(255) == 256 and next_fast_extent(33) == 48
    assert next_fast_extent(241, dtype=torch.float32) == 250 and next_fast_extent(241, nranks=8) == 256 and next_fast_extent(233, nranks=8) == 240      # (fp32: mixed-radix plans since round 3)
    assert next_fast_extent(9, axis=2) == 16 and next_fast_extent(1025, axis=2) == 2048
    with pytest.raises(ValueError):
        next_fast_extent(1025, axis=0)
    assert fused_extents(0)[:4] == [8, 16, 32, 48]
    box = np.diag([4.05, 4.05, 8.1])
    shape, ref = ecut2shape_fast(2000.0, box)
    kcut = np.sqrt(2 * 2000.0 / 27.211386245988)
    want = tuple(int(1 + 2 * np.ceil(kcut / (2 * np.pi / (L / 0.529177210903)))) for L in (4.05, 4.05, 8.1))
    assert ref == want and all(r % 2 == 1 for r in ref)
    assert all(s_ >= r for s_, r in zip(shape, ref)) and shape[2] % 2 == 0
    assert all(s_ in fused_extents(i) for i, s_ in enumerate(shape))


def test_lds_layout_table_of_the_wave_local_x_pass_is_a_conflict_free_permutation():
    """csrc/xwave.h: XwSwz (XOR layout of the line buffers, per length and precision) against the enumeration in
    tools/lds_conflicts.py: the header and the tool carry the same table, every entry is a permutation that stays inside its line
    buffer, and under the bank rules of MI355X_MICROARCH.md every read and write group of every exchange costs one LDS cycle
    (the padded layout it replaces: two) -- the model the 48.8 % -> 1.9 % conflict rate of profiles/r03_sq_counters.md confirmed"""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import lds_conflicts as lc
    src = open(os.path.join(ROOT, 'professad_amd', 'csrc', 'xwave.h')).read()
    # (the b32 table serves the fp32 build only when its exchange moves re / im separately, OFDFT_F32_CX=0; by default the fp32
    # build exchanges whole complex numbers -- 8-byte accesses -- and uses the b64 table like the fp64 build)
    f32_part, f64_part = src.split('#if defined(OFDFT_REAL_F32) && !OFDFT_F32_CX', 1)[1].split('#else', 1)
    f64_part = f64_part.split('#endif', 1)[0]
    pat = re.compile(r'XwSwz<(\d+)> \{ static constexpr int XS = (\d+), XM = (\d+), XMUL = (\d+), LMUL = (\d+), RS = (\d+); \}')
    for prec, part in (('f32', f32_part), ('f64', f64_part)):
        found = {int(m[0]): tuple(int(x) for x in m[1:]) for m in pat.findall(part)}
        assert found == lc.TABLE[prec], (prec, found)
        for LEN, (xs, xm, mul, lmul, RS) in found.items():
            f = lc.xor_layout(xs, xm, mul)
            for line in range(max(1, 64 // (LEN // 8))):
                img = [f(i) ^ ((line * lmul) & 31) for i in range(LEN)]
                assert len(set(img)) == LEN and max(img) < RS, (prec, LEN, line)
            rd, wr = lc.conflicts(LEN, lc.PLANS[LEN], f, RS, prec == 'f32', lmul)
            assert rd == 1.0 and wr == 1.0, (prec, LEN, rd, wr)
            old = lc.conflicts(LEN, lc.PLANS[LEN], lambda i: i + (i >> 4), LEN + (LEN >> 4) + 2, prec == 'f32')
            assert old[0] >= 2.0, (prec, LEN, old)


def test_host_box_cache_follows_the_tensor_and_keeps_nothing_alive():
    """professad_amd.functionals._host_box (round-3 advice): a hit needs the same tensor object, version counter and storage; an
    in-place change misses; tensors that require grad or have no version counter (inference mode) are never cached; only weak
    references are held"""
    import gc
    import weakref

    import numpy as np
    import torch
    from professad_amd import functionals as F
    F._BOX_CACHE.clear()
    box = torch.eye(3, dtype=torch.double) * 7.5
    b1, v1 = F._host_box(box)
    b2, v2 = F._host_box(box)
    assert b2 is b1 and v1 == v2 == 7.5 ** 3 and len(F._BOX_CACHE) == 1
    box *= 2.0                                                   # in place: the version counter moves
    b3, v3 = F._host_box(box)
    assert b3 is not b1 and abs(v3 - 15.0 ** 3) < 1e-9 and np.allclose(b3, np.eye(3) * 15.0)
    g = torch.eye(3, dtype=torch.double, requires_grad=True)
    F._host_box(g)
    assert id(g) not in F._BOX_CACHE                            # stress paths: never cached
    with torch.inference_mode():
        inf = torch.eye(3, dtype=torch.double) * 3.0
    bi, vi = F._host_box(inf)                                    # no version counter: works, uncached
    assert abs(vi - 27.0) < 1e-12
    ref = weakref.ref(box)
    del box
    gc.collect()
    assert ref() is None                                         # the cache did not keep the caller's tensor alive
    for i in range(3 * F._BOX_CACHE_MAX):                        # bounded
        F._host_box(torch.eye(3, dtype=torch.double) * (1.0 + i))
    assert len(F._BOX_CACHE) <= F._BOX_CACHE_MAX
