"""CPU pin of the fp32 build's Lindhard factor (csrc/pointwise_kernels.h: lindhard_shape(float), round 5): a numpy float32
emulation of the device function -- same constants, same operation order -- against the oracle's fp64 form of
functionals.py:617-628,648 (itself pinned to the reference's outputs by tests/test_oracle_golden.py).  The device function is
compared with the fp64 engine and the goldens by tests/test_gpu_f32.py; this test pins the ALGORITHM where no GPU is needed."""
import numpy as np

from oracle import closed_form as cf

F = np.float32


def lindhard_shape_f32(eta):
    """line-for-line float32 emulation of the device function"""
    eta = eta.astype(F)
    c = [F(0.0)] + [F(1.0) / F(4 * j * j - 1) for j in range(1, 14)]      # c_j = 1 / (4 j^2 - 1), j = 1..13
    e2 = (eta * eta).astype(F)
    with np.errstate(divide='ignore', invalid='ignore', over='ignore'):
        z = np.where(eta < 1, e2, (F(1) / e2).astype(F)).astype(F)
        r = np.full_like(z, c[12])
        v = np.full_like(z, c[13])
        for j in range(11, 0, -1):
            r = (r * z + c[j]).astype(F)
            v = (v * z + c[j + 1]).astype(F)
        s = (z * r).astype(F)
        small = (s * (F(1) / (F(1) - s)).astype(F) - F(3) * e2).astype(F)
        large = (F(-3) * v * (F(1) / r).astype(F) - F(1)).astype(F)
        f = np.where(eta < 1, small, large).astype(F)
        lg = (F(0.69314718055994530942) * np.log2(np.abs((F(1) + eta) * (F(1) / (F(1) - eta)).astype(F)).astype(F)).astype(F)).astype(F)
        g = ((F(1) - e2) * F(0.25) * (F(1) / eta).astype(F) * lg + F(0.5)).astype(F)
        mid = np.where(eta == 1, F(-2), (F(1) / g).astype(F) - F(3) * e2 - F(1)).astype(F)
    return np.where((eta > F(0.6)) & (eta < F(5.0 / 3.0)), mid, f).astype(F)


def reference_shape(eta):
    """1/G^-1 - 3 eta^2 - 1 in extended precision, cancellation-free where the direct form is not (series of the logarithm)"""
    L = np.longdouble
    eta = eta.astype(L)
    out = np.empty_like(eta)
    c = np.array([L(1) / (4 * j * j - 1) for j in range(1, 400)], dtype=L)
    for i, e in enumerate(eta):
        if e == 0:
            out[i] = 0
        elif e < 0.7:
            x = e * e
            S = np.sum(c * x ** np.arange(1, 400, dtype=L))
            out[i] = S / (1 - S) - 3 * x
        elif e > 1.4:
            x = 1 / (e * e)
            G = np.sum(c * x ** np.arange(1, 400, dtype=L))
            out[i] = 1 / G - 3 * e * e - 1
        else:
            g = L(0.5) if e == 1 else L(0.5) + ((1 - e * e) / (4 * e)) * np.log(abs((1 + e) / (1 - e)))
            out[i] = 1 / g - 3 * e * e - 1
    return out.astype(np.float64)


def test_reference_shape_agrees_with_the_oracle_where_the_direct_form_is_well_conditioned():
    eta = np.linspace(0.3, 3.0, 200)
    assert np.max(np.abs(reference_shape(eta) - cf.lindhard_kernel_shape(eta)) / np.abs(cf.lindhard_kernel_shape(eta))) < 1e-12


def test_fp32_series_form_has_no_cancellation():
    rng = np.random.default_rng(0)
    parts = {'small': rng.uniform(1e-4, 0.6, 3000), 'mid': rng.uniform(0.6, 5.0 / 3.0, 4000), 'large': rng.uniform(5.0 / 3.0, 40.0, 3000),
             'near one': 1 + rng.uniform(-1e-3, 1e-3, 500)}
    bars = {'small': (3e-7, 1e-7), 'mid': (4e-6, 6e-7), 'large': (3e-7, 1e-7), 'near one': (2e-6, 6e-7)}
    for name, eta in parts.items():
        eta = eta.astype(F).astype(np.float64)                # the device sees eta in fp32
        got = lindhard_shape_f32(eta).astype(np.float64)
        ref = reference_shape(eta)
        rel = np.abs(got - ref) / np.abs(ref)
        assert rel.max() < bars[name][0] and np.sqrt(np.mean(rel ** 2)) < bars[name][1], (name, rel.max(), np.sqrt(np.mean(rel ** 2)))
    # the special points of functionals.py:625-627
    assert lindhard_shape_f32(np.array([0.0, 1.0]))[0] == 0 and lindhard_shape_f32(np.array([0.0, 1.0]))[1] == -2
    # ... and what the direct form does in fp32 at the ends (why rounds 3-4 evaluated it in fp64)
    eta = np.array([0.01, 12.0], dtype=np.float64)
    e = eta.astype(F)
    direct = (F(1) / (F(0.5) + ((F(1) - e * e) / (F(4) * e)) * np.log(np.abs((F(1) + e) / (F(1) - e))).astype(F)).astype(F) - F(3) * e * e - F(1)).astype(np.float64)
    assert np.all(np.abs(direct - reference_shape(eta)) / np.abs(reference_shape(eta)) > 1e-4)
