"""ORACLE (test infrastructure, never shipped or measured as the product).

numpy restatement of the reference's ionic-potential path: recpot table post-processing and
cubic-Hermite interpolation (ion_utils.py:49-81, functional_tools.py:292-334), exact structure factor (:121-137),
particle-mesh Ewald structure factor with cardinal B-splines (:140-286) and the lattice sum (:88-118).

Parity status: PINNED -- tests/test_oracle_golden.py::test_ionic_potential_oracle checks every function against
tests/golden/ions.npz (outputs of the reference itself run on its own al.gga.recpot).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np

from oracle.closed_form import recip


def recpot_table(raw, k_max):
    """(ks, y, z): the table interpolate_recpot builds -- Coulomb tail 4 pi z / k^2 added for k > 0 (:62-73)."""
    ks, dk = np.linspace(0.0, k_max, raw.size, retstep=True)
    z = round((raw[1] - raw[0]) * dk * dk / (-4 * math.pi))
    y = raw.copy()
    y[1:] += 4 * math.pi * z / (ks[1:] * ks[1:])
    return ks, y, z


def hermite_slopes(x, y):
    """functional_tools.py:309-310"""
    m = (y[1:] - y[:-1]) / (x[1:] - x[:-1])
    return np.concatenate([m[:1], 0.5 * (m[1:] + m[:-1]), m[-1:]])


def hermite_interp(x, y, xs):
    """functional_tools.py:292-334 (searchsorted on x[1:], side left)"""
    m = hermite_slopes(x, y)
    idx = np.searchsorted(x[1:], xs)
    dx = x[idx + 1] - x[idx]
    t = (xs - x[idx]) / dx
    h00 = 1 - 3 * t ** 2 + 2 * t ** 3
    h10 = t - 2 * t ** 2 + t ** 3
    h01 = 3 * t ** 2 - 2 * t ** 3
    h11 = -t ** 2 + t ** 3
    return h00 * y[idx] + h10 * m[idx] * dx + h01 * y[idx + 1] + h11 * m[idx + 1] * dx


def recpot_on_grid(raw, k_max, kabs):
    """ion_utils.py:49-81 on |k| values"""
    ks, y, z = recpot_table(raw, k_max)
    v = hermite_interp(ks, y, np.minimum(kabs, ks[-1]))
    with np.errstate(divide='ignore'):
        return np.where(kabs != 0, v - 4 * math.pi * z / np.where(kabs != 0, kabs, 1.0) ** 2, v)


def cardinal_b_spline(x, order):
    """[M_n(x+i), i = 0..n-1] for 0 <= x < 1 (ion_utils.py:140-204; the in-place recursion of its docstring)."""
    x = np.asarray(x, dtype=np.float64)
    M = np.zeros((order,) + x.shape)
    M[0] = x
    M[1] = 1 - x
    for n in range(3, order + 1):
        for i in range(n - 1, 0, -1):
            M[i] = ((x + i) * M[i] + (n - x - i) * M[i - 1]) / (n - 1)
        M[0] = x / (n - 1) * M[0]
    return M


def spline_b(N, count, order):
    """exponential_spline_b for m = 0..count-1 (ion_utils.py:207-215)"""
    m = np.arange(count, dtype=np.float64)
    M = cardinal_b_spline(np.zeros(count), order)
    i = np.arange(order, dtype=np.float64)[:, None]
    b = np.sum(M * np.exp(1j * 2 * math.pi * m * (i - 1) / N), axis=0)
    return np.exp(1j * 2 * math.pi * m * (order - 1) / N) / b


def structure_factor_exact(box, shape, frac):
    kx, ky, kz, _ = recip(box, shape)
    cart = frac @ box
    kr = kx[..., None] * cart[:, 0] + ky[..., None] * cart[:, 1] + kz[..., None] * cart[:, 2]
    return np.exp(-1j * kr).sum(-1)


def structure_factor_pme(shape, frac, order):
    """ion_utils.py:218-286"""
    N = np.array(shape)
    f = frac - np.floor(frac)
    f = f - np.floor(f)
    u = f * N
    fl = np.floor(u).astype(np.int64)
    Q = np.zeros(shape)
    for a in range(frac.shape[0]):
        Ms = [cardinal_b_spline(u[a, d] - fl[a, d], order) for d in range(3)]
        ls = [np.mod(np.arange(order) - fl[a, d], N[d]) for d in range(3)]
        Q[np.ix_(ls[0], ls[1], ls[2])] += Ms[0][:, None, None] * Ms[1][None, :, None] * Ms[2][None, None, :]
    Qk = np.fft.rfftn(Q)
    b0, b1, b2 = spline_b(N[0], Qk.shape[0], order), spline_b(N[1], Qk.shape[1], order), spline_b(N[2], Qk.shape[2], order)
    return np.conj(b0[:, None, None] * b1[None, :, None] * b2[None, None, :] * Qk)


def ionic_potential(box, shape, frac, raw, k_max, order=None):
    """lattice_sum with the recpot (ion_utils.py:88-118; system.py:183-194 for one species)"""
    kx, ky, kz, k2 = recip(box, shape)
    vk = recpot_on_grid(raw, k_max, np.sqrt(k2))
    S = structure_factor_exact(box, shape, frac) if order is None else structure_factor_pme(shape, frac, order)
    vol = abs(np.linalg.det(box))
    return np.fft.irfftn(S * vk, s=shape, axes=(0, 1, 2), norm='forward') / vol


def half_weights(shape):
    """multiplicity of each rfftn half-spectrum coefficient in a full-spectrum sum of a real field's products"""
    nzc = shape[2] // 2 + 1
    w = np.full(nzc, 2.0)
    w[0] = 1.0
    if shape[2] % 2 == 0:
        w[-1] = 1.0
    return w


def ion_electron_forces(box, shape, frac, den, raw, k_max, order=None):
    """F_a = -d/dR_a [dV sum_r n v_ext]: what autograd through lattice_sum + IonElectron yields in
    System.__compute_forces (system.py:913-923), restated analytically.  -> [n_ion, 3] Ha/bohr"""
    kx, ky, kz, k2 = recip(box, shape)
    vk = recpot_on_grid(raw, k_max, np.sqrt(k2))
    vol = abs(np.linalg.det(box))
    dV = vol / np.prod(shape)
    nk = np.fft.rfftn(den)
    F = np.zeros((frac.shape[0], 3))
    if order is None:
        w = half_weights(shape)[None, None, :]
        cart = frac @ box
        for a in range(frac.shape[0]):
            ph = np.exp(-1j * (kx * cart[a, 0] + ky * cart[a, 1] + kz * cart[a, 2]))
            core = w * vk * (1j * ph * np.conj(nk))
            for j, kj in enumerate((kx, ky, kz)):
                F[a, j] = dV / vol * np.sum((kj * core).real)
        return F
    N = np.array(shape)
    b = (spline_b(N[0], nk.shape[0], order)[:, None, None] * spline_b(N[1], nk.shape[1], order)[None, :, None]
         * spline_b(N[2], nk.shape[2], order)[None, None, :])
    theta = np.fft.irfftn(vk * np.conj(b * nk), s=shape, axes=(0, 1, 2), norm='forward') / vol
    f = frac - np.floor(frac)
    f = f - np.floor(f)
    u = f * N
    fl = np.floor(u).astype(np.int64)
    inv = np.linalg.inv(box)
    for a in range(frac.shape[0]):
        Ms, Ds, ls = [], [], []
        for d in range(3):
            x = u[a, d] - fl[a, d]
            M = cardinal_b_spline(x, order)
            P = cardinal_b_spline(x, order - 1) if order > 2 else np.array([1.0])
            P = np.concatenate([[0.0], P, [0.0]])
            Ms.append(M)
            Ds.append(P[1:] - P[:-1])          # d/dx M_n(x+i) = M_{n-1}(x+i) - M_{n-1}(x+i-1)
            ls.append(np.mod(np.arange(order) - fl[a, d], N[d]))
        th = theta[np.ix_(ls[0], ls[1], ls[2])]
        G = np.array([np.einsum('ijk,i,j,k->', th, Ds[0], Ms[1], Ms[2]),
                      np.einsum('ijk,i,j,k->', th, Ms[0], Ds[1], Ms[2]),
                      np.einsum('ijk,i,j,k->', th, Ms[0], Ms[1], Ds[2])])
        F[a] = -dV * inv @ (N * G)
    return F
