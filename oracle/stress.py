"""ORACLE (test infrastructure, never shipped or measured as the product).

numpy closed forms of the per-term stress tensors sigma_ij = (1/Omega) dE/d eps_ij with the density scaled as
1/volume -- what the reference's get_stress (functional_tools.py:73-101) and System.__compute_stress
(system.py:925-935) obtain by autograd.  Hartree, Thomas-Fermi, von Weizsaecker, Wang-Teter nonlocal, LDA and PBE
follow the reference's own analytic test forms (tests/tools_for_tests.py:212-307, 367-472); the ion-electron stress is
derived here (the reference has no closed form): with S_k fixed by the fractional coordinates,
    sigma_ij = -(E/Omega) delta_ij - (1/(N Omega)) sum_k Re[S_k conj(n^_k)] v~'(|k|) k_i k_j / |k| .

Parity status: PINNED -- tests/test_oracle_golden.py::test_stress_oracle checks every function against
tests/golden/stress.npz (outputs of the reference's get_stress on seeded inputs).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np

from oracle import ions as oi
from oracle.closed_form import C_TF, Evaluator, Grid, recip
from oracle.ions import half_weights

PI = math.pi
PAIRS = [(0, 0), (1, 1), (2, 2), (0, 1), (0, 2), (1, 2)]


def _sym(c):
    """six components (xx, yy, zz, xy, xz, yz) -> 3x3"""
    s = np.zeros((3, 3))
    for v, (i, j) in zip(c, PAIRS):
        s[i, j] = s[j, i] = v
    return s


def _grad(g, n):
    kx, ky, kz, _ = recip(g.box, g.shape)
    nk = g.fwd(n)
    return [g.inv(1j * k * nk) for k in (kx, ky, kz)]


def hartree(box, n):
    """tools_for_tests.py:212-238"""
    g = Grid(box, n.shape)
    kx, ky, kz, k2 = recip(box, n.shape)
    w = half_weights(n.shape)[None, None, :]
    nk = np.fft.rfftn(n, norm='forward')
    with np.errstate(divide='ignore', invalid='ignore'):
        aux = np.where(k2 != 0, 4 * PI * w * np.abs(nk) ** 2 / k2 ** 2, 0.0)
    ks = (kx, ky, kz)
    E = Evaluator(g).hartree(n)[0]
    return _sym([np.sum(aux * ks[i] * ks[j]) for i, j in PAIRS]) - E / g.vol * np.eye(3)


def tf(box, n):
    """:241-243"""
    g = Grid(box, n.shape)
    return -2 / 3 * Evaluator(g).tf(n)[0] / g.vol * np.eye(3)


def vw(box, n):
    """The reference's analytic test form (:246-259) is -1/4 mean(d_i n d_j n / n); that is the continuum limit of what
    autograd returns.  The discrete functional (functionals.py:227-246) is E = -1/2 mean(s lap s) Omega with the
    spectral Laplacian of s = sqrt(n), i.e. E = (Omega/2) sum_k k^2 |s^_k|^2, whose exact strain derivative is
    sigma_ij = -sum_k k_i k_j |s^_k|^2 (s^ forward-normalised; Omega |s^|^2 is invariant).  That is what is used here
    and in the engine: it equals the reference's get_stress to round-off for ANY density, the analytic test form only
    for smooth ones."""
    kx, ky, kz, _ = recip(box, n.shape)
    w = half_weights(n.shape)[None, None, :]
    sk = np.fft.rfftn(np.sqrt(n), norm='forward')
    a = w * np.abs(sk) ** 2
    ks = (kx, ky, kz)
    return -_sym([np.sum(a * ks[i] * ks[j]) for i, j in PAIRS])


def wt_nl(box, n, alpha=5 / 6, beta=5 / 6):
    """long-range part of :262-307 (without the TF and vW stresses the reference adds for WangTeter)"""
    g = Grid(box, n.shape)
    kx, ky, kz, k2 = recip(box, n.shape)
    T = Evaluator(g).wt_nl(n, alpha, beta)[0]
    n0 = n.mean()
    kf = (3 * PI * PI * n0) ** (1 / 3)
    pref = 0.5 * PI * PI / alpha / beta / n0 ** (alpha + beta - 2) / kf
    w = half_weights(n.shape)[None, None, :]
    a = np.fft.rfftn(n ** alpha, norm='forward')
    b = np.fft.rfftn(n ** beta, norm='forward')
    aux1 = w * (a * np.conj(b)).real
    aux1[0, 0, 0] = 0.0
    with np.errstate(divide='ignore', invalid='ignore'):
        eta = np.sqrt(k2) / (2 * kf)
        lg = np.log(np.abs((1 + eta) / (1 - eta)))
        lind = 0.5 + (1 - eta * eta) / (4 * eta) * lg
        aux3 = eta / lind ** 2 * (0.5 / eta - 0.25 * (1 + 1 / (eta * eta)) * lg) + 6 * eta * eta
        aux3 = np.where(k2 != 0, aux3, 0.0)
        ks = (kx, ky, kz)
        comps = [np.sum(aux1 * aux3 * np.where(k2 != 0, ks[i] * ks[j] / k2 - (1 / 3 if i == j else 0.0), 0.0))
                 for i, j in PAIRS]
    return -2 * T / 3 / g.vol * np.eye(3) + pref * _sym(comps)


def wgc99_nl(box, n, params=None):
    """Nonlocal part of WangGovindCarter99 (functionals.py:941-985); derived here, the reference has no closed form.
    With forward-normalised spectra of A = n^b, B = A th, C = A th^2/2, P = n^a, Q = P th, S = P th^2/2 (th = n - n_ref)
        E = C_TF Omega sum_k [ w0 X0 + K1 X1 + K2 X2 + K3 X3 ],   X0 = Re P*A, X1 = Re(P*B + Q*A), X2 = Re(P*C + S*A),
        X3 = Re Q*B.  Under a strain every term scales as J^(-2/3) at fixed eta (n, n_ref ~ 1/J; T ~ n_ref^(5/3-a-b)), and
        d eta / d eps_ij = eta (delta_ij / 3 - k_i k_j / k^2)   (k_F ~ J^(-1/3); round(N_e) carries no gradient), so
        sigma_ij = -(2/3) E / Omega delta_ij + C_TF sum_k (w0' X0 + K1' X1 + K2' X2 + K3' X3) eta (delta_ij/3 - k_i k_j/k^2)
    with the eta-derivatives of the kernels at fixed n_ref (they need the third derivative of the kernel)."""
    from oracle.closed_form import WGC_ALPHA, WGC_BETA, wgc_kernel
    al, be, ga, ka = params or (WGC_ALPHA, WGC_BETA, 2.7, 1.0)
    g = Grid(box, n.shape)
    kx, ky, kz, k2 = recip(box, n.shape)
    nel = round(float(np.mean(n) * g.vol))
    nref = ka * nel / g.vol
    kf = (3 * PI * PI * nref) ** (1 / 3)
    eta = np.where(k2 != 0, np.sqrt(k2) / (2 * kf), 0.0)
    w0, w1, w2, w3 = wgc_kernel(eta, al, be, ga, third=True)
    T = 20 * nref ** (5 / 3 - al - be)
    w0, w1, w2, w3 = T * w0, T * w1, T * w2, T * w3
    K1 = -eta * w1 / (6 * nref)
    K2 = (eta ** 2 * w2 + (7 - ga) * eta * w1) / (36 * nref ** 2)
    K3 = (eta ** 2 * w2 + (1 + ga) * eta * w1) / (36 * nref ** 2)
    dK1 = -(w1 + eta * w2) / (6 * nref)
    dK2 = (2 * eta * w2 + eta ** 2 * w3 + (7 - ga) * (w1 + eta * w2)) / (36 * nref ** 2)
    dK3 = (2 * eta * w2 + eta ** 2 * w3 + (1 + ga) * (w1 + eta * w2)) / (36 * nref ** 2)
    th = n - nref
    A, P = n ** be, n ** al
    f = lambda a: np.fft.rfftn(a, norm='forward')                                    # noqa: E731
    Ak, Bk, Ck, Pk, Qk, Sk = f(A), f(A * th), f(A * th * th / 2), f(P), f(P * th), f(P * th * th / 2)
    w = half_weights(n.shape)[None, None, :]
    X0 = w * (np.conj(Pk) * Ak).real
    X1 = w * (np.conj(Pk) * Bk + np.conj(Qk) * Ak).real
    X2 = w * (np.conj(Pk) * Ck + np.conj(Sk) * Ak).real
    X3 = w * (np.conj(Qk) * Bk).real
    E = C_TF * g.vol * np.sum(w0 * X0 + K1 * X1 + K2 * X2 + K3 * X3)
    G = C_TF * (w1 * X0 + dK1 * X1 + dK2 * X2 + dK3 * X3) * eta
    ks = (kx, ky, kz)
    with np.errstate(divide='ignore', invalid='ignore'):
        comps = [np.sum(np.where(k2 != 0, G * ((1 / 3 if i == j else 0.0) - ks[i] * ks[j] / k2), 0.0)) for i, j in PAIRS]
    return -2 / 3 * E / g.vol * np.eye(3) + _sym(comps)


def lda(box, n, name):
    """:367-390: (E - int v n) / Omega on the diagonal; name in lda_x, pz_c, pw_c, chachiyo_c"""
    g = Grid(box, n.shape)
    E, v = getattr(Evaluator(g), name)(n)
    return (E - g.integral(v * n)) / g.vol * np.eye(3)


def pbe(box, n, do_x=True, do_c=True):
    """:393-472"""
    g = Grid(box, n.shape)
    d = _grad(g, n)
    gn2 = d[0] ** 2 + d[1] ** 2 + d[2] ** 2
    f, dfdn, dfdg = Evaluator(g).pbe_pointwise(n, gn2, do_x, do_c)
    t2 = _sym([-2 * np.mean(((gn2 if i == j else 0.0) + d[i] * d[j]) * dfdg) for i, j in PAIRS])
    return np.mean(f - n * dfdn) * np.eye(3) + t2


def pauli_gaussian(box, n, mu=40 / 27, beta=0.25, lam=0.0, sigma=0.0):
    """Pauli part of the Pauli-Gaussian family with the q-dependent terms (functionals.py:336-403), derived here (the
    reference has no closed form).  For f(n, g = |grad n|^2, l = lap n) a strain at fixed electron number gives
    n -> n/J, g -> (g - 2 eps_ij d_i n d_j n)/J^2, l -> (l - 2 eps_ij d_i d_j n)/J, hence
        sigma_ij = delta_ij mean(f - n f_n - 2 g f_g - l f_l) - 2 mean(f_g d_i n d_j n) - 2 mean(f_l d_i d_j n)
    with the spectral Hessian d_i d_j n = F^-1[-k_i k_j n^] (the derivative of the reference's -k^2 with respect to the cell)."""
    g = Grid(box, n.shape)
    kx, ky, kz, k2 = recip(box, n.shape)
    nk = g.fwd(n)
    d = [g.inv(1j * k * nk) for k in (kx, ky, kz)]
    gn2 = d[0] ** 2 + d[1] ** 2 + d[2] ** 2
    lap = g.inv(-k2 * nk)
    cs = 0.25 * (3 * PI * PI) ** (-2 / 3)
    s2, q = cs * gn2 / n ** (8 / 3), cs * lap / n ** (5 / 3)
    tau = C_TF * n ** (5 / 3)
    ex = np.exp(-mu * s2)
    Fe = ex + beta * q * q - lam * q * s2 + sigma * s2 * s2
    Fs, Fq = -mu * ex - lam * q + 2 * sigma * s2, 2 * beta * q - lam * s2
    f = tau * Fe
    f_n = (5 / 3) * tau / n * Fe + tau * (Fs * (-(8 / 3) * s2 / n) + Fq * (-(5 / 3) * q / n))
    f_g, f_l = tau * Fs * cs / n ** (8 / 3), tau * Fq * cs / n ** (5 / 3)
    ks = (kx, ky, kz)
    hess = lambda i, j: g.inv(-ks[i] * ks[j] * nk)                                    # noqa: E731
    t2 = _sym([-2 * np.mean(f_g * d[i] * d[j]) - 2 * np.mean(f_l * hess(i, j)) for i, j in PAIRS])
    return np.mean(f - n * f_n - 2 * gn2 * f_g - lap * f_l) * np.eye(3) + t2


def wang_teter_style(box, n, alpha=5 / 6, beta=5 / 6, kind='exp'):
    """tools_for_tests.py:310-364: T = vW + T_TF f(X), X = T_NL / (f'(0) T_TF) -> sigma = sigma_vW + sigma_TF (f - f' X)
    + sigma_NL f'(X) / f'(0); f enumerated as 'linear' (1 + x) or 'exp'"""
    g = Grid(box, n.shape)
    ev = Evaluator(g)
    Et, En = ev.tf(n)[0], ev.wt_nl(n, alpha, beta)[0]
    X = En / Et                                                                       # f'(0) = 1 for both choices
    fx, dfx = (1 + X, 1.0) if kind == 'linear' else (math.exp(X), math.exp(X))
    return vw(box, n) + tf(box, n) * (fx - dfx * X) + wt_nl(box, n, alpha, beta) * dfx


def hermite_derivative(x, y, xs):
    """d/dxs of oracle.ions.hermite_interp (functional_tools.py:292-334)"""
    m = oi.hermite_slopes(x, y)
    idx = np.searchsorted(x[1:], xs)
    dx = x[idx + 1] - x[idx]
    t = (xs - x[idx]) / dx
    return ((6 * t * t - 6 * t) * y[idx] + (3 * t * t - 4 * t + 1) * m[idx] * dx + (-6 * t * t + 6 * t) * y[idx + 1]
            + (3 * t * t - 2 * t) * m[idx + 1] * dx) / dx


def ion_electron(box, n, frac, raw, k_max, order=None):
    """derived here (module docstring); S from the exact or the PME structure factor at fixed fractional coordinates"""
    shape = n.shape
    kx, ky, kz, k2 = recip(box, shape)
    kabs = np.sqrt(k2)
    vol = abs(np.linalg.det(box))
    N = np.prod(shape)
    ks, y, z = oi.recpot_table(raw, k_max)
    vk = oi.recpot_on_grid(raw, k_max, kabs)
    with np.errstate(divide='ignore', invalid='ignore'):
        dv = np.where(kabs < ks[-1], hermite_derivative(ks, y, np.minimum(kabs, ks[-1])), 0.0)
        dv = np.where(kabs != 0, dv + 8 * PI * z / kabs ** 3, 0.0)
    S = oi.structure_factor_exact(box, shape, frac) if order is None else oi.structure_factor_pme(shape, frac, order)
    nk = np.fft.rfftn(n)
    w = half_weights(shape)[None, None, :]
    core = w * (S * np.conj(nk)).real
    E = np.sum(core * vk) / N
    kk = (kx, ky, kz)
    with np.errstate(divide='ignore', invalid='ignore'):
        comps = [np.sum(np.where(kabs != 0, core * dv * kk[i] * kk[j] / kabs, 0.0)) / N for i, j in PAIRS]
    return (-E * np.eye(3) - _sym(comps)) / vol
