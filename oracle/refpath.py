"""ORACLE (test infrastructure, never shipped or measured as the product).

Op-for-op CPU restatement, in torch fp64 + autograd, of the reference's energy path for the
hot-path terms of SURVEY.md §8a.  It performs the *same sequence* of torch.fft.rfftn/irfftn and
pointwise ops as profess-ad's Python (so it is a faithful CPU baseline for bench.py's
`cpu_baseline` leg, kind "port") and obtains dE/dn by autograd exactly as the reference does.

Parity status: PINNED -- tests/test_oracle_golden.py checks every function here against the
fixtures under tests/golden/ that tests/golden/make_golden.py produced by running the reference
itself (imported read-only in the build container).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Reference citations are `path:line` under /root/reference.
"""
import math

import numpy as np
import torch

PI = math.pi
C_TF = 0.3 * (3 * PI * PI) ** (2 / 3)          # functionals.py:223


def _vol(box):
    return torch.abs(torch.linalg.det(box))


def recip_grid(box, shape):
    """kx, ky, kz, k^2 on the rfftn half grid (functional_tools.py:135-162).

    Integer frequencies on axes 0/1 are fftfreq*N with the Nyquist entry made positive
    (:152-154); axis 2 uses rfftfreq (:155)."""
    recip = 2 * PI * torch.linalg.inv(box.T)
    freqs = []
    for ax in (0, 1):
        n = shape[ax]
        f = torch.fft.fftfreq(n, dtype=torch.double, device=box.device) * n
        f[n // 2] = torch.abs(f[n // 2])
        freqs.append(f)
    freqs.append(torch.fft.rfftfreq(shape[2], dtype=torch.double, device=box.device) * shape[2])
    ia, ib, ic = torch.meshgrid(*freqs, indexing='ij')
    comps = [ia * recip[0, c] + ib * recip[1, c] + ic * recip[2, c] for c in range(3)]
    ksq = comps[0].pow(2) + comps[1].pow(2) + comps[2].pow(2)
    return comps[0], comps[1], comps[2], ksq


def d_dxi(ki, f):
    """Spectral partial derivative (functional_tools.py:166-183)."""
    return torch.fft.irfftn(1j * ki * torch.fft.rfftn(f), f.shape)


def grad_sq(kx, ky, kz, f):
    """|grad f|^2 with three independent rfftn/irfftn pairs (functional_tools.py:186-206)."""
    gx, gy, gz = d_dxi(kx, f), d_dxi(ky, f), d_dxi(kz, f)
    return gx * gx + gy * gy + gz * gz


def lap(ksq, f):
    """Spectral Laplacian (functional_tools.py:209-227)."""
    return torch.fft.irfftn(-ksq * torch.fft.rfftn(f), f.shape)


# ----------------------------------------------------------------------------- electrostatics
def ion_electron(box, den, vext):
    """functionals.py:31-46"""
    return torch.mean(den * vext) * _vol(box)


def hartree(box, den):
    """functionals.py:49-72 (k=0 zeroed :67-70)."""
    nk = torch.fft.rfftn(den)
    _, _, _, ksq = recip_grid(box, den.shape)
    green = torch.zeros(ksq.shape, dtype=torch.double, device=den.device)
    green[ksq != 0] = 4 * PI / ksq[ksq != 0]
    vh = torch.fft.irfftn(nk * green, den.shape)
    return 0.5 * torch.mean(den * vh) * _vol(box)


# ----------------------------------------------------------------------------- local / semilocal KEDF
def thomas_fermi(box, den):
    """functionals.py:207-224"""
    return torch.mean(C_TF * den.pow(5 / 3)) * _vol(box)


def weizsaecker(box, den):
    """functionals.py:227-246 (incl. the dead 0.25*lap(n) term and the den!=0 guard :242-243)."""
    rootn = torch.zeros(den.shape, dtype=torch.double, device=den.device)
    rootn[den != 0] = torch.sqrt(den[den != 0])
    _, _, _, ksq = recip_grid(box, den.shape)
    ked = 0.25 * lap(ksq, den) - 0.5 * rootn * lap(ksq, rootn)
    return torch.mean(ked) * _vol(box)


# ----------------------------------------------------------------------------- Lindhard / WT family
def lindhard_inverse(eta):
    """G^-1(eta) with the eta=0 -> 1 and eta=1 -> 1/2 limits patched in (functionals.py:617-628)."""
    raw = 0.5 + ((1 - eta.pow(2)) / (4 * eta)) * torch.log(torch.abs((1 + eta) / (1 - eta)))
    out = torch.empty(eta.shape, dtype=torch.double, device=eta.device)
    out[eta == 0.0] = 1.0
    out[eta == 1.0] = 0.5
    regular = (eta != 0.0) & (eta != 1.0)
    out[regular] = raw[regular]
    return out


def lindhard_on_grid(box, den):
    """eta grid and G^-1 for the *un-rounded* electron count (functionals.py:631-639)."""
    _, _, _, ksq = recip_grid(box, den.shape)
    nel = (torch.mean(den) * _vol(box)).item()
    nbar = nel / _vol(box)
    kf = (3 * PI * PI * nbar).pow(1 / 3)
    eta = torch.zeros(ksq.shape, dtype=torch.double, device=den.device)
    eta[ksq != 0] = torch.sqrt(ksq[ksq != 0]) / (2 * kf)
    return eta, lindhard_inverse(eta)


def wt_nonlocal(box, den, alpha, beta):
    """Density-independent-kernel nonlocal term (functionals.py:644-652)."""
    vol = _vol(box)
    nel = (torch.mean(den) * vol).item()
    nbar = nel / vol
    eta, ginv = lindhard_on_grid(box, den)
    kern = 5 / (9 * alpha * beta * nbar.pow(alpha + beta - 5 / 3)) * (1 / ginv - 3 * eta.pow(2) - 1)
    conv = torch.fft.irfftn(kern * torch.fft.rfftn(den.pow(beta) - nbar.pow(beta)), den.shape)
    return C_TF * torch.mean((den.pow(alpha) - nbar.pow(alpha)) * conv) * vol


def _wt_family(alpha, beta):
    def f(box, den):
        return weizsaecker(box, den) + thomas_fermi(box, den) + wt_nonlocal(box, den, alpha, beta)
    return f


wang_teter = _wt_family(5 / 6, 5 / 6)                                    # functionals.py:655-670
perrot = _wt_family(1, 1)                                                # :673-689
smargiassi_madden = _wt_family(0.5, 0.5)                                 # :692-707
wgc98 = _wt_family((5 + np.sqrt(5)) / 6, (5 - np.sqrt(5)) / 6)           # :710-725


# ----------------------------------------------------------------------------- WGC99
class Wgc99:
    """Density-dependent-kernel WGC99, 2nd-order Taylor form (functionals.py:787-985).

    The kernel series (generate_kernel :845-939) is restated with the same truncation
    (100 terms) and the same eta-cache rule (:961-966)."""

    def __init__(self, alpha=(5 + np.sqrt(5)) / 6, beta=(5 - np.sqrt(5)) / 6, gamma=2.7, kappa=1.0,
                 num_terms=100):
        self.alpha, self.beta, self.gamma, self.kappa = float(alpha), float(beta), float(gamma), float(kappa)
        self.num_terms = num_terms
        self.eta = None
        self.kernel = None

    # -- series coefficients (functionals.py:817-843)
    @staticmethod
    def coeff_a(nt):
        a = np.zeros(nt + 1)
        for idx in range(nt + 1):
            i = idx - 1
            if i == -1:
                a[idx] = 3.0
            else:
                for j in range(-1, i):
                    a[idx] += -3.0 * a[j + 1] / (4 * (i - j + 1) ** 2 - 1)
        out = np.empty(nt)
        out[0] = a[1] - 1.0
        out[1:] = a[2:]
        return out

    @staticmethod
    def coeff_b(nt):
        b = np.zeros(nt)
        for i in range(nt):
            if i == 0:
                b[i] = 1.0
            else:
                for j in range(i):
                    b[i] += b[j] / (4 * (i - j) ** 2 - 1)
        out = np.empty(nt)
        out[0] = 0.0
        out[1] = b[1] - 3.0
        out[2:] = b[2:]
        return out

    def build_kernel(self, eta):
        """w, w', w'' on the eta grid (functionals.py:845-939)."""
        nt = self.num_terms
        dev = eta.device
        u = 3 * (self.alpha + self.beta) - self.gamma / 2
        v = u * u - 36 * self.alpha * self.beta
        A = torch.as_tensor(self.coeff_a(nt), dtype=torch.double, device=dev)
        B = torch.as_tensor(self.coeff_b(nt), dtype=torch.double, device=dev)
        i = torch.arange(nt, dtype=torch.double, device=dev)
        da = (u + 2 * i).pow(2) - v
        db = (u - 2 * i).pow(2) - v
        Sd = torch.sum(A / da - B / db)
        Ss = -2 * torch.sum(i * (A / da + B / db))
        sgn = float(np.sign(u))
        if v > 0:
            rv = math.sqrt(v)
            c1 = sgn * ((rv - u) * Sd + Ss)
            c2 = sgn * ((rv + u) * Sd - Ss) / (2 * rv)
        elif v == 0:
            c1 = sgn * Sd
            c2 = sgn * (Ss - u * Sd)
        else:
            c1 = sgn * Sd
            c2 = sgn * (Ss - u * Sd) / math.sqrt(-v)
        inner = eta <= 1
        C1 = torch.empty(eta.shape, dtype=torch.double, device=dev)
        C2 = torch.empty(eta.shape, dtype=torch.double, device=dev)
        if u >= 0:
            C1[inner], C1[~inner] = c1, 0
            C2[inner], C2[~inner] = c2, 0
        else:
            C1[inner], C1[~inner] = 0, c1
            C2[inner], C2[~inner] = 0, c2
        H = [torch.zeros(eta.shape, dtype=torch.double, device=dev) for _ in range(3)]
        nz = eta != 0
        e, a1, a2 = eta[nz], C1[nz], C2[nz]
        if v > 0:
            x, y = u + math.sqrt(v), u - math.sqrt(v)
            H[0][nz] = a1 * e.pow(x) + a2 * e.pow(y)
            H[1][nz] = a1 * x * e.pow(x - 1) + a2 * y * e.pow(y - 1)
            H[2][nz] = a1 * x * (x - 1) * e.pow(x - 2) + a2 * y * (y - 1) * e.pow(y - 2)
        elif v == 0:
            le = torch.log(e)
            H[0][nz] = e.pow(u) * (a2 * le + a1)
            H[1][nz] = a2 * e.pow(u - 1) * (1 + u * le) + a1 * u * e.pow(u - 1)
            H[2][nz] = a2 * ((u - 1) * e.pow(u - 2) * (1 + u * le) + e.pow(u - 2)) + a1 * u * (u - 1) * e.pow(u - 2)
        else:
            rv = math.sqrt(-v)
            le = torch.log(e)
            tc, ts = torch.cos(rv * le), torch.sin(rv * le)
            H[0][nz] = e.pow(u) * (a1 * tc + a2 * ts)
            H[1][nz] = e.pow(u - 1) * (a1 * (u * tc - rv * ts) + a2 * (u * ts + rv * tc))
            H[2][nz] = (u - 1) * e.pow(u - 2) * a1 * (u * tc - rv * ts) \
                - rv * e.pow(u - 2) * a1 * (u * ts + rv * tc) \
                + (u - 1) * e.pow(u - 2) * a2 * (u * ts + rv * tc) \
                + rv * e.pow(u - 2) * a2 * (u * tc - rv * ts)
        P = [torch.zeros(eta.shape, dtype=torch.double, device=dev) for _ in range(3)]
        m_in = inner & nz
        ein = eta[m_in].unsqueeze(-1)
        cb = B / db
        P[0][m_in] = torch.sum(cb * ein.pow(2 * i), axis=-1)
        P[1][m_in] = torch.sum(cb * (2 * i) * ein.pow(2 * i - 1), axis=-1)
        P[2][m_in] = torch.sum(cb * (2 * i) * (2 * i - 1) * ein.pow(2 * i - 2), axis=-1)
        eout = eta[~inner].unsqueeze(-1)
        ca = A / da
        P[0][~inner] = torch.sum(ca / eout.pow(2 * i), axis=-1)
        P[1][~inner] = torch.sum(ca * (-2 * i) / eout.pow(2 * i + 1), axis=-1)
        P[2][~inner] = torch.sum(ca * (2 * i) * (2 * i + 1) / eout.pow(2 * i + 2), axis=-1)
        return torch.stack([H[0] + P[0], H[1] + P[1], H[2] + P[2]])

    def __call__(self, box, den):
        """functionals.py:941-985 (N_elec *rounded* :952; six convolutions :976-981)."""
        vol = _vol(box)
        _, _, _, ksq = recip_grid(box, den.shape)
        nel = round((torch.mean(den) * vol).detach().item())
        nref = self.kappa * (nel / vol)
        kf = (3 * PI * PI * nref).pow(1 / 3)
        eta = torch.zeros(ksq.shape, dtype=torch.double, device=den.device)
        eta[ksq != 0] = torch.sqrt(ksq[ksq != 0]) / (2 * kf)
        if self.kernel is None or not torch.equal(self.eta, eta):
            self.eta = eta
            self.kernel = self.build_kernel(eta)
        pref = 20 * nref.pow(5 / 3 - self.alpha - self.beta)
        w0, w1, w2 = pref * self.kernel
        K1 = -eta * w1 / (6 * nref)
        K2 = (eta.pow(2) * w2 + (7 - self.gamma) * eta * w1) / (36 * nref.pow(2))
        K3 = (eta.pow(2) * w2 + (1 + self.gamma) * eta * w1) / (36 * nref.pow(2))
        th = den - nref
        rf, irf, b = torch.fft.rfftn, torch.fft.irfftn, self.beta
        conv = irf(w0 * rf(den.pow(b)), den.shape) \
            + th * irf(K1 * rf(den.pow(b)), den.shape) \
            + irf(K1 * rf(den.pow(b) * th), den.shape) \
            + th.pow(2) / 2 * irf(K2 * rf(den.pow(b)), den.shape) \
            + irf(K2 * rf(den.pow(b) * th.pow(2) / 2), den.shape) \
            + th * irf(K3 * rf(den.pow(b) * th), den.shape)
        t_nl = C_TF * torch.mean(den.pow(self.alpha) * conv) * vol
        return weizsaecker(box, den) + thomas_fermi(box, den) + t_nl


# ----------------------------------------------------------------------------- XC
def lda_exchange(box, den):
    """functionals.py:1510-1512"""
    return -(3 / 4) * (3 / PI) ** (1 / 3) * torch.mean(den.pow(4 / 3)) * _vol(box)


def _rs(den):
    return (3 / 4 / PI / den).pow(1 / 3)


def pz_correlation(box, den):
    """functionals.py:1515-1521"""
    g, b1, b2 = -0.1423, 1.0529, 0.3334
    A, B, C, D = 0.0311, -0.048, 0.002, -0.0116
    rs = _rs(den)
    eps = torch.where(rs < 1, A * torch.log(rs) + B + C * rs * torch.log(rs) + D * rs,
                      g / (1 + b1 * torch.sqrt(rs) + b2 * rs))
    return torch.mean(eps * den) * _vol(box)


def _pw92_eps(rs):
    A, a1 = 0.0310907, 0.2137
    b1, b2, b3, b4 = 7.5957, 3.5876, 1.6382, 0.49294
    return -2 * A * (1 + a1 * rs) * torch.log(1 + 1 / (2 * A * (b1 * rs.pow(0.5) + b2 * rs
                                                                 + b3 * rs.pow(1.5) + b4 * rs.pow(2))))


def pw_correlation(box, den):
    """functionals.py:1524-1530"""
    return torch.mean(_pw92_eps(_rs(den)) * den) * _vol(box)


def chachiyo_correlation(box, den):
    """functionals.py:1533-1537"""
    a, b = (np.log(2) - 1) / 2 / PI / PI, 20.4562557
    rs = _rs(den)
    return torch.mean(a * torch.log(1 + b / rs + b / rs.pow(2)) * den) * _vol(box)


def pbe_exchange(box, den):
    """functionals.py:1597-1603 (s^2 via functional_tools.py:252-268)."""
    kx, ky, kz, _ = recip_grid(box, den.shape)
    ex_lda = -(3 / 4) * (3 / PI) ** (1 / 3) * den.pow(4 / 3)
    s2 = 0.25 * (3 * PI * PI) ** (-2 / 3) * grad_sq(kx, ky, kz, den) / den.pow(8 / 3)
    kappa, mu = 0.804, 0.066725 * PI * PI / 3
    fx = 1 + kappa - kappa / (1 + mu / kappa * s2)
    return torch.mean(fx * ex_lda) * _vol(box)


def pbe_correlation(box, den):
    """functionals.py:1606-1618 (PW92 eps_c; +1e-30 guards :1614-1615)."""
    kx, ky, kz, _ = recip_grid(box, den.shape)
    eps = _pw92_eps(_rs(den))
    beta, gam = 0.066725, (1 - np.log(2)) / PI / PI
    A = beta / gam / (torch.exp(-eps / gam) - 1 + 1e-30)
    t2 = (1 / 16) * (PI / 3) ** (1 / 3) * grad_sq(kx, ky, kz, den) / (den.pow(7 / 3) + 1e-30)
    At2 = A * t2
    H = gam * torch.log(1 + beta / gam * t2 * ((1 + At2) / (1 + At2 + At2.pow(2))))
    return torch.mean((eps + H) * den) * _vol(box)


def pz_lda(box, den):
    return lda_exchange(box, den) + pz_correlation(box, den)          # functionals.py:1540


def pbe(box, den):
    return pbe_exchange(box, den) + pbe_correlation(box, den)         # functionals.py:1621-1635


# ----------------------------------------------------------------------------- drivers
def lkt(box, den):
    """functionals.py:309-333"""
    kx, ky, kz, _ = recip_grid(box, den.shape)
    gdg = grad_sq(kx, ky, kz, den)
    ag = torch.zeros(den.shape, dtype=torch.double, device=den.device)
    ag[gdg != 0] = torch.sqrt(gdg[gdg != 0])                          # functional_tools.py:246-249
    s = 0.5 * (3 * PI * PI) ** (-1 / 3) * ag / den.pow(4 / 3)
    tf_ked = 0.3 * (3 * PI * PI) ** (2 / 3) * den.pow(5 / 3)
    return weizsaecker(box, den) + torch.mean(tf_ked / torch.cosh(1.3 * s.clamp(max=100))) * _vol(box)


def pauli_gaussian(mu):
    """functionals.py:336-403 with beta = lambda = sigma = 0"""
    def f(box, den):
        kx, ky, kz, _ = recip_grid(box, den.shape)
        s2 = 0.25 * (3 * PI * PI) ** (-2 / 3) * grad_sq(kx, ky, kz, den) / den.pow(8 / 3)
        tf_ked = 0.3 * (3 * PI * PI) ** (2 / 3) * den.pow(5 / 3)
        return weizsaecker(box, den) + torch.mean(tf_ked * torch.exp(-abs(mu) * s2)) * _vol(box)
    return f


def pauli_gaussian_full(mu, beta, lamb, sigma):
    """functionals.py:336-403 (reduced Laplacian functional_tools.py:271-287)"""
    def f(box, den):
        kx, ky, kz, ksq = recip_grid(box, den.shape)
        c = 0.25 * (3 * PI * PI) ** (-2 / 3)
        s2 = c * grad_sq(kx, ky, kz, den) / den.pow(8 / 3)
        q = c * lap(ksq, den) / den.pow(5 / 3)
        tf_ked = 0.3 * (3 * PI * PI) ** (2 / 3) * den.pow(5 / 3)
        F = torch.exp(-abs(mu) * s2) + abs(beta) * q.pow(2) - abs(lamb) * q * s2 + abs(sigma) * s2.pow(2)
        return weizsaecker(box, den) + torch.mean(tf_ked * F) * _vol(box)
    return f


def vwgtf(kind):
    """functionals.py:251-306"""
    def f(box, den):
        vol = _vol(box)
        n0 = round((torch.mean(den) * vol).detach().item()) / vol
        d = den / n0
        if kind == 1:
            G = 0.9892 * d.pow(-1.2994)
        else:
            G = torch.sqrt(1 / (0.5 * (1 + torch.tanh(5.7001 * d.pow(0.2563) - 5.7001))) - 1)
        return weizsaecker(box, den) + torch.mean(G * 0.3 * (3 * PI * PI) ** (2 / 3) * den ** (5 / 3)) * vol
    return f


def wt_style_exp(box, den):
    """functionals.py:728-782 with (alpha, beta, f) = (5/6, 5/6, exp)"""
    tf = thomas_fermi(box, den)
    return weizsaecker(box, den) + tf * torch.exp(wt_nonlocal(box, den, 5 / 6, 5 / 6) / tf)


def term_table(vext=None):
    """name -> callable(box, den), same keys as tests/golden/cases.py."""
    return {
        'ion_electron': (lambda b, d: ion_electron(b, d, vext)),
        'hartree': hartree, 'tf': thomas_fermi, 'vw': weizsaecker,
        'wt_nl': (lambda b, d: wt_nonlocal(b, d, 5 / 6, 5 / 6)),
        'wt': wang_teter, 'perrot': perrot, 'sm': smargiassi_madden, 'wgc98': wgc98,
        'wgc99': Wgc99(),
        'lda_x': lda_exchange, 'pz_c': pz_correlation, 'pw_c': pw_correlation,
        'chachiyo_c': chachiyo_correlation, 'pbe_x': pbe_exchange, 'pbe_c': pbe_correlation,
        'lkt': lkt, 'pg1': pauli_gaussian(1.0), 'pgs': pauli_gaussian(40 / 27), 'wts_exp': wt_style_exp,
        'pgsl025': pauli_gaussian_full(40 / 27, 0.25, 0.0, 0.0), 'pgslr': pauli_gaussian_full(40 / 27, 0.25, 0.4, 0.2),
        'vwgtf1': vwgtf(1), 'vwgtf2': vwgtf(2),
    }


def energy_and_potential(box, den, f):
    """E and dE/dn by autograd, scaled as functional_tools.py:9-31."""
    den = den.clone().requires_grad_()
    E = f(box, den)
    (g,) = torch.autograd.grad(E, den)
    return E.detach(), g / (_vol(box) / den.numel())


def closure(box, chi, n_elec, fns):
    """The optimize_density closure (system.py:830-838): chi -> E, chi.grad."""
    chi = chi.clone().requires_grad_()
    ntilde = torch.mean(chi.pow(2)) * _vol(box)
    den = (n_elec / ntilde) * chi.pow(2)
    E = torch.zeros((1,), dtype=torch.double, device=chi.device)
    for f in fns:
        E = E + f(box, den)
    E.backward()
    return E.detach(), chi.grad.detach()
