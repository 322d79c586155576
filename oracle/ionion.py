"""ORACLE (test infrastructure, never shipped or measured as the product).

numpy restatement of the reference's ion-ion energy (ion_utils.py:293-333: real-space pairwise damped electrostatic
sum in a neutralising background, Phys. Rev. Materials 2, 013806) with the parameter heuristics of
System.__ion_ion_interaction (system.py:733-754), plus the forces and stress autograd gives the reference
(system.py:913-935), written out analytically.

The reference gets its pair list from torch-nl's compute_neighborlist (third-party, unpinned in pyproject.toml, absent
here): the pair set is restated as every (i, j, lattice shift) with 0 < |R_j + shift - R_i| <= Rc, enumerated over a
bounding box of shifts.  Parity status: energies PINNED by the reference's own known-answer tests
(tests/test_ion_utils.py:12-147, CASTEP / Madelung values to 1e-10, stored as data in
tests/golden/ion_ion_known_answers.json); forces and stress are checked against finite differences of that energy, as the
reference's own test does (tests/test_ion_utils.py:149-180) -- no reference output exists for them without torch-nl.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math

import numpy as np
from scipy.special import erf, erfc

PI = math.pi


def heuristics(box, Rc=None):
    """Rd, Rc of System.__ion_ion_interaction (system.py:744-750)"""
    h = 1.0 / np.sqrt(np.sum(np.linalg.inv(box.T) ** 2, axis=1))
    h_max = h.max()
    if Rc is None:
        Rd = 2 * h_max
        return 3 * Rd * Rd / h_max, Rd
    return Rc, math.sqrt(h_max * Rc / 3)


def pairs(box, coords, Rc):
    """all (i, j, d) with d = R_j + shift - R_i, 0 < |d| <= Rc"""
    h = 1.0 / np.sqrt(np.sum(np.linalg.inv(box.T) ** 2, axis=1))       # interplanar spacings
    frac = coords @ np.linalg.inv(box)
    span = frac.max(0) - frac.min(0)
    nmax = np.ceil(Rc / h + span).astype(int)
    s = np.stack(np.meshgrid(*[np.arange(-n, n + 1) for n in nmax], indexing='ij'), -1).reshape(-1, 3) @ box
    out = []
    for i in range(coords.shape[0]):
        for j in range(coords.shape[0]):
            d = coords[j] + s - coords[i]
            r = np.linalg.norm(d, axis=1)
            m = (r > 1e-12) & (r <= Rc)
            out.append((i, j, d[m], r[m]))
    return out


def energy(box, coords, charges, Rc, Rd):
    """ion_utils.py:293-333"""
    vol = abs(np.linalg.det(box))
    rho = charges.sum() / vol
    Q = charges.astype(np.float64).copy()
    E_local = 0.0
    for i, j, d, r in pairs(box, coords, Rc):
        Q[i] += charges[j] * r.size
        E_local += 0.5 * charges[i] * charges[j] * np.sum(erfc(r / Rd) / r)
    Ra = np.cbrt(0.75 / PI * Q / rho)
    E_corr = np.sum(-PI * charges * rho * Ra ** 2 + PI * charges * rho * (Ra ** 2 - 0.5 * Rd * Rd) * erf(Ra / Rd)
                    + math.sqrt(PI) * charges * rho * Ra * Rd * np.exp(-Ra ** 2 / (Rd * Rd))
                    - charges ** 2 / math.sqrt(PI) / Rd)
    return E_local + E_corr


def forces_stress(box, coords, charges, Rc, Rd):
    """-dE/dR_i and (1/vol) dE/d eps at fixed pair list, Rc, Rd (what autograd differentiates): the pair term through
    r_ij, the background term through rho = sum(Z)/vol and Ra = (3 Q_i / (4 pi rho))^(1/3)."""
    vol = abs(np.linalg.det(box))
    rho = charges.sum() / vol
    n = coords.shape[0]
    F = np.zeros((n, 3))
    sig = np.zeros((3, 3))
    Q = charges.astype(np.float64).copy()
    for i, j, d, r in pairs(box, coords, Rc):
        Q[i] += charges[j] * r.size
        fp = charges[i] * charges[j] * (-2 / (math.sqrt(PI) * Rd) * np.exp(-(r / Rd) ** 2) / r - erfc(r / Rd) / r ** 2)
        F[i] += np.sum((fp / r)[:, None] * d, axis=0)
        sig += 0.5 * np.einsum('p,pa,pb->ab', fp / r, d, d)
    Ra = np.cbrt(0.75 / PI * Q / rho)
    Z = charges
    ex, er = np.exp(-Ra ** 2 / (Rd * Rd)), erf(Ra / Rd)
    e_rho = (-PI * Z * Ra ** 2 + PI * Z * (Ra ** 2 - 0.5 * Rd * Rd) * er + math.sqrt(PI) * Z * Ra * Rd * ex)      # dE/d rho
    dE_dRa = (-2 * PI * Z * rho * Ra + 2 * PI * Z * rho * Ra * er
              + PI * Z * rho * (Ra ** 2 - 0.5 * Rd * Rd) * 2 / (math.sqrt(PI) * Rd) * ex
              + math.sqrt(PI) * Z * rho * Rd * ex * (1 - 2 * Ra ** 2 / (Rd * Rd)))
    # d rho / d eps_aa = -rho, d Ra / d eps_aa = Ra / 3
    sig += np.sum(-rho * e_rho + dE_dRa * Ra / 3) * np.eye(3)
    return F, sig / vol
