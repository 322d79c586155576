"""Ionic (external) potential from ion positions -- the step right before the energy path.

`read_recpot` restates the reference's recpot parsing (src/professad/ion_utils.py:20-81: units, ion charge from the
first two table entries, Coulomb tail added for interpolation); `ionic_potential` builds v_ext on the GPU through
`ofdft_ionic_potential` (exact or particle-mesh-Ewald structure factor), species by species like
System.__potential_from_ions (src/professad/system.py:183-194).
"""
import ctypes as C
import math

import numpy as np
import torch

# the recpot unit conversion uses its own constants in the reference (ion_utils.py:11-13)
BOHR = 0.529177208607388
HARTREE_TO_EV = 27.2113834279111
POT_CONV = 1.0 / (BOHR * BOHR * BOHR * HARTREE_TO_EV)


def recpot_table(raw, k_max):
    """(ks, v, z) from the raw table in atomic units: uniform k grid, Coulomb tail 4 pi z / k^2 added for k > 0."""
    raw = np.asarray(raw, dtype=np.float64)
    ks, dk = np.linspace(0.0, float(k_max), raw.size, retstep=True)
    z = round((raw[1] - raw[0]) * dk * dk / (-4 * math.pi))                     # ion_utils.py:66
    v = raw.copy()
    v[1:] += 4 * math.pi * z / (ks[1:] * ks[1:])                                # ion_utils.py:67
    return ks, v, z


def recpot_fields(path):
    """(raw table values [file units], k_max [1/angstrom]) of a CASTEP-style .recpot file.

    Layout after the comment block (ion_utils.py:49-73 reads the same fields): one line of two integers, one line with
    k_max, then the table in rows of three numbers; a trailing row with fewer than three entries (the 1000 marker) is
    not part of the table."""
    with open(path, 'r') as fh:
        text = fh.read()
    head, sep, tail = text.partition('END COMMENT')
    if not sep:
        raise ValueError('%s: no END COMMENT marker' % path)
    rows = [ln.split() for ln in tail.splitlines()[1:]]       # [0] is the rest of the marker line
    rows = [r for r in rows if r]
    k_max = float(rows[1][0])                                 # rows[0] = the two integers
    table = [x for r in rows[2:] if len(r) == 3 for x in r]
    return np.asarray(table, dtype=np.float64), k_max


def read_recpot(path):
    """Parse a CASTEP-style .recpot file -> (ks, v, z) as `recpot_table` (units of ion_utils.py:11-13,62-66)."""
    raw, k_max = recpot_fields(path)
    return recpot_table(raw * POT_CONV, k_max * BOHR)


def _f64(engine):
    """The once-per-geometry-step routines are fp64 work: an fp32 engine hands them to its fp64 sibling (same grid,
    same device); densities are widened on the way in, the potential is narrowed on the way out."""
    if engine.dtype == torch.double:
        return engine
    from .engine import engine_for
    return engine_for(engine.global_shape, engine.device)


def ionic_potential(engine, box_vecs, species, pme_order=None):
    """v_ext on the engine's grid.  species: iterable of (frac_coords [n,3], (ks, v, z)) per ion type.
    pme_order None -> exact O(N_ion N_k) structure factor; even int >= 2 -> particle-mesh Ewald."""
    want = engine.dtype
    engine = _f64(engine)
    engine.set_cell(box_vecs)
    out = torch.zeros(engine.shape, dtype=torch.double, device=engine.device)
    dp = C.POINTER(C.c_double)
    for i, (frac, (ks, v, z)) in enumerate(species):
        frac = np.ascontiguousarray(np.asarray(torch.as_tensor(frac).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 3))
        ks = np.ascontiguousarray(ks, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        if ks.shape != v.shape or ks.ndim != 1:
            raise ValueError('table k and v must be 1-D arrays of equal length')
        rc = engine.lib.ofdft_ionic_potential(engine._ctx, frac.ctypes.data_as(dp), frac.shape[0], ks.ctypes.data_as(dp),
                                              v.ctypes.data_as(dp), ks.size, float(z), 0 if pme_order is None else int(pme_order),
                                              C.c_void_p(out.data_ptr()), 1 if i else 0, engine._stream())
        engine._check(rc, 'ofdft_ionic_potential')
    return out.to(want)


def ion_electron_forces(engine, box_vecs, den, species, pme_order=None):
    """Forces F = -dU/dR of U = int n v_ext on every ion (Ha/bohr), species by species -> list of [n,3] arrays
    (the IonElectron part of System.forces(), system.py:913-923)."""
    den = engine._grid_tensor(den, 'den').double()
    engine = _f64(engine)
    engine.set_cell(box_vecs)
    dp = C.POINTER(C.c_double)
    out = []
    for frac, (ks, v, z) in species:
        frac = np.ascontiguousarray(np.asarray(torch.as_tensor(frac).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 3))
        ks = np.ascontiguousarray(ks, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        f = np.zeros_like(frac)
        rc = engine.lib.ofdft_ion_electron_forces(engine._ctx, C.c_void_p(den.data_ptr()), frac.ctypes.data_as(dp), frac.shape[0],
                                                  ks.ctypes.data_as(dp), v.ctypes.data_as(dp), ks.size, float(z),
                                                  0 if pme_order is None else int(pme_order), f.ctypes.data_as(dp),
                                                  engine._stream())
        engine._check(rc, 'ofdft_ion_electron_forces')
        out.append(f)
    return out


def ion_electron_stress(engine, box_vecs, den, species, pme_order=None):
    """Ion-electron stress (3x3, Ha/bohr^3) with the potential rebuilt from the ions at fixed fractional coordinates
    (the IonElectron part of System.stress(), system.py:925-935), summed over species."""
    den = engine._grid_tensor(den, 'den').double()
    engine = _f64(engine)
    engine.set_cell(box_vecs)
    dp = C.POINTER(C.c_double)
    total = np.zeros((3, 3))
    for frac, (ks, v, z) in species:
        frac = np.ascontiguousarray(np.asarray(torch.as_tensor(frac).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 3))
        ks = np.ascontiguousarray(ks, dtype=np.float64)
        v = np.ascontiguousarray(v, dtype=np.float64)
        s = np.zeros(9)
        rc = engine.lib.ofdft_ion_electron_stress(engine._ctx, C.c_void_p(den.data_ptr()), frac.ctypes.data_as(dp), frac.shape[0],
                                                  ks.ctypes.data_as(dp), v.ctypes.data_as(dp), ks.size, float(z),
                                                  0 if pme_order is None else int(pme_order), s.ctypes.data_as(dp),
                                                  engine._stream())
        engine._check(rc, 'ofdft_ion_electron_stress')
        total += s.reshape(3, 3)
    return total


def ion_ion(engine, box_vecs, frac, charges, Rc=None):
    """Ion-ion energy [Ha], forces [n,3] (Ha/bohr) and stress [3,3] (Ha/bohr^3): ion_interaction_sum with System's
    parameter heuristics (ion_utils.py:293-333, system.py:733-754) and its autograd derivatives (system.py:913-935)."""
    engine = _f64(engine)
    engine.set_cell(box_vecs)
    dp = C.POINTER(C.c_double)
    frac = np.ascontiguousarray(np.asarray(torch.as_tensor(frac).detach().cpu().numpy(), dtype=np.float64).reshape(-1, 3))
    z = np.ascontiguousarray(np.asarray(charges, dtype=np.float64).reshape(-1))
    if z.size != frac.shape[0]:
        raise ValueError('one charge per ion')
    E = C.c_double(0.0)
    F = np.zeros_like(frac)
    S = np.zeros(9)
    rc = engine.lib.ofdft_ion_ion(engine._ctx, frac.ctypes.data_as(dp), z.ctypes.data_as(dp), frac.shape[0],
                                  float(Rc) if Rc else 0.0, C.byref(E), F.ctypes.data_as(dp), S.ctypes.data_as(dp),
                                  engine._stream())
    engine._check(rc, 'ofdft_ion_ion')
    return E.value, F, S.reshape(3, 3)
