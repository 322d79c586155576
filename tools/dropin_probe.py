"""Milliseconds per call of the drop-in terms (professad_amd.functionals: one ofdft_energy_potential per term, as the
reference's System evaluates its `terms` list) on a small grid, persistent kernel on / off.  usage: python tools/dropin_probe.py [N]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from professad_amd import _native as N  # noqa: E402
from professad_amd import synth  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402
from professad_amd.functionals import NativeTerms  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shape, dev = (n, n, n), 'cuda:0'
den = torch.as_tensor(synth.smooth_density(shape, seed=3), device=dev)
vext = torch.as_tensor(synth.random_potential(shape, seed=4), device=dev)
row = {'grid': n}
for label, terms in (('ion_electron', ['ion_electron']), ('hartree', ['hartree']), ('wt', ['wt']), ('pbe', ['pbe']),
                     ('all_fused', ['ion_electron', 'hartree', 'wt', 'pbe'])):
    for mode in (0, 2):
        eng = Engine(shape, dev).set_cell(torch.as_tensor(synth.cubic_cell(n))).set_terms(NativeTerms(terms).names).set_option(N.OPT_RESIDENT, mode)
        ve = vext if 'ion_electron' in terms else None
        for _ in range(5):
            eng.energy_potential(den, ve)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            eng.energy_potential(den, ve)
        torch.cuda.synchronize()
        row['%s_%s_ms' % (label, 'resident' if mode else 'staged')] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
        eng.close()
# the drop-in terms themselves, called the way the reference's System calls them (system.py:771): f(box_vecs, den) with box_vecs a
# DEVICE tensor that System keeps for the whole optimisation -- and, for comparison, with a fresh box tensor per call (round 2's cost:
# a device -> host copy of the lattice vectors, i.e. a stream synchronisation, per term call)
from professad_amd import functionals as F  # noqa: E402
box_dev = torch.as_tensor(synth.cubic_cell(n), device=dev)
for label, fn in (('Hartree', F.Hartree), ('WangTeter', F.WangTeter), ('PerdewBurkeErnzerhof', F.PerdewBurkeErnzerhof)):
    for fresh in (False, True):
        for _ in range(5):
            fn(box_dev, den)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            float(fn(box_dev.clone() if fresh else box_dev, den))          # (float(): System reads the energy, system.py:871)
        torch.cuda.synchronize()
        row['dropin_%s_%s_ms' % (label, 'fresh_box' if fresh else 'same_box')] = round((time.perf_counter() - t0) / 200 * 1e3, 4)
print(json.dumps(row))
