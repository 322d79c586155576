"""Soak of the persistent small-grid kernel: the same evaluation many thousand times -- every result must be bitwise the first
one (its sums are formed in a fixed order; a missed barrier or a stale cache line would show as a different bit somewhere).
usage: python tools/resident_soak.py [evaluations per case]"""
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

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = 'cuda:0'
CASES = {'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'pz'], 'cfg2': ['ion_electron', 'hartree', 'wt', 'pz'],
         'wtpbe': ['ion_electron', 'hartree', 'wt', 'pbe'], 'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe']}
out = []
for n in (16, 32, 64):
    shape = (n, n, n)
    rng = np.random.default_rng(n)
    chi = torch.as_tensor(np.sqrt(synth.smooth_density(shape, seed=3)) * (1 + 0.1 * rng.random(shape)), device=dev)
    vext = torch.as_tensor(synth.random_potential(shape, seed=4), device=dev)
    for cfg, terms in CASES.items():
        eng = Engine(shape, dev).set_cell(torch.as_tensor(synth.triclinic_cell(n / 4.0))).set_terms(NativeTerms(terms).names)
        E0, mu0, g0 = eng.energy_grad_chi(chi, 11.0, vext)
        if not eng.query(N.Q_RESIDENT_EVALS):
            eng.close()
            continue
        bad = 0
        t0 = time.perf_counter()
        for i in range(reps):
            E, mu, g = eng.energy_grad_chi(chi, 11.0, vext)
            if E != E0 or mu != mu0 or (i % 64 == 0 and not torch.equal(g, g0)):
                bad += 1
        dt = time.perf_counter() - t0
        assert torch.equal(g, g0)
        out.append({'grid': n, 'cfg': cfg, 'evaluations': reps, 'different_results': bad, 'ms_per_eval': round(dt / reps * 1e3, 4)})
        print(json.dumps(out[-1]), flush=True)
        eng.close()
assert all(r['different_results'] == 0 for r in out), 'the persistent kernel is not reproducible'
print('soak ok: %d evaluations' % sum(r['evaluations'] for r in out))
