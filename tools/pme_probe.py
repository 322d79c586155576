#!/usr/bin/env python3
"""Timing of the per-geometry-step ion routines at the config-5 size (run on the GPU box): PME ionic potential and
ion-electron forces for 4 (n/32)^3 ions on an n^3 grid, order 10.   python tools/pme_probe.py [n ...] >> gpurun_out/pme.jsonl"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from professad_amd import synth  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402
from professad_amd.ions import ion_electron_forces, ion_electron_stress, ionic_potential, recpot_table  # noqa: E402

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'recpots.npz'))
tab = recpot_table(g['al_raw'], float(g['al_kmax']))
for n in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
    r = n // 32
    frac32 = np.array([[0.0, 0.0, 0.0], [0.0, 0.5, 0.5], [0.5, 0.0, 0.5], [0.5, 0.5, 0.0]]) + 0.013
    shifts = np.stack(np.meshgrid(np.arange(r), np.arange(r), np.arange(r), indexing='ij'), -1).reshape(-1, 1, 3)
    frac = ((frac32[None] + shifts) / r).reshape(-1, 3)
    eng = Engine((n,) * 3, 'cuda:0')
    box = synth.cubic_cell(n)
    den = torch.as_tensor(synth.smooth_density((32,) * 3, seed=9), device='cuda:0').repeat(r, r, r)
    out = {'grid': n, 'ions': int(frac.shape[0]), 'pme_order': 10}
    for name, fn in (('ionic_potential_ms', lambda: ionic_potential(eng, box, [(frac, tab)], 10)),
                     ('ion_electron_forces_ms', lambda: ion_electron_forces(eng, box, den, [(frac, tab)], 10)),
                     ('ion_electron_stress_ms', lambda: ion_electron_stress(eng, box, den, [(frac, tab)], 10))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        out[name] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
    out['workspace_GB'] = round(eng.query(1) / 1e9, 2)
    eng.close()
    print(json.dumps(out), flush=True)
