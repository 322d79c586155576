#!/usr/bin/env python3
"""Soak of the chirp-z path (csrc/bluestein.h): repeated closure evaluations on odd grids must reproduce the first one BITWISE
(energies, mu, gradient checksum) -- the kernels synchronise their LDS hand-overs with wave-local fences and a few workgroup
barriers, and a missing one would show up as rare run-to-run differences.
usage: python tools/chirpz_soak.py [f32] [reps]   -> one JSON line per shape"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from professad_amd.engine import Engine  # noqa: E402

CFG3 = ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']
DT = torch.float32 if 'f32' in sys.argv[1:] else torch.double
REPS = [int(a) for a in sys.argv[1:] if a.isdigit()]
REPS = REPS[0] if REPS else 300
for shape, reps in (((53, 53, 53), REPS * 4), ((27, 35, 33), REPS * 4), ((63, 61, 59), REPS * 2), ((129, 135, 127), REPS), ((255, 255, 63), REPS),
                    ((255, 31, 255), REPS)):
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(7)
    chi = (0.17 * (1.0 + 0.2 * torch.rand(shape, generator=g, dtype=torch.double))).sqrt().to(dev).to(DT)
    vext = (0.1 * torch.rand(shape, generator=g, dtype=torch.double)).to(dev).to(DT)
    box = np.diag([7.65 * s / 32.0 for s in shape])
    nel = float(0.17 * 1.1 * np.prod(np.diag(box)))
    eng = Engine(shape, dev, dtype=DT).set_cell(torch.as_tensor(box)).set_terms(CFG3)
    ref = None
    bad = 0
    for r in range(reps):
        E, mu, grad = eng.energy_grad_chi(chi, nel, vext)
        sig = (tuple(sorted(E.items())), mu, float(grad.double().sum()), float((grad.double() * grad.double()).sum()))
        if ref is None:
            ref = sig
        elif sig != ref:
            bad += 1
    print(json.dumps({'shape': shape, 'dtype': str(DT), 'evaluations': reps, 'different_from_first': bad, 'graph_replays': int(eng.query(6)),
                      'E_total': sum(ref[0][i][1] for i in range(len(ref[0])))}), flush=True)
    eng.close()
    assert bad == 0, shape
