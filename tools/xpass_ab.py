#!/usr/bin/env python3
"""A/B of the fused x pass kernels on one box: for each shape, the cfg3 closure with OFDFT_OPT_XWAVE = each listed value --
results against the first value's (energy, mu, max gradient difference) and the x-pass classes' serialised times.
usage: python tools/xpass_ab.py [f32] [opts=1,5,6] 256x256x256 512x512x512 ...   -> one JSON line per (shape, option)"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from professad_amd.engine import Engine  # noqa: E402

CFG3 = ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']
DT = torch.float32 if 'f32' in sys.argv[1:] else torch.double
opts = [1, 5]
for a in sys.argv[1:]:
    if a.startswith('opts='):
        opts = [int(x) for x in a[5:].split(',')]
for arg in [a for a in sys.argv[1:] if a[0].isdigit()]:
    shape = tuple(int(x) for x in arg.split('x'))
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(1)
    chi = (0.17 * (1.0 + 0.2 * torch.rand(shape, generator=g, dtype=torch.double))).sqrt().to(dev).to(DT)
    vext = (0.1 * torch.rand(shape, generator=g, dtype=torch.double)).to(dev).to(DT)
    box = np.diag([7.65 * s / 32.0 for s in shape])
    nel = float(0.17 * 1.1 * np.prod(np.diag(box)))
    ref = None
    for opt in opts:
        eng = Engine(shape, dev, dtype=DT).set_cell(torch.as_tensor(box)).set_terms(CFG3)
        eng.set_option(8, opt)
        for _ in range(3):
            E, mu, gr = eng.energy_grad_chi(chi, nel, vext)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            E, mu, gr = eng.energy_grad_chi(chi, nel, vext)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 5 * 1e3
        Et = sum(E.values())
        if ref is None:
            ref = (Et, mu, gr.clone())
        eng.set_option(1, 0)
        eng.set_profiling(True)
        for _ in range(3):
            eng.energy_grad_chi(chi, nel, vext)
        prof = eng.profile()
        print(json.dumps({'shape': shape, 'dtype': str(DT), 'xwave': opt, 'ms': round(ms, 3),
                          'rel_dE': abs(Et - ref[0]) / abs(ref[0]), 'rel_dmu': abs(mu - ref[1]) / abs(ref[1]),
                          'dgrad_rel': float((gr - ref[2]).abs().max() / ref[2].abs().max()),
                          'x_us': {k: round(v[0] / 3 * 1e3, 1) for k, v in sorted(prof.items()) if k.startswith('xfused')}}), flush=True)
        eng.close()
    del chi, vext, ref
