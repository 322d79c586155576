#!/usr/bin/env python3
"""Per-kernel-class time per grid point for non-cubic shapes (which axis makes a pass slow?): cfg3 closure, streams serialised.
usage: python tools/shape_probe.py [f32] [cfg2] [nomixed] 256x256x256 512x256x256 ...   -> one JSON line per shape
(f32: the fp32 build; cfg2: the Wang-Teter + LDA term set of BASELINE configs 2 / 5 instead of the bench's;
nomixed: OFDFT_OPT_MIXED_RADIX off = chirp-z transforms + unfused pipeline on extents with factors 3 / 5)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from professad_amd.engine import Engine  # noqa: E402

CFG3 = ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']
if 'cfg2' in sys.argv[1:]:
    CFG3 = ['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'lda_x', 'pz_c']
DT = torch.float32 if 'f32' in sys.argv[1:] else torch.double
NOMIXED = 'nomixed' in sys.argv[1:]
for arg in [a for a in sys.argv[1:] if a[0].isdigit()]:
    shape = tuple(int(x) for x in arg.split('x'))
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(1)
    chi = (0.17 * (1.0 + 0.2 * torch.rand(shape, generator=g, dtype=torch.double))).sqrt().to(dev).to(DT)
    vext = (0.1 * torch.rand(shape, generator=g, dtype=torch.double)).to(dev).to(DT)
    box = np.diag([7.65 * s / 32.0 for s in shape])
    nel = float(0.17 * 1.1 * np.prod(np.diag(box)))
    eng = Engine(shape, dev, dtype=DT).set_cell(torch.as_tensor(box)).set_terms(CFG3)
    if NOMIXED:
        eng.set_option(9, 0)
    for a in sys.argv[1:]:          # optK=V: ofdft_set_option(K, V) (A/B of run-time switches, e.g. opt28=0: no folded WGC99 table reads)
        if a.startswith('opt') and '=' in a:
            eng.set_option(int(a[3:a.index('=')]), float(a[a.index('=') + 1:]))
    for _ in range(3):
        eng.energy_grad_chi(chi, nel, vext)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5):
        eng.energy_grad_chi(chi, nel, vext)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    eng.set_option(1, 0)
    eng.set_profiling(True)
    for _ in range(3):
        eng.energy_grad_chi(chi, nel, vext)
    prof = eng.profile()
    npts = float(np.prod(shape))
    print(json.dumps({'shape': shape, 'opts': [a for a in sys.argv[1:] if a.startswith('opt') and '=' in a], 'terms': 'cfg2' if 'cfg2' in sys.argv[1:] else 'cfg3', 'dtype': str(DT), 'mixed_radix': not NOMIXED, 'ms': round(ms, 3), 'ns_per_point': round(ms * 1e6 / npts, 4),
                      'ps_per_point': {k: round(v[0] / 3 * 1e9 / npts, 1) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}}), flush=True)
    eng.close()
    del chi, vext
