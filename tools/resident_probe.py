"""Closure evaluations on one small grid through the persistent kernel (for rocprofv3 --kernel-trace --stats).
usage: python tools/resident_probe.py N cfg1|cfg2|wtpbe|cfg3 [mode 1|2] [reps]"""
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

CFG = {'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'pz'], 'cfg2': ['ion_electron', 'hartree', 'wt', 'pz'],
       'wtpbe': ['ion_electron', 'hartree', 'wt', 'pbe'], 'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe']}

n, cfg = int(sys.argv[1]), sys.argv[2]
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 300
shape = (n, n, n)
dev = 'cuda:0'
chi = torch.as_tensor(np.sqrt(synth.smooth_density(shape, seed=3)), device=dev)
vext = torch.as_tensor(synth.random_potential(shape, seed=4), device=dev)
eng = Engine(shape, dev).set_cell(torch.as_tensor(synth.cubic_cell(n))).set_terms(NativeTerms(CFG[cfg]).names).set_option(N.OPT_RESIDENT, mode)
for _ in range(10):
    eng.energy_grad_chi(chi, 12.0, vext)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.energy_grad_chi(chi, 12.0, vext)
torch.cuda.synchronize()
print(json.dumps({'grid': n, 'cfg': cfg, 'mode': mode, 'ms_per_eval': round((time.perf_counter() - t0) / reps * 1e3, 4),
                  'resident_evals': int(eng.query(N.Q_RESIDENT_EVALS)),
                  'phase_clock_us': [round(eng.query(16 + i), 2) for i in range(12)]}))
