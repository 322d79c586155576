"""First rows of the config-1 optimisation log (E in eV per outer iteration) with both optimisers on both closures (persistent
kernel / staged pipeline), beside the reference's log: how far round-off-level differences of the closure move the fixed-step
L-BFGS trajectory.  usage: python tools/opt_rows_probe.py"""
import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo/tests/golden')
from professad_amd.engine import Engine
from professad_amd import functionals as F
from professad_amd.optimize import optimize_density
d = np.load('/root/repo/tests/golden/cfg1_fccAl_32.npz')
box, vext, n_elec = d['box'], d['vext'], float(d['n_elec'])
dev = 'cuda:0'
for opt in ('fused', 'torch'):
    for res_mode in (2, 0):
        eng = Engine((32, 32, 32), dev).set_cell(torch.as_tensor(box)).set_terms(F.NativeTerms(['ion_electron', 'hartree', 'tf', 'vw', 'pz']).names).set_option(10, res_mode)
        res = optimize_density(eng, n_elec, torch.as_tensor(vext, device=dev), volume=abs(np.linalg.det(box)), optimizer=opt)
        print(opt, 'resident', res_mode, res['iterations'], [round(r[1], 6) for r in res['history'][:6]])
        eng.close()
print('reference', [68.191536, 65.989145, 65.547128, 65.459420])
