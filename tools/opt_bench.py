#!/usr/bin/env python3
"""Density-optimisation loop timing on one GPU: outer steps of the fixed-step L-BFGS around the cfg3 closure at
N^3, with the fused device-side optimiser and with the op-by-op torch form.  usage: opt_bench.py [N] [outer steps] [cfg3|cfg1|wtpbe]"""
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
from professad_amd.functionals import NativeTerms  # noqa: E402
from professad_amd.optimize import FixedStepLBFGS, HipLbfgsBackend, VectorFreeLBFGS  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cfg = sys.argv[3] if len(sys.argv) > 3 else 'cfg3'
    terms = {'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe'], 'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'pz'],
             'wtpbe': ['ion_electron', 'hartree', 'wt', 'pbe']}[cfg]
    shape = (n, n, n)
    dev = torch.device('cuda:0')
    box = torch.as_tensor(synth.cubic_cell(n))
    vol = float(abs(np.linalg.det(box.numpy())))
    n_elec = 12.0 * (n // 32) ** 3
    vext = torch.as_tensor(synth.random_potential(shape, seed=42), device=dev)
    eng = Engine(shape, dev).set_cell(box).set_terms(NativeTerms(terms).names)
    out = {'grid': n, 'outer_steps': steps, 'cfg': cfg}
    for name in ('fused', 'torch'):
        chi = torch.full(shape, float(np.sqrt(n_elec / vol)), dtype=torch.double, device=dev)
        chi *= 1 + 0.05 * torch.as_tensor(synth.smooth_density(shape, seed=5, n0=1.0, amp=1.0), device=dev)
        t_closure = [0.0]

        def closure():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            E, mu, g = eng.energy_grad_chi(chi, n_elec, vext)
            torch.cuda.synchronize()
            t_closure[0] += time.perf_counter() - t0
            return sum(E.values()), g

        opt = (VectorFreeLBFGS(chi, HipLbfgsBackend(chi.numel(), 8, dev)) if name == 'fused' else FixedStepLBFGS(chi))
        opt.step(closure)                 # warm-up (fills part of the history)
        e0, i0 = opt.func_evals, opt.total_iter
        t_closure[0] = 0.0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = opt.step(closure)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        evals, iters = opt.func_evals - e0, opt.total_iter - i0
        out[name] = {'inner_iterations': iters, 'closure_calls': evals, 'wall_ms': round(wall * 1e3, 2),
                     'closure_ms_per_call': round(t_closure[0] / evals * 1e3, 3),
                     'optimizer_ms_per_inner_iteration': round((wall - t_closure[0]) / iters * 1e3, 3), 'last_loss': loss}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
