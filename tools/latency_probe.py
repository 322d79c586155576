"""Milliseconds per closure evaluation on small grids: kernel-by-kernel launches, the hipGraph replay, the persistent kernel
(host wall time per call, and the HIP-event time of the last call's device work).
usage: python tools/latency_probe.py [N ...]"""
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
       'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe'],
       'wtpbe': ['ion_electron', 'hartree', 'wt', 'pbe']}           # the reference's standard term set (tests/test_den_opt.py:59)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128]
    dev = 'cuda:0'
    for dt in (torch.double, torch.float32):
        for n in sizes:
            shape = (n, n, n)
            box = torch.as_tensor(synth.cubic_cell(n))
            den = synth.smooth_density(shape, seed=3)
            chi = torch.as_tensor(np.sqrt(den), dtype=dt, device=dev)
            vext = torch.as_tensor(synth.random_potential(shape, seed=4), dtype=dt, device=dev)
            row = {'grid': n, 'dtype': str(dt).replace('torch.', '')}
            for cfg, terms in CFG.items():
                for mode in ('launches', 'graph', 'resident', 'resident_untimed'):
                    graph = 0 if mode == 'launches' else 1
                    res = {'resident': 1, 'resident_untimed': 2}.get(mode, 0)
                    eng = (Engine(shape, dev, dtype=dt).set_cell(box).set_terms(NativeTerms(terms).names).set_option(N.OPT_GRAPH, graph)
                           .set_option(N.OPT_RESIDENT, res))
                    for _ in range(6):
                        eng.energy_grad_chi(chi, 12.0, vext)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    reps = 200
                    for _ in range(reps):
                        eng.energy_grad_chi(chi, 12.0, vext)
                    torch.cuda.synchronize()
                    if res and not eng.query(N.Q_RESIDENT_EVALS):
                        eng.close()
                        continue                                  # term set / grid the persistent kernel does not serve
                    row['%s_%s_ms' % (cfg, mode)] = round((time.perf_counter() - t0) / reps * 1e3, 4)
                    row['%s_%s_kernel_ms' % (cfg, mode)] = round(eng.query(N.Q_KERNEL_MS), 4)
                    eng.close()
            print(json.dumps(row), flush=True)


if __name__ == '__main__':
    main()
