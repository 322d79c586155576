#!/usr/bin/env python3
"""One-GPU probe of the multi-GPU decomposition: single-engine evaluation time at a grid size, and the LOCAL compute
time per rank and evaluation when the same problem is cut into P slabs (8 contexts emulated in one process; the
exchange itself is not timed -- it is a device copy here, xGMI on a real node).  usage: scale_probe.py N [P]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from local_ranks import LocalRanks  # noqa: E402
from professad_amd import synth  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402
from professad_amd.functionals import NativeTerms  # noqa: E402


def main():
    n = int(sys.argv[1])
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    shape = (n, n, n)
    dev = torch.device('cuda:0')
    box = torch.as_tensor(synth.cubic_cell(n))
    rng = np.random.default_rng(5)
    chi = torch.sqrt(0.03 * (1 + 0.2 * torch.rand(shape, dtype=torch.double, device=dev)))
    vext = -0.1 * torch.rand(shape, dtype=torch.double, device=dev)
    n_elec = 12.0 * (n // 32) ** 3
    names = NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe']).names
    eng = Engine(shape, dev).set_cell(box).set_terms(names)
    for _ in range(2):
        eng.energy_grad_chi(chi, n_elec, vext)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        Er, mur, gr = eng.energy_grad_chi(chi, n_elec, vext)
    torch.cuda.synchronize()
    single = (time.perf_counter() - t0) / reps
    eng.close()
    from professad_amd.distributed import DistEngine
    de = DistEngine(shape, dev).set_cell(box).set_terms(names)      # one rank: the staged host protocol without exchange
    for _ in range(2):
        de.energy_grad_chi(chi, n_elec, vext)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        de.energy_grad_chi(chi, n_elec, vext)
    torch.cuda.synchronize()
    staged1 = (time.perf_counter() - t0) / reps
    de.close()
    loc = LocalRanks(shape, dev, P).set_cell(box).set_terms(names)
    loc.closure(chi, n_elec, vext)
    loc.compute_s = [0.0] * P
    loc.exchanged_bytes = 0
    loc.st[0].set_option(1, 0)          # serialise rank 0's streams so that its per-kernel event times are clean
    loc.st[0].set_profiling(True)
    for _ in range(reps):
        E, mu, g = loc.closure(chi, n_elec, vext)
    per_rank = [s / reps for s in loc.compute_s]
    prof = {k: (round(ms / reps, 4), n // reps) for k, (ms, n) in sorted(loc.st[0].profile().items(), key=lambda kv: -kv[1][0])}
    # ... and rank 0's evaluation the way a real rank runs it: every step enqueued without a host wait (both streams, chunked
    # launches, the device-resident scalars), the exchanges skipped, ONE synchronisation at the end (the emulator above waits for the
    # device around every step to attribute time to ranks: 48 waits per evaluation inflate its per-rank figure)
    from professad_amd.distributed import Comm, _NoExchange, run_closure
    st0 = loc.st[0]
    st0.set_profiling(False)
    st0.set_option(1, 1)
    xs = st0.plan.x_range()
    chi0, vext0 = chi[xs].contiguous(), vext[xs].contiguous()
    nocomm = _NoExchange(Comm())
    for _ in range(2):
        run_closure(st0, nocomm, chi0, n_elec, vext0, loc.vol, loc.npts, torch.empty_like)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps * 2):
        run_closure(st0, nocomm, chi0, n_elec, vext0, loc.vol, loc.npts, torch.empty_like)
    torch.cuda.synchronize()
    async_wall = (time.perf_counter() - t0) / (reps * 2)
    print('rank-0 kernels (ms per eval, launches per eval):', prof, file=sys.stderr)
    loc.close()
    err = float((g - gr).abs().max() / gr.abs().max())
    print(json.dumps({'grid': n, 'ranks': P, 'single_gpu_ms': round(single * 1e3, 3), 'staged_protocol_1rank_ms': round(staged1 * 1e3, 3),
                      'local_compute_ms_per_rank_max': round(max(per_rank) * 1e3, 3),
                      'local_compute_ms_per_rank_mean': round(float(np.mean(per_rank)) * 1e3, 3),
                      'rank0_wall_ms_no_exchange_async': round(async_wall * 1e3, 3),
                      'exchange_MB_per_rank_per_eval': round(loc.exchanged_bytes / reps / P * (P - 1) / P / 1e6, 1),
                      'grad_rel_diff_vs_single': err}))


if __name__ == '__main__':
    main()
