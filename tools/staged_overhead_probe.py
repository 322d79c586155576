"""Host-sequenced evaluation, cost of the sequencing itself: ONE rank driven through the staged ABI (ofdft_dist_begin / _stage x 8 /
_finish from Python, side-stream hand-offs, two RCCL all-reduces on the device scalars; a one-rank context has nothing to
transpose, so no all-to-all is issued) against the plain single-GPU engine on the same grid: what the host side of the collective
transport costs per evaluation before any link is involved.  usage: python tools/staged_overhead_probe.py [N ...]"""
import json
import os
import sys
import time

os.environ['OFDFT_COMM_ONE_RANK'] = '1'
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from professad_amd import synth  # noqa: E402
from professad_amd.distributed import DistEngine  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402
from professad_amd.functionals import NativeTerms  # noqa: E402

dev = torch.device('cuda:0')
torch.cuda.set_device(dev)
dist.init_process_group('nccl', init_method='tcp://127.0.0.1:29741', world_size=1, rank=0)
names = NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe']).names
for n in [int(a) for a in sys.argv[1:]] or [64, 128, 256]:
    shape = (n, n, n)
    box = torch.as_tensor(synth.cubic_cell(n))
    chi = torch.as_tensor(np.sqrt(synth.smooth_density(shape, seed=3)), device=dev)
    vext = torch.as_tensor(synth.random_potential(shape, seed=4), device=dev)
    row = {'grid': n}
    for label, eng in (('single_gpu_engine', Engine(shape, dev).set_cell(box).set_terms(names)),
                       ('staged_one_rank_rccl', DistEngine(shape, dev).set_cell(box).set_terms(names))):
        for _ in range(4):
            E, mu, g = eng.energy_grad_chi(chi, 50.0, vext)
        torch.cuda.synchronize()
        reps = 30
        t0 = time.perf_counter()
        for _ in range(reps):
            E, mu, g = eng.energy_grad_chi(chi, 50.0, vext)
        torch.cuda.synchronize()
        row[label + '_ms'] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        row[label + '_E'] = sum(E.values())
        eng.close()
    row['rel_dE'] = abs(row['single_gpu_engine_E'] - row['staged_one_rank_rccl_E']) / abs(row['single_gpu_engine_E'])
    print(json.dumps(row), flush=True)
dist.destroy_process_group()
