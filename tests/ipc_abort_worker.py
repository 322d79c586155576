"""Worker of test_ipc_transport_fails_an_evaluation_on_all_ranks_and_recovers: two ranks sharing cuda:0 over the ipc transport
with a short patience; rank 1 arrives late for the SECOND evaluation.  Expected: rank 0's delivery wait runs out and posts the
abort word, rank 1 finds it when it finally arrives -- BOTH ranks raise for that evaluation -- and the third evaluation is
clean again on both (epochs derive from the evaluation number, buffers and parities restart per call)."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from professad_amd import _native as N  # noqa: E402
from professad_amd import synth  # noqa: E402
from professad_amd.distributed import DistEngine  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402


def main():
    out = sys.argv[1]
    dist.init_process_group('gloo')
    rank = dist.get_rank()
    dev = torch.device('cuda:0')
    shape = (32, 32, 32)
    box = synth.triclinic_cell(1.3)
    den = synth.random_density(shape, seed=41)
    vext = synth.random_potential(shape, seed=42)
    chi = np.sqrt(den)
    names = ['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'lda_x', 'pz_c']
    eng = DistEngine(shape, dev, transport='ipc').set_cell(torch.as_tensor(box)).set_terms(names)
    eng.stages.set_option(N.OPT_IPC_WAIT_MS, 400.0)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.double, device=dev)  # noqa: E731
    xs = eng.plan.x_range()
    c, v = t(chi[xs]), t(vext[xs])
    res = {'rank': rank}
    E1, mu1, g1 = eng.energy_grad_chi(c, 9.0, v)            # evaluation 1: healthy
    dist.barrier()
    if rank == 1:
        time.sleep(2.0)                                     # far beyond the 0.4 s patience
    try:
        eng.energy_grad_chi(c, 9.0, v)                      # evaluation 2: must fail on BOTH ranks
        res['second'] = 'returned'
    except RuntimeError as e:
        res['second'] = 'raised: ' + str(e)[-160:]
    dist.barrier()
    E3, mu3, g3 = eng.energy_grad_chi(c, 9.0, v)            # evaluation 3: clean again
    res['third_equals_first'] = bool(all(E1[k] == E3[k] for k in E1) and mu1 == mu3 and torch.equal(g1, g3))
    if rank == 0:
        ref = Engine(shape, dev).set_cell(torch.as_tensor(box)).set_terms(names)
        Er, mur, gr = ref.energy_grad_chi(t(chi), 9.0, t(vext))
        res['dE_vs_single_gpu'] = max(abs(E3[k] - Er[k]) for k in Er)
        ref.close()
    with open('%s.%d' % (out, rank), 'w') as fh:
        json.dump(res, fh)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
