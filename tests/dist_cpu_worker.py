"""gloo/CPU worker: drives the product's multi-rank host logic with the numpy stage double."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import cases  # noqa: E402
from dist_double import NumpyStages  # noqa: E402
from oracle import closed_form as cf  # noqa: E402
from professad_amd import synth  # noqa: E402
from professad_amd.distributed import Comm, run_closure, run_potential  # noqa: E402


def main():
    shape = tuple(int(x) for x in sys.argv[1].split('x'))
    out = sys.argv[2]
    dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()
    box = cases.make_cell(('tri', 0.9))
    den = synth.random_density(shape, seed=51)
    vext = synth.random_potential(shape, seed=52)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(53).random(shape))
    n_elec = 7.3
    comm = Comm()
    assert comm.active and comm.nranks == world and comm.backend == 'gloo'
    st = NumpyStages(shape, box, world, rank)
    st.nchunks = int(sys.argv[3]) if len(sys.argv) > 3 else 1       # kz chunks of the exchange (step / chunk sequencing)
    pl = st.plan
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    vol, npts = abs(np.linalg.det(box)), int(np.prod(shape))
    E, mu, g = run_closure(st, comm, t(pl.scatter(chi)), n_elec, t(pl.scatter(vext)), vol, npts, torch.empty_like)
    # default: both chains on the world group (one communicator: the ordering every rank shares by construction); opt-in
    # (OFDFT_COMM_TWO_GROUPS=1): the chains' exchanges go out on different process groups (under RCCL: one communicator and
    # stream each).  Exchanges per chain here: one per chunk in steps 1 (both chains) and 2 (chain 0)
    if os.environ.get('OFDFT_COMM_TWO_GROUPS') == '1':
        assert comm.groups[0] is not comm.groups[1] and comm.groups[0] is None, comm.groups
    else:
        assert comm.groups[0] is comm.groups[1] and comm._own_group is None, comm.groups
    assert comm.issued == [2 * st.nchunks, st.nchunks] and st.chain1_ok == st.nchunks, (comm.issued, st.chain1_ok)
    E2, v = run_potential(st, comm, t(pl.scatter(den)), t(pl.scatter(vext)), vol, npts, torch.empty_like)
    gs = [torch.empty(pl.local_shape, dtype=torch.double) for _ in range(world)]
    vs = [torch.empty(pl.local_shape, dtype=torch.double) for _ in range(world)]
    dist.all_gather(gs, g.contiguous())
    dist.all_gather(vs, v.contiguous())
    if rank == 0:
        ev = cf.Evaluator(cf.Grid(box, shape))
        names = ['ion_electron', 'hartree', 'tf']
        Ec, go, muo = ev.closure(names, chi, n_elec, vext)
        Eo, Es, vo = ev.terms(names, den, vext)
        res = dict(dE=abs(sum(E.values()) - Ec) / abs(Ec), dmu=abs(mu - muo) / abs(muo),
                   dg=float(np.abs(torch.cat(gs).numpy() - go).max() / np.abs(go).max()),
                   dE2=abs(sum(E2.values()) - Eo) / abs(Eo),
                   dv=float(np.abs(torch.cat(vs).numpy() - vo).max() / np.abs(vo).max()))
        json.dump(res, open(out, 'w'))
    comm.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
