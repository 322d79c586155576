"""Worker of the multi-rank tests: `python dist_worker.py <case> <outfile> [f64|f32]` with RANK/WORLD_SIZE/MASTER_* set.
Default transport: all ranks share cuda:0 (gloo backend, host-staged exchange) so the slab-decomposed path -- HIP
kernels, pack / un-pack, stage sequencing, collectives -- is exercised with several ranks on a one-GPU box.
OFDFT_TEST_BACKEND=nccl: one GPU per rank (cuda:LOCAL_RANK) and the product transport, RCCL all-to-alls on the engine's
own exchange buffers (tests/test_dist_gpu.py enables that case when the box has at least two GPUs)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

import cases  # noqa: E402
from professad_amd import synth  # noqa: E402
from professad_amd.distributed import DistEngine  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402
from professad_amd.functionals import NativeTerms  # noqa: E402

CFG = {'cfg1': ['ion_electron', 'hartree', 'tf', 'vw', 'pz'], 'cfg2': ['ion_electron', 'hartree', 'wt', 'pz'],
       'cfg3': ['ion_electron', 'hartree', 'wgc99', 'pbe'], 'wgc98': ['wgc98x']}


def main():
    shape = tuple(int(x) for x in sys.argv[1].split('x'))
    out = sys.argv[2]
    dt = torch.float32 if (len(sys.argv) > 3 and sys.argv[3] == 'f32') else torch.double
    backend = os.environ.get('OFDFT_TEST_BACKEND', 'gloo')
    if backend == 'nccl':
        dev = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
        torch.cuda.set_device(dev)
        dist.init_process_group('nccl', device_id=dev)
    else:
        dev = torch.device('cuda:0')
        dist.init_process_group('gloo')
    rank, world = dist.get_rank(), dist.get_world_size()

    def gather_slabs(x):
        """every rank's slab of x -> list of host tensors"""
        if backend == 'nccl':
            full = torch.empty((world,) + tuple(x.shape), dtype=x.dtype, device=dev)
            dist.all_gather_into_tensor(full, x.contiguous())
            return list(full.cpu())
        parts = [torch.empty(tuple(x.shape), dtype=x.dtype) for _ in range(world)]
        dist.all_gather(parts, x.cpu())
        return parts
    box = cases.make_cell(('tri', 1.3))
    den = synth.random_density(shape, seed=41)
    vext = synth.random_potential(shape, seed=42)
    chi = np.sqrt(den) * (1 + 0.1 * np.random.default_rng(43).random(shape))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box))) + 0.3)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)  # noqa: E731
    transport = os.environ.get('OFDFT_TEST_TRANSPORT', 'collective')
    want_chunks = int(os.environ['OFDFT_TEST_XCHG_CHUNKS']) if os.environ.get('OFDFT_TEST_XCHG_CHUNKS') else None
    eng = DistEngine(shape, dev, dtype=dt, transport=transport, xchg_chunks=want_chunks).set_cell(torch.as_tensor(box))
    plan = eng.plan
    worst = {}
    if want_chunks:
        # the kz-chunked exchange (every step's exchange in `want_chunks` pieces, consumed piece by piece) against the unchunked
        # sequence of a second engine on the same ranks: same kernels per line, so every number must be IDENTICAL
        # (same transport: the ipc transport adds the ranks' sums in rank order, the collective one in the backend's order)
        one = DistEngine(shape, dev, dtype=dt, transport=transport, xchg_chunks=1).set_cell(torch.as_tensor(box))
        names = NativeTerms(CFG['cfg3']).names
        eng.set_terms(names)
        one.set_terms(names)
        worst['chunks'] = dict(dE=0.0, dE2=0.0, dmu=0.0, dg=0.0, dv=0.0, ffts=eng.stages.nchunks, ffts_ref=want_chunks)
        for rep in range(3):          # repeated: buffers alternate between exchanges and between evaluations
            E, mu, g = eng.energy_grad_chi(t(plan.scatter(chi)) * (1 + 0.01 * rep), n_elec, t(plan.scatter(vext)))
            E1, mu1, g1 = one.energy_grad_chi(t(plan.scatter(chi)) * (1 + 0.01 * rep), n_elec, t(plan.scatter(vext)))
            w = worst['chunks']
            w['dE'] = max(w['dE'], max(abs(E[k] - E1[k]) for k in E))
            w['dmu'] = max(w['dmu'], abs(mu - mu1))
            w['dg'] = max(w['dg'], float((g - g1).abs().max()))
        E2, v = eng.energy_potential(t(plan.scatter(den)), t(plan.scatter(vext)))
        E21, v1 = one.energy_potential(t(plan.scatter(den)), t(plan.scatter(vext)))
        worst['chunks']['dE2'] = max(abs(E2[k] - E21[k]) for k in E2)
        worst['chunks']['dv'] = float((v - v1).abs().max())
        assert one.stages.nchunks == 1
        one.close()
    for cfg in ('cfg1', 'cfg2', 'cfg3'):
        names = NativeTerms(CFG[cfg]).names
        eng.set_terms(names)
        E, mu, g = eng.energy_grad_chi(t(plan.scatter(chi)), n_elec, t(plan.scatter(vext)))
        E2, v = eng.energy_potential(t(plan.scatter(den)), t(plan.scatter(vext)))
        parts_g, parts_v = gather_slabs(g), gather_slabs(v)
        if rank == 0:
            ref = Engine(shape, dev, dtype=dt).set_cell(torch.as_tensor(box)).set_terms(names)
            Er, mur, gr = ref.energy_grad_chi(t(chi), n_elec, t(vext))
            Er2, vr = ref.energy_potential(t(den), t(vext))
            gfull, vfull = torch.cat(parts_g).numpy(), torch.cat(parts_v).numpy()
            worst[cfg] = dict(
                dE=max(abs(E[k] - Er[k]) / max(1.0, abs(Er[k])) for k in E),
                dE2=max(abs(E2[k] - Er2[k]) / max(1.0, abs(Er2[k])) for k in E2),
                dmu=abs(mu - mur) / max(1.0, abs(mur)),
                dg=float(np.abs(gfull - gr.cpu().numpy()).max() / np.abs(gr.cpu().numpy()).max()),
                dv=float(np.abs(vfull - vr.cpu().numpy()).max() / np.abs(vr.cpu().numpy()).max()),
                ffts=eng.query(0), ffts_ref=ref.query(0))
            ref.close()
    # stress from a density slab -- slab-decomposed (distributed transforms through the engine's collective callbacks,
    # per-rank partial sums + small all-reduces; no gathered grid) -- vs one GPU; two term sets so that every stress
    # kernel family runs on slabs (Hartree, WGC99, PBE | vW, Wang-Teter, TF, LDA, the q-dependent Pauli-Gaussian)
    worst['stress'] = dict(dE=0.0, dE2=0.0, dmu=0.0, dg=0.0, dv=0.0, ffts=0, ffts_ref=0)
    for bits, params in ((NativeTerms(['hartree', 'wgc99', 'pbe']).names, None),
                         (('tf', 'vw', 'wt_nl', 'lda_x', 'pz_c', 'gga_k'), {'ggak_kind': 1.0, 'ggak_beta': 0.25, 'ggak_lambda': 0.4})):
        sd = eng.stress(t(plan.scatter(den)), bits, params)
        if rank == 0:
            ref = Engine(shape, dev, dtype=dt).set_cell(torch.as_tensor(box)).set_terms(bits, params)
            sr = ref.stress(t(den))
            worst['stress']['dE'] = max(worst['stress']['dE'], max(float(np.abs(sd[k] - sr[k]).max()) for k in sr))
            ref.close()
    # ionic potential slab and ion-electron forces through the slab-decomposed engine vs one GPU
    from professad_amd.ions import ion_electron_forces, ionic_potential
    ks = np.linspace(0.0, 12.0, 400)
    tab = (ks, -4 * np.pi * 3.0 / (ks ** 2 + 1.5) * np.exp(-0.05 * ks ** 2) + np.where(ks > 0, 4 * np.pi * 3.0 / np.where(ks > 0, ks, 1.0) ** 2, 0.0) * 0 , 3)
    frac = np.array([[0.1, 0.2, 0.3], [0.6, 0.55, 0.8]])
    from professad_amd.ions import ion_electron_stress
    worst['ions'] = dict(dE=0.0, dE2=0.0, dmu=0.0, dg=0.0, dv=0.0, ffts=0, ffts_ref=0)
    for order in (4, None):          # particle-mesh Ewald (each rank spreads / gathers on its own planes) and the exact sum
        vs = eng.ionic_potential([(frac, tab)], pme_order=order)
        Fs = eng.ion_electron_forces(t(plan.scatter(den)), [(frac, tab)], pme_order=order)[0]
        Ss = eng.ion_electron_stress(t(plan.scatter(den)), [(frac, tab)], pme_order=order)
        if rank == 0:
            ref = Engine(shape, dev)
            vr = ionic_potential(ref, box, [(frac, tab)], pme_order=order)
            Fr = ion_electron_forces(ref, box, t(den).double(), [(frac, tab)], pme_order=order)[0]
            Sr = ion_electron_stress(ref, box, t(den).double(), [(frac, tab)], pme_order=order)
            worst['ions']['dE'] = max(worst['ions']['dE'], float((vs.double() - vr[plan.x_range()]).abs().max() / vr.abs().max()))
            worst['ions']['dE2'] = max(worst['ions']['dE2'], float(np.abs(Fs - Fr).max()), float(np.abs(Ss - Sr).max()))
            ref.close()
    # density optimisation over slabs (fused L-BFGS sweeps on each rank's slab, all-reduced scalars) vs one GPU
    if shape == (32, 32, 32):
        from professad_amd.optimize import optimize_density
        names = NativeTerms(CFG['cfg1']).names
        vol = abs(np.linalg.det(box))
        eng.set_terms(names)
        res = optimize_density(eng, n_elec, t(plan.scatter(vext)), volume=vol, n_maxiter=8)
        parts = gather_slabs(res['chi'])
        if rank == 0:
            ref = Engine(shape, dev, dtype=dt).set_cell(torch.as_tensor(box)).set_terms(names)
            rr = optimize_density(ref, n_elec, t(vext), volume=vol, n_maxiter=8)
            worst['opt'] = dict(dE=abs(res['E_Ha'] - rr['E_Ha']) / abs(rr['E_Ha']), dE2=0.0, dmu=0.0, dv=0.0,
                                dg=float((torch.cat(parts) - rr['chi'].cpu()).abs().max() / rr['chi'].abs().max()),
                                ffts=res['iterations'], ffts_ref=rr['iterations'])
            ref.close()
    if rank == 0:
        with open(out, 'w') as fh:
            json.dump(worst, fh)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
