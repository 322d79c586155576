"""P ranks of the slab decomposition emulated in ONE process on one GPU (test / diagnostics only).

Every rank is its own `ofdft_ctx` (created with nranks=P, rank=r); the all-to-all of each stage is done by copying
peer chunks between the contexts' exchange buffers, the all-reduces by summing on the host.  This runs the very
kernels, pack / un-pack geometry and stage sequence an 8-GPU job runs, at rank counts a one-GPU box cannot host
as processes.
"""
import time

import numpy as np
import torch

from professad_amd.distributed import HipStages


class LocalRanks:
    def __init__(self, shape, device, nranks, dtype=torch.double):
        self.P = nranks
        self.dev = device
        self.st = [HipStages(shape, device, nranks=nranks, rank=r, dtype=dtype) for r in range(nranks)]
        self.npts = int(np.prod(shape))
        self.compute_s = [0.0] * nranks
        self.exchanged_bytes = 0          # bytes in all ranks' send buffers (incl. the diagonal chunks)

    def set_cell(self, box):
        for s in self.st:
            s.set_cell(box)
        self.vol = float(abs(np.linalg.det(np.asarray(torch.as_tensor(box).cpu().numpy(), dtype=np.float64))))
        return self

    def set_terms(self, names, params=None):
        for s in self.st:
            s.set_terms(names, params)
        return self

    def _timed(self, r, fn, *a):
        torch.cuda.synchronize(self.dev)
        t0 = time.perf_counter()
        out = fn(*a)
        torch.cuda.synchronize(self.dev)
        self.compute_s[r] += time.perf_counter() - t0
        return out

    def set_xchg_chunks(self, n):
        for s in self.st:
            s.set_xchg_chunks(n)
        return self

    legacy_stage_api = False      # True: sequence with ofdft_dist_stage (whole stages; one chunk only) instead of ofdft_dist_step

    def _stages_legacy(self):
        P = self.P
        for k in (1, 2, 3, 4):
            for chain in (0, 1):
                ex = [self._timed(r, s.stage, k, chain) for r, s in enumerate(self.st)]
                if ex[0] is None:
                    continue
                self.exchanged_bytes += sum(e[0].numel() for e in ex)
                for r in range(P):
                    rc = ex[r][1].chunk(P)
                    for p in range(P):
                        rc[p].copy_(ex[p][0].chunk(P)[r])
        return sum(self._timed(r, s.finish) for r, s in enumerate(self.st))

    def _stages(self):
        if self.legacy_stage_api:
            return self._stages_legacy()
        P = self.P
        K = self.st[0].nchunks
        for step in (1, 2, 3, 4, 5, 6):
            for chain in (0, 1):
                for k in range(K):         # chunk k of the exchange: one equal-split all-to-all over the chunk's region
                    ex = [self._timed(r, s.step, step, chain, k) for r, s in enumerate(self.st)]
                    if ex[0] is None:
                        continue
                    self.exchanged_bytes += sum(e[0].numel() for e in ex)
                    for r in range(P):
                        rc = ex[r][1].chunk(P)
                        for p in range(P):
                            rc[p].copy_(ex[p][0].chunk(P)[r])
        return sum(self._timed(r, s.finish) for r, s in enumerate(self.st))

    def closure(self, chi, n_elec, vext):
        """full-grid chi / v_ext (device tensors) -> (E_terms, mu, full-grid dE/dchi)"""
        sl = [s.plan.x_range() for s in self.st]
        chis = [chi[x].contiguous() for x in sl]
        vexts = [vext[x].contiguous() if vext is not None else None for x in sl]
        s2 = sum(self._timed(r, s.sumsq, chis[r], True) for r, s in enumerate(self.st))
        cscale = n_elec / (s2 / self.npts * self.vol)
        vs = [torch.empty_like(c) for c in chis]
        for r, s in enumerate(self.st):
            self._timed(r, s.begin, chis[r], True, cscale, n_elec, vexts[r], vs[r])
        gs = self._stages()
        out = []
        for r, s in enumerate(self.st):
            E, vn = s.energies(gs)
            mu = vn / n_elec
            out.append(self._timed(r, s.chi_grad, chis[r], vs[r], cscale, mu))
        return E, mu, torch.cat(out)

    def close(self):
        for s in self.st:
            s.close()
