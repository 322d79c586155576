"""TEST DOUBLE (test infrastructure only): a numpy implementation of the `stages` interface of
professad_amd.distributed for Hartree + Thomas-Fermi, used to exercise the product's host-side multi-rank logic
(SlabPlan, Comm, run_closure / run_potential: stage sequencing, all-to-all semantics, reductions) under gloo on CPU.
It is never imported by the package."""
import math

import numpy as np
import torch

from professad_amd.distributed import NSUMS, SlabPlan

C_TF = 0.3 * (3 * math.pi ** 2) ** (2 / 3)


def _freqs(n):
    i = np.arange(n)
    return np.where(i <= n // 2, i, i - n).astype(np.float64)


class NumpyStages:
    device = 'cpu'

    def __init__(self, shape, box, nranks, rank):
        self.plan = SlabPlan(shape, nranks, rank)
        self.box = np.asarray(box, dtype=np.float64)
        self.vol = abs(np.linalg.det(self.box))
        self.dV = self.vol / np.prod(shape)
        self.k = (0, 0)

    def sumsq(self, x, square=True):
        x = x.numpy()
        return float((x * x).sum() if square else x.sum())

    def begin(self, src, from_chi, cscale, nel, vext, v_out):
        x = src.numpy()
        self.n = cscale * x * x if from_chi else x.copy()
        self.vext = None if vext is None else vext.numpy()
        self.v_out = v_out
        self.k = (0, 0)

    def _bytes(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr).view(np.uint8).reshape(-1))

    nchunks = 1          # kz chunks of the exchange (tests set 1..4): planes [kz0, kz1) of every message travel together

    def _kz(self, k):
        nzc, K = self.plan.nzc, self.nchunks
        return slice(nzc * k // K, nzc * (k + 1) // K)

    def step(self, step, chain, k):
        """the product's step protocol (ofdft_dist_step): steps 1 / 2 send chunk k, steps 3 / 6 consume, 4 / 5 idle here"""
        p = self.plan
        if chain == 1:       # no nonlocal-KEDF physics in the double: a tagged message per chunk, checked on arrival (step 2),
            P = p.nranks     # so that the second chain's exchanges (own process group, Comm.groups[1]) are exercised too
            if step == 1:
                send = np.array([[1000.0 * p.rank + 10.0 * q + k] * 4 for q in range(P)])
                self.recv1 = getattr(self, 'recv1', {})
                self.recv1[k] = torch.empty(send.nbytes, dtype=torch.uint8)
                return self._bytes(send), self.recv1[k]
            if step == 2:
                got = self.recv1.pop(k).numpy().view(np.float64).reshape(P, 4)
                want = np.array([[1000.0 * q + 10.0 * p.rank + k] * 4 for q in range(P)])
                assert np.array_equal(got, want), (got, want)
                self.chain1_ok = getattr(self, 'chain1_ok', 0) + 1
            return None
        K = self.nchunks
        assert (step, k) == ((self.k[0], self.k[1] + 1) if self.k[1] + 1 < K and self.k[0] > 0 else (self.k[0] + 1, 0)), (step, k, self.k)
        self.k = (step, k)
        P = p.nranks
        kz = self._kz(k)
        nz = kz.stop - kz.start
        if step == 1:      # z, y transforms on the x-slab (with chunk 0); chunk for peer q = its y range, planes of chunk k
            if k == 0:
                self.nk = np.fft.fft(np.fft.rfft(self.n, axis=2), axis=1)
                self.recv = {}
            send = np.stack([self.nk[:, p.y_range(q), kz] for q in range(P)])
            self.recv[(1, k)] = torch.empty(send.nbytes, dtype=torch.uint8)
            return self._bytes(send), self.recv[(1, k)]
        if step == 2:      # x transform + Coulomb kernel on the y-slab, planes of chunk k; chunk for peer q = its x range
            got = self.recv.pop((1, k)).numpy().view(np.complex128).reshape(P, p.nxl, p.nyl, nz)
            full = np.concatenate(list(got), axis=0)                      # [n0][nyl][nz]
            fk = np.fft.fft(full, axis=0)
            b = 2 * math.pi * np.linalg.inv(self.box.T)
            n0, n1, n2 = p.shape
            ja, jb, jc = np.meshgrid(_freqs(n0), _freqs(n1)[p.y_range()], np.arange(p.nzc, dtype=np.float64)[kz], indexing='ij')
            kv = [ja * b[0, c] + jb * b[1, c] + jc * b[2, c] for c in range(3)]
            k2 = kv[0] ** 2 + kv[1] ** 2 + kv[2] ** 2
            with np.errstate(divide='ignore'):
                green = np.where(k2 != 0, 4 * math.pi / k2, 0.0)
            res = np.fft.ifft(fk * green, axis=0)
            send = np.stack([res[p.x_range(q)] for q in range(P)])
            self.recv[(2, k)] = torch.empty(send.nbytes, dtype=torch.uint8)
            return self._bytes(send), self.recv[(2, k)]
        if step == 3:      # back on the x-slab: inverse y of the chunk's planes
            got = self.recv.pop((2, k)).numpy().view(np.complex128).reshape(P, p.nxl, p.nyl, nz)
            if k == 0:
                self.vk = np.empty((p.nxl, p.shape[1], p.nzc), dtype=np.complex128)
            self.vk[:, :, kz] = np.fft.ifft(np.concatenate(list(got), axis=1), axis=1)
            return None
        if step == 4 and k == 0:      # whole rows: inverse z
            self.vh = np.fft.irfft(self.vk, n=p.shape[2], axis=2)
        return None

    def finish(self):
        n = self.n
        v = self.vh + (5 / 3) * C_TF * n ** (2 / 3)
        s = np.zeros(NSUMS)
        s[1] = 0.5 * (n * self.vh).sum()
        s[2] = C_TF * (n ** (5 / 3)).sum()
        if self.vext is not None:
            s[0] = (n * self.vext).sum()
            v = v + self.vext
        s[8] = (v * n).sum()
        self.v_out.copy_(torch.from_numpy(v))
        return s

    def energies(self, g):
        return {'ion_electron': g[0] * self.dV, 'hartree': g[1] * self.dV, 'tf': g[2] * self.dV}, g[8] * self.dV

    def chi_grad(self, chi, v, cscale, mu):
        return cscale * 2.0 * chi * (v - mu) * self.dV
