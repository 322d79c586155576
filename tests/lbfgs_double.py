"""TEST DOUBLE (test infrastructure only): numpy implementation of the backend contract of
professad_amd.optimize.VectorFreeLBFGS (the semantics of ofdft_lbfgs_dots / _commit / _update in include/ofdft_hip.h).
Used to test the host-side recursion on CPU and, on the GPU, as the checker of the HIP sweeps."""
import numpy as np


class NumpyLbfgsBackend:
    def __init__(self, n, history):
        self.n, self.m = n, history
        self.S, self.Y = [], []
        self.d = np.zeros(n)
        self.g_prev = np.zeros(n)
        self.t_prev = 0.0
        self.have_prev = False
        self.cand = None

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy().reshape(-1) if hasattr(t, 'detach') else np.asarray(t).reshape(-1)

    def dots(self, g):
        g = self._np(g)
        if self.have_prev:
            y, s = g - self.g_prev, self.t_prev * self.d
        else:
            y, s = np.zeros(self.n), np.zeros(self.n)
        out = []
        for V in (self.S, self.Y):
            for vj in V:
                out += [s @ vj, y @ vj, g @ vj]
        out += [s @ s, s @ y, y @ y, g @ s, g @ y, g @ g, np.abs(g).sum()]
        self.cand = (s, y) if self.have_prev else None
        return np.array(out, dtype=np.float64), len(self.S)

    def commit(self, push):
        if push:
            assert self.cand is not None
            if len(self.S) == self.m:
                self.S.pop(0)
                self.Y.pop(0)
            self.S.append(self.cand[0].copy())
            self.Y.append(self.cand[1].copy())
        self.cand = None

    def update(self, cs, cy, cg, t, x, g):
        gn = self._np(g)
        d = cg * gn
        for j in range(len(self.S)):
            d = d + cs[j] * self.S[j] + cy[j] * self.Y[j]
        self.d = d
        if hasattr(x, 'detach'):
            import torch
            x.add_(torch.as_tensor(t * d, dtype=x.dtype, device=x.device).view_as(x))
        else:
            x += t * d
        self.g_prev = gn.copy()
        self.t_prev = t
        self.have_prev = True
        return float(np.abs(t * d).sum())

    def reset(self):
        self.__init__(self.n, self.m)
