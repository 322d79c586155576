"""Density optimisation on top of the engine's closure: the caller side of the hot path.

`optimize_density` mirrors the control flow and stopping rules of the reference's
`System.optimize_density` (src/professad/system.py:774-908) for its default method: chi = sqrt(n) as the
unconstrained variable, a fixed-step limited-memory BFGS (history 8, at most 6 inner iterations / 7 closure calls
per outer step, step 0.1; the reference instantiates its vendored optimiser that way, system.py:822), energy
differences in eV against `ntol`, convergence only counted after the 5th outer iteration.

`FixedStepLBFGS` is a from-scratch implementation of that algorithm (standard two-loop recursion with the
y.s > 1e-10 |s|^2 curvature test and gamma = y.s / y.y scaling, reference semantics
_optimizers/lbfgs/lbfgsnew.py:512-769) that keeps the (s, y) history as two [m, N] device matrices so the
two-loop dot products and updates are batched tensor ops on the GPU.
"""
import math

import torch

EV_PER_HA = 4.3597447222071e-18 / 1.602176634e-19      # system.py:27-33


class FixedStepLBFGS:
    def __init__(self, x, lr=0.1, history_size=8, max_iter=6, tolerance_grad=1e-5, tolerance_change=1e-9):
        self.x = x                                   # flat view is taken lazily; updated in place
        self.lr, self.m, self.max_iter = float(lr), int(history_size), int(max_iter)
        self.max_eval = self.max_iter * 5 // 4       # lbfgsnew.py:69-70
        self.tol_g, self.tol_c = tolerance_grad, tolerance_change
        n = x.numel()
        self.S = torch.zeros(self.m, n, dtype=x.dtype, device=x.device)
        self.Y = torch.zeros(self.m, n, dtype=x.dtype, device=x.device)
        self.count = 0                                # valid pairs (oldest first in rows 0..count-1)
        self.gamma = 1.0
        self.total_iter = 0
        self.d = None
        self.t = None
        self.g_prev = None
        self.func_evals = 0

    def _push(self, s, y):
        if self.count == self.m:
            self.S = torch.roll(self.S, -1, 0)
            self.Y = torch.roll(self.Y, -1, 0)
            self.count -= 1
        self.S[self.count].copy_(s)
        self.Y[self.count].copy_(y)
        self.count += 1

    def _direction(self, g):
        """-H g by the two-loop recursion over the stored pairs."""
        k = self.count
        q = g.neg()
        if k == 0:
            return q * self.gamma
        S, Y = self.S[:k], self.Y[:k]
        rho = 1.0 / (S * Y).sum(1)
        al = torch.empty(k, dtype=g.dtype, device=g.device)
        for i in range(k - 1, -1, -1):
            al[i] = torch.dot(S[i], q) * rho[i]
            q.add_(Y[i], alpha=-float(al[i]))
        r = q * self.gamma
        for i in range(k):
            be = torch.dot(Y[i], r) * rho[i]
            r.add_(S[i], alpha=float(al[i] - be))
        return r

    def step(self, closure):
        """One outer step: closure() -> (loss float, flat gradient tensor); x is updated in place.
        Returns the loss of the first closure call (as the reference's optimiser does)."""
        x = self.x.view(-1)
        loss0, g = closure()
        loss = loss0
        evals = 1
        self.func_evals += 1
        g = g.reshape(-1)
        g1 = float(g.abs().sum())
        if g1 <= self.tol_g:
            return loss0
        n_iter = 0
        while n_iter < self.max_iter and not math.isnan(float(g.norm())):
            n_iter += 1
            self.total_iter += 1
            if self.total_iter == 1:
                d = g.neg()
                self.count = 0
                self.gamma = 1.0
            else:
                y = g - self.g_prev
                s = self.d * self.t
                ys = float(torch.dot(y, s))
                sn = float(s.norm())
                if ys > 1e-10 * sn * sn:                          # lbfgsnew.py:622
                    self._push(s, y)
                    self.gamma = ys / float(torch.dot(y, y))
                d = self._direction(g)
            self.g_prev = g.clone()
            prev_loss = loss
            t = min(1.0, 1.0 / g1) * self.lr if self.total_iter == 1 else self.lr
            gtd = float(torch.dot(g, d))
            x.add_(d, alpha=t)                                     # fixed step, no line search
            self.d, self.t = d, t
            if n_iter != self.max_iter:
                loss, g = closure()
                g = g.reshape(-1)
                g1 = float(g.abs().sum())
                evals += 1
                self.func_evals += 1
                if math.isnan(g1):
                    break
            if n_iter == self.max_iter or evals >= self.max_eval:
                break
            if g1 <= self.tol_g or gtd > -self.tol_c:
                break
            if float((d * t).abs().sum()) <= self.tol_c or abs(loss - prev_loss) < self.tol_c:
                break
        return loss0


def optimize_density(engine, n_elec, vext=None, chi0=None, ntol=1e-7, n_conv_cond_count=3, n_step_size=0.1,
                     n_maxiter=1000, conv_target='dE', verbose=False, volume=None):
    """Minimise E[n = N_e chi^2 / int chi^2] with the engine's terms.  Returns a dict with the converged density,
    chi, energy [Ha], iteration count and the convergence history.

    engine: an `Engine` (cell and terms already set); `volume` = cell volume (needed when chi0 is None to start from
    the uniform density, as System.optimize_density does after System.__init__)."""
    shape, dev = engine.shape, engine.device
    if chi0 is None:
        if volume is None:
            raise ValueError('volume is needed for the uniform start')
        chi = torch.full(shape, math.sqrt(n_elec / volume), dtype=torch.double, device=dev)
    else:
        chi = chi0.detach().clone().to(dev)
    state = {}

    def closure():
        E_terms, mu, g = engine.energy_grad_chi(chi, n_elec, vext)
        state.update(E=sum(E_terms.values()), mu=mu, g=g, E_terms=E_terms)
        return state['E'], g

    opt = FixedStepLBFGS(chi, lr=n_step_size, history_size=8, max_iter=6)
    E_prev = closure()[0] * EV_PER_HA
    history, conv = [], 0
    dV = None
    for it in range(1, int(round(n_maxiter)) + 1):
        opt.step(closure)
        # like the reference, energy / gradient are those of the LAST closure call of the step (system.py:869-871)
        E = state['E'] * EV_PER_HA
        dE = E - E_prev
        E_prev = E
        if dV is None and volume is not None:
            dV = volume / chi.numel()
        dEdchi = float(state['g'].abs().max()) / dV if dV else float('nan')
        history.append((it, E, dE, dEdchi))
        if verbose:
            print('%5d %16.8f %12.4e %12.4e' % history[-1])
        stop = abs(dE) if conv_target == 'dE' else dEdchi
        if it > 5:
            conv = conv + 1 if stop < ntol else 0
        if conv == n_conv_cond_count:
            break
    E_terms, mu, g = engine.energy_grad_chi(chi, n_elec, vext)
    ntilde = float((chi * chi).mean()) * (volume if volume is not None else 1.0)
    den = (n_elec / ntilde) * chi * chi if volume is not None else None
    return dict(chi=chi, den=den, E_Ha=sum(E_terms.values()), E_terms=E_terms, mu=mu, iterations=it,
                converged=conv == n_conv_cond_count, history=history, func_evals=opt.func_evals)
