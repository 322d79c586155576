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

`VectorFreeLBFGS` is the same algorithm arranged for the GPU: the history lives behind the
`ofdft_lbfgs_*` C ABI and every inner iteration is ONE multi-dot sweep (all inner products of {s, y, g} with the
stored pairs) plus ONE multi-axpy sweep (direction, chi update, gradient copy) -- about 40 grid-sized reads/writes
instead of the ~135 of the op-by-op form -- while the two-loop recursion itself runs here on the (2m+1) coefficients
of the direction in the basis {S_j, Y_j, g}.  On several ranks the sweep's local sums need one small all-reduce.
"""
import ctypes as C
import math

import numpy as np
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


class HipLbfgsBackend:
    """History and sweeps on the device (ofdft_lbfgs_* in libofdft_hip.so; no CPU fallback)."""

    def __init__(self, n, history, device, dtype=torch.double):
        from . import _native as N
        self.dtype = dtype
        self.lib = N.load(N.F64 if dtype == torch.double else N.F32)      # the vectors take the library's precision
        self.device = torch.device(device)
        # the device index is resolved exactly as Engine does: a bare 'cuda' means torch's current device
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device('cuda', idx)
        self._h = C.c_void_p(0)
        rc = self.lib.ofdft_lbfgs_create(C.byref(self._h), int(n), int(history), idx)
        if rc != 0:
            raise RuntimeError('ofdft_lbfgs_create failed with code %d' % rc)
        self._buf = (C.c_double * (6 * 8 + 7))()
        self._cs, self._cy, self._cg = (C.c_double * 8)(), (C.c_double * 8)(), 0.0
        self.n = int(n)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s: %s' % (what, self.lib.ofdft_lbfgs_last_error(self._h).decode()))

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _vec(self, t, name):
        if not (t.is_cuda and t.dtype == self.dtype and t.is_contiguous() and t.numel() == self.n):
            raise ValueError('%s must be a contiguous %s device tensor of %d elements' % (name, self.dtype, self.n))
        return C.c_void_p(t.data_ptr())

    def dots(self, g):
        k = C.c_int(0)
        self._check(self.lib.ofdft_lbfgs_dots(self._h, self._vec(g, 'g'), self._buf, C.byref(k), self._stream()), 'ofdft_lbfgs_dots')
        return np.array(self._buf[:6 * k.value + 7], dtype=np.float64), k.value

    def dots_raw(self, g):
        """the sums left in the backend's own buffer (for `direction`): -> (number of pairs, tail = s.s, s.y, y.y, g.s, g.y, g.g, |g|_1)"""
        k = C.c_int(0)
        self._check(self.lib.ofdft_lbfgs_dots(self._h, self._vec(g, 'g'), self._buf, C.byref(k), self._stream()), 'ofdft_lbfgs_dots')
        kv = k.value
        return kv, self._buf[6 * kv:6 * kv + 7]

    def set_dots(self, vals):
        """put all-reduced sums back into the buffer `direction` reads"""
        for i, v in enumerate(vals):
            self._buf[i] = v

    def direction(self, k, first):
        """curvature test + commit + Gram blocks + two-loop recursion, natively (ofdft_lbfgs_direction) -> g.d; the coefficients
        stay in the backend for `update_direct`"""
        cg, gtd = C.c_double(0.0), C.c_double(0.0)
        self._check(self.lib.ofdft_lbfgs_direction(self._h, self._buf, int(k), 1 if first else 0, self._cs, self._cy, C.byref(cg),
                                                   C.byref(gtd), None, None), 'ofdft_lbfgs_direction')
        self._cg = cg.value
        return gtd.value

    def update_direct(self, t, x, g):
        """x += t d with the coefficients of the last `direction`; nothing waits for the stream (`abs_step` reads the sum later)"""
        self._check(self.lib.ofdft_lbfgs_update(self._h, self._cs, self._cy, self._cg, float(t), self._vec(x, 'x'), self._vec(g, 'g'),
                                                None, self._stream()), 'ofdft_lbfgs_update')

    def abs_step(self):
        out = C.c_double(0.0)
        self._check(self.lib.ofdft_lbfgs_abs_step(self._h, C.byref(out)), 'ofdft_lbfgs_abs_step')
        return out.value

    def commit(self, push):
        self._check(self.lib.ofdft_lbfgs_commit(self._h, 1 if push else 0), 'ofdft_lbfgs_commit')

    def update(self, cs, cy, cg, t, x, g):
        dp = C.POINTER(C.c_double)
        cs = np.ascontiguousarray(cs, dtype=np.float64)
        cy = np.ascontiguousarray(cy, dtype=np.float64)
        out = C.c_double(0.0)
        self._check(self.lib.ofdft_lbfgs_update(self._h, cs.ctypes.data_as(dp), cy.ctypes.data_as(dp), float(cg), float(t),
                                                self._vec(x, 'x'), self._vec(g, 'g'), C.byref(out), self._stream()),
                    'ofdft_lbfgs_update')
        return out.value

    def reset(self):
        self._check(self.lib.ofdft_lbfgs_reset(self._h), 'ofdft_lbfgs_reset')

    def close(self):
        if self._h:
            self.lib.ofdft_lbfgs_destroy(self._h)
            self._h = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class VectorFreeLBFGS:
    """FixedStepLBFGS with the vectors behind a backend (`dots`, `commit`, `update`) and the recursion on coefficients.

    Gram blocks kept on the host for the stored pairs (oldest first): SS[i,j] = S_i.S_j, SY[i,j] = S_i.Y_j,
    YY[i,j] = Y_i.Y_j.  `all_reduce(vec) -> vec` sums the sweeps' local scalars over ranks (None on one GPU)."""

    def __init__(self, x, backend, lr=0.1, history_size=8, max_iter=6, tolerance_grad=1e-5, tolerance_change=1e-9,
                 all_reduce=None):
        self.x, self.b = x, backend
        self.lr, self.m, self.max_iter = float(lr), int(history_size), int(max_iter)
        self.max_eval = self.max_iter * 5 // 4       # lbfgsnew.py:69-70
        self.tol_g, self.tol_c = tolerance_grad, tolerance_change
        self.all_reduce = all_reduce
        self.SS = np.zeros((0, 0))
        self.SY = np.zeros((0, 0))
        self.YY = np.zeros((0, 0))
        self.gamma = 1.0
        self.total_iter = 0
        self.func_evals = 0
        self.info = None

    # ---- sweep 1: everything the iteration needs to know about the new gradient
    def _analyse(self, g):
        vals, k = self.b.dots(g.view(-1))
        if self.all_reduce is not None:
            vals = self.all_reduce(vals)
        v = vals[:6 * k].reshape(2, k, 3)        # [S|Y][j][s,y,g]
        tail = vals[6 * k:]
        self.info = dict(k=k, sS=v[0, :, 0], yS=v[0, :, 1], gS=v[0, :, 2], sY=v[1, :, 0], yY=v[1, :, 1], gY=v[1, :, 2],
                         ss=tail[0], sy=tail[1], yy=tail[2], gs=tail[3], gy=tail[4], gg=tail[5], g1=tail[6])
        return self.info

    def _push(self, f):
        """append the candidate pair's row / column to the Gram blocks (dropping the oldest pair when full)"""
        k = f['k']
        sS, yS, sY, yY, gS, gY = f['sS'], f['yS'], f['sY'], f['yY'], f['gS'], f['gY']
        SS, SY, YY = self.SS, self.SY, self.YY
        if k == self.m:
            SS, SY, YY = SS[1:, 1:], SY[1:, 1:], YY[1:, 1:]
            sS, yS, sY, yY, gS, gY = sS[1:], yS[1:], sY[1:], yY[1:], gS[1:], gY[1:]
            k -= 1
        n = k + 1
        nSS, nSY, nYY = np.zeros((n, n)), np.zeros((n, n)), np.zeros((n, n))
        nSS[:k, :k], nSY[:k, :k], nYY[:k, :k] = SS, SY, YY
        nSS[k, :k] = nSS[:k, k] = sS
        nSS[k, k] = f['ss']
        nYY[k, :k] = nYY[:k, k] = yY
        nYY[k, k] = f['yy']
        nSY[k, :k] = sY              # s_new . Y_j
        nSY[:k, k] = yS              # S_j . y_new
        nSY[k, k] = f['sy']
        self.SS, self.SY, self.YY = nSS, nSY, nYY
        return np.append(gS, f['gs']), np.append(gY, f['gy'])

    def _coefficients(self, gS, gY, gg):
        """two-loop recursion (lbfgsnew.py:649-663) on the coefficients of the direction in {S_j, Y_j, g}"""
        k = self.SS.shape[0]
        dS, dY, dg = np.zeros(k), np.zeros(k), -1.0
        rho = 1.0 / np.diag(self.SY) if k else np.zeros(0)
        al = np.zeros(k)
        for i in range(k - 1, -1, -1):
            al[i] = (dS @ self.SS[:, i] + dY @ self.SY[i, :] + dg * gS[i]) * rho[i]
            dY[i] -= al[i]
        dS *= self.gamma
        dY *= self.gamma
        dg *= self.gamma
        for i in range(k):
            be = (dS @ self.SY[:, i] + dY @ self.YY[:, i] + dg * gY[i]) * rho[i]
            dS[i] += al[i] - be
        gtd = dS @ gS + dY @ gY + dg * gg
        return dS, dY, dg, gtd

    def step(self, closure):
        """Same contract as FixedStepLBFGS.step."""
        if hasattr(self.b, 'direction'):
            return self._step_native(closure)
        loss0, g = closure()
        loss = loss0
        evals = 1
        self.func_evals += 1
        f = self._analyse(g)
        g1 = f['g1']
        if g1 <= self.tol_g:
            return loss0
        n_iter = 0
        while n_iter < self.max_iter and not math.isnan(f['gg']):
            n_iter += 1
            self.total_iter += 1
            if self.total_iter == 1:
                self.b.commit(False)
                self.SS = self.SY = self.YY = np.zeros((0, 0))
                self.gamma = 1.0
                gS, gY = np.zeros(0), np.zeros(0)
            else:
                push = f['sy'] > 1e-10 * f['ss']                      # lbfgsnew.py:622 (sn*sn = s.s)
                self.b.commit(push)
                if push:
                    gS, gY = self._push(f)
                    self.gamma = f['sy'] / f['yy']
                else:
                    gS, gY = f['gS'], f['gY']
            cs, cy, cg, gtd = self._coefficients(gS, gY, f['gg'])
            prev_loss = loss
            t = min(1.0, 1.0 / g1) * self.lr if self.total_iter == 1 else self.lr
            abs_step = self.b.update(cs, cy, cg, t, self.x.view(-1), g.view(-1))       # x += t d, g_prev = g
            if self.all_reduce is not None:
                abs_step = float(self.all_reduce(np.array([abs_step]))[0])
            if n_iter != self.max_iter:
                loss, g = closure()
                f = self._analyse(g)
                g1 = f['g1']
                evals += 1
                self.func_evals += 1
                if math.isnan(g1):
                    break
            if n_iter == self.max_iter or evals >= self.max_eval:
                break
            if g1 <= self.tol_g or gtd > -self.tol_c:
                break
            if abs_step <= self.tol_c or abs(loss - prev_loss) < self.tol_c:
                break
        return loss0

    # ---- the same iteration with the host side of the recursion inside the library (ofdft_lbfgs_direction): per inner
    # iteration two ctypes calls and no numpy between the sweeps; the update does not wait for its stream (the closure
    # evaluation that follows synchronises, the |step| sum is read after it)
    def _dots_native(self, g):
        k, tail = self.b.dots_raw(g.view(-1))
        if self.all_reduce is not None:
            vals = self.all_reduce(np.array(self.b._buf[:6 * k + 7], dtype=np.float64))
            self.b.set_dots(vals)
            tail = [float(v) for v in vals[6 * k:]]
        return k, tail           # tail = s.s, s.y, y.y, g.s, g.y, g.g, |g|_1

    def _step_native(self, closure):
        loss0, g = closure()
        loss = loss0
        evals = 1
        self.func_evals += 1
        k, tail = self._dots_native(g)
        g1 = tail[6]
        if g1 <= self.tol_g:
            return loss0
        n_iter = 0
        while n_iter < self.max_iter and not math.isnan(tail[5]):
            n_iter += 1
            self.total_iter += 1
            gtd = self.b.direction(k, self.total_iter == 1)
            prev_loss = loss
            t = min(1.0, 1.0 / g1) * self.lr if self.total_iter == 1 else self.lr
            self.b.update_direct(t, self.x.view(-1), g.view(-1))                       # x += t d, g_prev = g
            last = n_iter == self.max_iter
            if not last:
                loss, g = closure()                                                   # (synchronises the stream)
                k, tail = self._dots_native(g)
                g1 = tail[6]
                evals += 1
                self.func_evals += 1
                if math.isnan(g1):
                    break
            if last or evals >= self.max_eval:
                break
            if g1 <= self.tol_g or gtd > -self.tol_c:
                break
            abs_step = self.b.abs_step()
            if self.all_reduce is not None:
                abs_step = float(self.all_reduce(np.array([abs_step]))[0])
            if abs_step <= self.tol_c or abs(loss - prev_loss) < self.tol_c:
                break
        return loss0


class TwoPointGradientDescent:
    """Barzilai-Borwein two-point gradient descent on one flat variable -- the reference's alternative density optimiser
    (`n_method='TPGD'`, system.py:823-824; _optimizers/tpgd/two_point_gradient_descent.py:25-65): one closure call per step,
    step length (dx . dx) / (dx . dg) from the previous point, the fixed `lr` on the first step or when the quotient is not
    positive.  `all_reduce(vec) -> vec` sums the two local dot products over ranks (None on one GPU)."""

    def __init__(self, x, lr=0.1, all_reduce=None):
        if lr <= 0.0:
            raise ValueError('lr must be positive')
        self.x, self.lr, self.all_reduce = x, float(lr), all_reduce
        self.x_prev = self.g_prev = None
        self.func_evals = 0
        self.total_iter = 0

    def step(self, closure):
        loss, g = closure()
        self.func_evals += 1
        alpha = self.lr
        if self.x_prev is not None:
            dx, dg = self.x - self.x_prev, g - self.g_prev
            dots = torch.stack(((dx * dx).sum(dtype=torch.double), (dx * dg).sum(dtype=torch.double))).cpu().numpy()
            if self.all_reduce is not None:
                dots = self.all_reduce(dots)
            if dots[1] != 0.0 and dots[0] / dots[1] > 0.0:
                alpha = float(dots[0] / dots[1])
        self.x_prev, self.g_prev = self.x.clone(), g.clone()
        self.x.add_(g, alpha=-alpha)
        self.total_iter += 1
        return loss


def optimize_density(engine, n_elec, vext=None, chi0=None, ntol=1e-7, n_conv_cond_count=3, n_step_size=0.1,
                     n_maxiter=1000, conv_target='dE', verbose=False, volume=None, optimizer='fused', n_method='LBFGS'):
    """Minimise E[n = N_e chi^2 / int chi^2] with the engine's terms.  Returns a dict with the converged density,
    chi, energy [Ha], iteration count and the convergence history.

    engine: an `Engine` (cell and terms already set) or a `DistEngine` (slab-decomposed: chi / vext / the results are
    this rank's x-slabs and every rank runs this function; the optimiser's scalars are all-reduced, so all ranks take
    identical decisions); `volume` = cell volume (needed when chi0 is None to start from the uniform density, as
    System.optimize_density does after System.__init__).  `optimizer`: 'fused' (HIP sweeps, `VectorFreeLBFGS`) or
    'torch' (`FixedStepLBFGS` on torch tensors; single GPU only).
    With an fp32 engine the energies carry ~1e-6 relative round-off: choose `ntol` above that noise (the default 1e-7 eV
    suits fp64) or the loop runs to `n_maxiter`."""
    if conv_target not in ('dE', 'dEdchi', 'euler'):
        raise ValueError("Only 'dE', 'dEdchi' or 'euler' recognized as 'conv_target' argument")       # system.py:889-890
    if n_method not in ('LBFGS', 'TPGD'):
        raise ValueError("Only 'LBFGS' or 'TPGD' recognized for 'n_method' argument")                 # system.py:826
    if conv_target != 'dE' and volume is None:
        raise ValueError("conv_target=%r needs `volume` (the cell volume fixes dV)" % conv_target)
    comm = getattr(engine, 'comm', None)                    # DistEngine
    multi = comm is not None and comm.active
    if comm is not None:
        shape, dev, dtype = engine.plan.local_shape, engine.stages.device, engine.stages.dtype
        npts_global = engine.npts_global
    else:
        shape, dev, dtype = engine.shape, engine.device, engine.dtype
        npts_global = int(np.prod(shape))

    def gsum(x):                                            # sum over ranks of a python float
        return float(comm.all_reduce_sum(np.array([x]), dev)[0]) if multi else x

    def gmax(x):
        if not multi:
            return x
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.double, device=dev if comm.backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=comm.group)
        return float(t[0])
    if chi0 is None:
        if volume is None:
            raise ValueError('volume is needed for the uniform start')
        chi = torch.full(shape, math.sqrt(n_elec / volume), dtype=dtype, device=dev)
    else:
        chi = chi0.detach().clone().to(device=dev, dtype=dtype)
    state = {}

    def closure():
        E_terms, mu, g = engine.energy_grad_chi(chi, n_elec, vext)
        state.update(E=sum(E_terms.values()), mu=mu, g=g, E_terms=E_terms)
        if conv_target == 'euler':          # the Euler residual belongs to the density of THIS call (see below): keep its chi
            if 'chi_c' not in state:
                state['chi_c'] = torch.empty_like(chi)
            state['chi_c'].copy_(chi)
        return state['E'], g

    if n_method == 'TPGD':          # system.py:823-824
        hook = (lambda v: comm.all_reduce_sum(np.ascontiguousarray(v, dtype=np.float64), dev)) if multi else None
        opt = TwoPointGradientDescent(chi, lr=n_step_size, all_reduce=hook)
    elif optimizer == 'fused':      # device-resident history, two sweeps per inner iteration
        hook = (lambda v: comm.all_reduce_sum(np.ascontiguousarray(v, dtype=np.float64), dev)) if multi else None
        opt = VectorFreeLBFGS(chi, HipLbfgsBackend(chi.numel(), 8, dev, dtype), lr=n_step_size, history_size=8, max_iter=6,
                              all_reduce=hook)
    elif optimizer == 'torch' and not multi:      # op-by-op form on torch tensors
        opt = FixedStepLBFGS(chi, lr=n_step_size, history_size=8, max_iter=6)
    else:
        raise ValueError("optimizer must be 'fused' or 'torch'")
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
            dV = volume / npts_global
        dEdchi = gmax(float(state['g'].abs().max())) / dV if dV else float('nan')
        history.append((it, E, dE, dEdchi))
        if verbose:
            print('%5d %16.8f %12.4e %12.4e' % history[-1])
        if conv_target == 'euler':
            # max |mu - dE/dn| (system.py:377-412) at the density of the step's LAST closure call -- the reference's System keeps
            # that density (system.py:833-834) and check_density_convergence differentiates at it, not at the chi the optimiser
            # moved to afterwards.  No extra evaluation: the closure's gradient is chi.grad = 2 c chi (dE/dn - mu) dV
            # (system.py:850-853), so mu - dE/dn = -chi.grad / (2 c chi dV) wherever chi != 0.
            chi_c = state['chi_c'].double()
            c2 = n_elec / (gsum(float((chi_c * chi_c).sum())) / npts_global * volume)
            res = torch.where(chi_c != 0, state['g'].double() / (2.0 * c2 * dV * chi_c), torch.zeros_like(chi_c))
            stop = gmax(float(res.abs().max()))
        else:
            stop = abs(dE) if conv_target == 'dE' else dEdchi
        if it > 5:
            conv = conv + 1 if stop < ntol else 0
        if conv == n_conv_cond_count:
            break
    E_terms, mu, g = engine.energy_grad_chi(chi, n_elec, vext)
    ntilde = gsum(float((chi.double() * chi.double()).sum())) / npts_global * (volume if volume is not None else 1.0)
    den = (n_elec / ntilde) * chi * chi if volume is not None else None
    return dict(chi=chi, den=den, E_Ha=sum(E_terms.values()), E_terms=E_terms, mu=mu, iterations=it,
                converged=conv == n_conv_cond_count, history=history, func_evals=opt.func_evals)
