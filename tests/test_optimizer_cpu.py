"""The from-scratch fixed-step L-BFGS against the converged reference state, driven with the oracle closure on CPU
(no GPU, no engine): pins the optimiser's semantics -- it reproduces the reference's own optimisation log."""
import math
import os

import numpy as np
import torch

import cases
from oracle import refpath as rp
from professad_amd.optimize import EV_PER_HA, FixedStepLBFGS

GOLDEN = os.path.dirname(os.path.abspath(cases.__file__))


def test_lbfgs_reproduces_reference_log_with_oracle_closure():
    d = np.load(os.path.join(GOLDEN, 'cfg1_fccAl_32.npz'))
    box, vext, n_elec = torch.as_tensor(d['box']), torch.as_tensor(d['vext']), float(d['n_elec'])
    vol = float(torch.abs(torch.linalg.det(box)))
    tab = rp.term_table(vext)
    fns = [tab[k] for k in ('ion_electron', 'hartree', 'tf', 'vw', 'lda_x', 'pz_c')]
    chi = torch.full((32, 32, 32), math.sqrt(n_elec / vol), dtype=torch.double)
    last = {}

    def closure():
        E, g = rp.closure(box, chi, n_elec, fns)
        last['E'] = float(E)
        return float(E), g

    opt = FixedStepLBFGS(chi, lr=0.1, history_size=8, max_iter=6)
    assert abs(closure()[0] * EV_PER_HA - 72.397190) < 1e-6          # row 0 of the reference's table
    want = [68.191536, 65.989145, 65.547128, 65.459420]               # rows 1-4 (reference log in the fixture)
    for w in want:
        opt.step(closure)
        assert abs(last['E'] * EV_PER_HA - w) < 2e-6


def _test_problem(n=400, seed=3):
    """a smooth non-quadratic test function with an ill-conditioned quadratic part"""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n)) / math.sqrt(n)
    H = torch.as_tensor(A @ A.T + np.diag(np.linspace(0.05, 3.0, n)))
    b = torch.as_tensor(rng.standard_normal(n))

    def make(x):
        def closure():
            xx = x.detach().clone().requires_grad_(True)
            f = 0.5 * xx @ (H @ xx) - b @ xx + 0.05 * (xx ** 4).sum()
            f.backward()
            return float(f.detach()), xx.grad.detach()
        return closure
    return make


def test_vector_free_lbfgs_equals_two_loop_form():
    """the coefficient-space recursion + sweeps (numpy double of the HIP backend) walk the same path as the
    op-by-op two-loop form, including history wrap-around, rejected pairs and the step's break conditions"""
    from lbfgs_double import NumpyLbfgsBackend
    from professad_amd.optimize import VectorFreeLBFGS
    n = 400
    make = _test_problem(n)
    xa = torch.full((n,), 0.3, dtype=torch.double)
    xb = xa.clone()
    a = FixedStepLBFGS(xa, lr=0.1, history_size=8, max_iter=6)
    b = VectorFreeLBFGS(xb, NumpyLbfgsBackend(n, 8), lr=0.1, history_size=8, max_iter=6)
    ca, cb = make(xa), make(xb)
    for it in range(12):          # 12 outer steps = up to 72 inner iterations: the 8-pair history wraps several times
        la, lb = a.step(ca), b.step(cb)
        assert abs(la - lb) <= 1e-9 * max(1.0, abs(la)), (it, la, lb)
        assert float((xa - xb).abs().max()) <= 1e-8 * float(xa.abs().max()), it
    assert a.func_evals == b.func_evals and a.total_iter == b.total_iter


def test_vector_free_lbfgs_sharded_sums():
    """two 'ranks' each holding half of the vector, local sums added by the all_reduce hook: same path"""
    from lbfgs_double import NumpyLbfgsBackend
    from professad_amd.optimize import VectorFreeLBFGS
    n = 400
    make = _test_problem(n)
    xa = torch.full((n,), 0.3, dtype=torch.double)
    xb = xa.clone()
    a = VectorFreeLBFGS(xa, NumpyLbfgsBackend(n, 8), lr=0.1, history_size=8, max_iter=6)

    class Halves:
        """backend over two shards; returns rank 0's local sums, the hook adds rank 1's"""
        def __init__(self):
            self.h = [NumpyLbfgsBackend(n // 2, 8), NumpyLbfgsBackend(n // 2, 8)]
            self.other = None

        def dots(self, g):
            v0, k = self.h[0].dots(g[:n // 2])
            self.other, _ = self.h[1].dots(g[n // 2:])
            return v0, k

        def commit(self, push):
            for h in self.h:
                h.commit(push)

        def update(self, cs, cy, cg, t, x, g):
            s0 = self.h[0].update(cs, cy, cg, t, x[:n // 2], g[:n // 2])
            self.other = np.array([self.h[1].update(cs, cy, cg, t, x[n // 2:], g[n // 2:])])
            return s0

    hb = Halves()
    b = VectorFreeLBFGS(xb, hb, lr=0.1, history_size=8, max_iter=6, all_reduce=lambda v: v + hb.other)
    ca, cb = make(xa), make(xb)
    for it in range(6):
        la, lb = a.step(ca), b.step(cb)
        assert abs(la - lb) <= 1e-10 * max(1.0, abs(la))
    assert float((xa - xb).abs().max()) <= 1e-9 * float(xa.abs().max())


def test_two_point_gradient_descent_minimises_the_test_function():
    """the reference's alternative optimiser (Barzilai-Borwein steps, one closure call per step): first step = lr * gradient,
    later steps (dx.dx)/(dx.dg); reaches the minimum the L-BFGS reaches"""
    from professad_amd.optimize import FixedStepLBFGS, TwoPointGradientDescent
    make = _test_problem()
    x = torch.zeros(400, dtype=torch.double)
    opt = TwoPointGradientDescent(x, lr=0.1)
    f0, g0 = make(x)()
    opt.step(make(x))
    assert torch.allclose(x, -0.1 * g0)                      # first step: fixed length
    for _ in range(400):
        opt.step(make(x))
    y = torch.zeros(400, dtype=torch.double)
    ref = FixedStepLBFGS(y, lr=0.1, history_size=8, max_iter=6)
    for _ in range(80):
        ref.step(make(y))
    assert abs(make(x)()[0] - make(y)()[0]) < 1e-9 and opt.func_evals == 401
