"""Python host object over one ofdft_ctx (one grid shape on one GPU).

PyTorch is used only as plumbing: it owns the device tensors and the HIP stream; every number is
produced by the HIP library behind the C ABI.  There is no CPU path: construction raises when the
library or a GPU is missing.
"""
import ctypes as C

import numpy as np
import torch

from . import _native as N


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Engine:
    """energy + functional derivative of a set of OFDFT terms on a fixed grid shape."""

    def __init__(self, shape, device=None, nranks=1, rank=0, dtype=torch.double):
        """dtype: torch.double (the reference's precision; libofdft_hip.so) or torch.float32 (the fp32 build of the
        same kernels, libofdft_hip_f32.so: hot path and L-BFGS only; energy sums stay fp64)."""
        if dtype not in (torch.double, torch.float32):
            raise TypeError('dtype must be torch.double or torch.float32')
        self.dtype = dtype
        self.cdtype = torch.complex128 if dtype == torch.double else torch.complex64
        self._code = N.F64 if dtype == torch.double else N.F32
        if not torch.cuda.is_available():
            raise N.NativeLibraryError('professad_amd needs a ROCm GPU (torch.cuda.is_available() is False); '
                                       'there is no CPU fallback')
        self.lib = N.load(self._code)
        self.device = torch.device(device if device is not None else 'cuda:0')
        if self.device.type != 'cuda':
            raise ValueError('Engine tensors must live on a GPU device, got %s' % self.device)
        self.global_shape = tuple(int(s) for s in shape)
        if len(self.global_shape) != 3:
            raise ValueError('shape must be (n0, n1, n2)')
        # `shape` of the tensors this engine takes: the rank's x-slab (the whole grid on one GPU)
        self.shape = (self.global_shape[0] // int(nranks),) + self.global_shape[1:]
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._ctx = C.c_void_p(0)
        rc = self.lib.ofdft_create_dist(C.byref(self._ctx), *self.global_shape, self._code, idx, int(nranks), int(rank))
        if rc != 0:
            raise RuntimeError('ofdft_create failed (%d): %s' % (rc, self.lib.ofdft_last_error(None).decode()))
        self._box_key = None
        self._terms_key = None
        self._terms_memo = {}
        self.npts = int(np.prod(self.shape))
        self._dev_index = idx
        self._dev_exact = torch.device('cuda', idx)
        # (a private torch entry point, ~1.5 microseconds cheaper per call than torch.cuda.current_stream(); optional)
        self._raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)

    def close(self):
        if getattr(self, '_ctx', None) is not None and self._ctx.value:
            self.lib.ofdft_destroy(self._ctx)
            self._ctx = C.c_void_p(0)

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    # -- helpers
    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed (%d): %s' % (what, rc, self.lib.ofdft_last_error(self._ctx).decode()))

    def _grid_tensor(self, t, name):
        if t is None:
            return None
        if not isinstance(t, torch.Tensor):
            raise TypeError('%s must be a torch.Tensor' % name)
        if t.device == self._dev_exact and t.dtype == self.dtype and t.shape == self.shape and not t.requires_grad and t.is_contiguous():
            return t                                      # the hot path's case: nothing to convert
        if t.device != self.device and not (t.device.type == 'cuda' and self.device.type == 'cuda'
                                            and (t.device.index or 0) == (self.device.index or 0)):
            raise ValueError('%s is on %s, engine is on %s' % (name, t.device, self.device))
        if t.dtype != self.dtype:
            raise TypeError('%s must be %s for this engine (the reference is fp64 throughout; an fp32 engine takes '
                            'torch.float32 only)' % (name, self.dtype))
        if tuple(t.shape) != self.shape:
            raise ValueError('%s has shape %s, engine grid is %s' % (name, tuple(t.shape), self.shape))
        if t.requires_grad or not t.is_contiguous():
            t = t.detach().contiguous()
        return t

    def _stream(self):
        if self._raw_stream is not None:
            return C.c_void_p(self._raw_stream(self._dev_index))
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- configuration
    def set_cell(self, box_vecs):
        # a host array is compared by its 72 bytes (no device sync; a caller may rescale its box array IN PLACE between
        # calls, so object identity says nothing); anything else goes through torch first
        if isinstance(box_vecs, np.ndarray):
            box = np.ascontiguousarray(box_vecs, dtype=np.float64).reshape(9)
        else:
            box = np.ascontiguousarray(torch.as_tensor(box_vecs).detach().cpu().numpy(), dtype=np.float64).reshape(9)
        key = box.tobytes()
        if key != self._box_key:
            self._check(self.lib.ofdft_set_cell(self._ctx, box.ctypes.data_as(C.POINTER(C.c_double))), 'ofdft_set_cell')
            self._box_key = key
        return self

    def set_terms(self, names, params=None):
        """names: iterable of keys of _native.TERM_BITS; params: dict slot->value
        (wt_alpha, wt_beta, wgc_alpha, wgc_beta, wgc_gamma, wgc_kappa, ggak_kind, ggak_mu, ggak_beta, ggak_lambda, ggak_sigma, vwgtf_kind,
        wts_kind)."""
        memo = None
        if isinstance(names, tuple) and (params is None or isinstance(params, tuple)):      # hashable call (the drop-in terms): memoised
            memo = (names, params)
            key = self._terms_memo.get(memo)
            if key is not None:
                if key != self._terms_key:
                    vals = np.frombuffer(key[1], dtype=np.float64)
                    self._check(self.lib.ofdft_set_terms(self._ctx, key[0], vals.ctypes.data_as(C.POINTER(C.c_double)), len(vals)),
                                'ofdft_set_terms')
                    self._terms_key = key
                return self
            params = dict(params) if params else None
        mask = 0
        for nm in names:
            mask |= N.TERM_BITS[nm]
        slots = ['wt_alpha', 'wt_beta', 'wgc_alpha', 'wgc_beta', 'wgc_gamma', 'wgc_kappa', 'ggak_kind', 'ggak_mu', 'ggak_beta',
                 'ggak_lambda', 'ggak_sigma', 'vwgtf_kind', 'wts_kind']
        s5 = np.sqrt(5.0)
        vals = np.array([5 / 6, 5 / 6, (5 + s5) / 6, (5 - s5) / 6, 2.7, 1.0, 0.0, 40 / 27, 0.0, 0.0, 0.0, 1.0, 0.0], dtype=np.float64)
        for k, v in (params or {}).items():
            vals[slots.index(k)] = float(v)
        key = (mask, vals.tobytes())
        if key != self._terms_key:
            self._check(self.lib.ofdft_set_terms(self._ctx, mask, vals.ctypes.data_as(C.POINTER(C.c_double)), len(vals)),
                        'ofdft_set_terms')
            self._terms_key = key
        if memo is not None:
            self._terms_memo[memo] = key
        return self

    # -- hot path
    def energy_potential(self, den, vext=None, want_potential=True):
        """-> (dict term -> E [Ha], dE/dn tensor or None)."""
        den = self._grid_tensor(den, 'den')
        vext = self._grid_tensor(vext, 'v_ext')
        out = torch.empty_like(den) if want_potential else None
        E = (C.c_double * N.NTERMS)()
        self._check(self.lib.ofdft_energy_potential(self._ctx, _ptr(den), _ptr(vext), E, _ptr(out), self._stream()),
                    'ofdft_energy_potential')
        return {nm: E[i] for i, nm in enumerate(N.TERM_ORDER)}, out

    def energy_grad_chi(self, chi, n_elec, vext=None, want_grad=True):
        """The optimize_density closure: -> (dict term -> E, mu, chi.grad tensor or None)."""
        chi = self._grid_tensor(chi, 'chi')
        vext = self._grid_tensor(vext, 'v_ext')
        out = torch.empty_like(chi) if want_grad else None
        E = (C.c_double * N.NTERMS)()
        mu = C.c_double(0.0)
        self._check(self.lib.ofdft_energy_grad_chi(self._ctx, _ptr(chi), _ptr(vext), float(n_elec), E, C.byref(mu),
                                                   _ptr(out), self._stream()), 'ofdft_energy_grad_chi')
        return dict(zip(N.TERM_ORDER, E[:])), mu.value, out

    # -- validation entry points
    def rfftn(self, x):
        x = self._grid_tensor(x, 'x')
        out = torch.empty(self.shape[0], self.shape[1], self.shape[2] // 2 + 1, dtype=self.cdtype, device=self.device)
        self._check(self.lib.ofdft_rfftn(self._ctx, _ptr(x), _ptr(out), self._stream()), 'ofdft_rfftn')
        return out

    def irfftn(self, xk):
        want = (self.shape[0], self.shape[1], self.shape[2] // 2 + 1)
        if tuple(xk.shape) != want or xk.dtype != self.cdtype:
            raise ValueError('spectrum must be %s of shape %s' % (self.cdtype, want))
        xk = xk.detach().contiguous()
        out = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        self._check(self.lib.ofdft_irfftn(self._ctx, _ptr(xk), _ptr(out), self._stream()), 'ofdft_irfftn')
        return out

    def debug_math(self, kind, x):
        """the lean transcendentals of csrc/fastmath.h applied elementwise (validation only)"""
        x = x.detach().contiguous()
        if x.dtype != self.dtype or not x.is_cuda:
            raise TypeError('x must be a %s device tensor' % self.dtype)
        out = torch.empty_like(x)
        kinds = {'rcp': 0, 'log': 1, 'exp': 2, 'rsixth': 3, 'cbrt': 4, 'inv': 5}
        self._check(self.lib.ofdft_debug_math(self._ctx, kinds[kind], _ptr(x), _ptr(out), x.numel(), self._stream()), 'ofdft_debug_math')
        return out

    def query(self, what):
        v = C.c_double(0.0)
        self._check(self.lib.ofdft_query(self._ctx, int(what), C.byref(v)), 'ofdft_query')
        return v.value

    def set_option(self, option, value):
        self._check(self.lib.ofdft_set_option(self._ctx, int(option), float(value)), 'ofdft_set_option')
        return self

    def stress(self, den):
        """per-term stress tensors {term name: 3x3 numpy array, Ha/bohr^3} for the active terms at fixed electron number
        (get_stress semantics, functional_tools.py:73-101); the ion-electron entry is zero (see ions.ion_electron_stress)"""
        den = self._grid_tensor(den, 'den')
        if self.dtype != torch.double:       # stress is fp64 work: widen the density and use the fp64 sibling engine
            if self._box_key is None or self._terms_key is None:
                raise RuntimeError('Engine.stress: set_cell and set_terms must be called first')
            sib = engine_for(self.global_shape, self.device)
            sib._box_key, sib._terms_key = None, None          # the sibling is shared: always (re)configure it
            sib._check(sib.lib.ofdft_set_cell(sib._ctx, np.frombuffer(self._box_key, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))),
                       'ofdft_set_cell')
            sib._check(sib.lib.ofdft_set_terms(sib._ctx, self._terms_key[0],
                                               np.frombuffer(self._terms_key[1], dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double)),
                                               N.NPARAMS), 'ofdft_set_terms')
            return sib.stress(den.double())
        buf = (C.c_double * (N.NTERMS * 9))()
        self._check(self.lib.ofdft_stress(self._ctx, C.c_void_p(den.data_ptr()), buf, self._stream()), 'ofdft_stress')
        a = np.array(list(buf), dtype=np.float64).reshape(N.NTERMS, 3, 3)
        return {nm: a[i] for i, nm in enumerate(N.TERM_ORDER)}

    def set_profiling(self, on):
        self._check(self.lib.ofdft_set_profiling(self._ctx, 1 if on else 0), 'ofdft_set_profiling')

    def profile(self):
        """-> {kernel class: (total ms, launches)} accumulated since set_profiling(True)."""
        out = {}
        buf = C.create_string_buffer(64)
        for i in range(self.lib.ofdft_profile_count(self._ctx)):
            ms, n = C.c_double(0.0), C.c_longlong(0)
            self._check(self.lib.ofdft_profile_get(self._ctx, i, buf, 64, C.byref(ms), C.byref(n)), 'ofdft_profile_get')
            out[buf.value.decode()] = (ms.value, n.value)
        return out

    @property
    def fast_path(self):
        return bool(self.query(N.Q_FAST_PATH))


_ENGINES = {}


# extents whose line transforms run in registers / LDS, i.e. grids that take the fused pipelines (csrc/engine_ctx.h:
# line_extent_ok / row_extent_ok); every other extent works too, through chirp-z line transforms and the unfused pipeline
FUSED_EXTENTS_MIXED = (48, 96, 120, 144, 160, 192, 240, 250, 270, 288, 320, 384, 480)       # both builds (fp32 since round 3)


def fused_extents(axis=0, dtype=torch.double, nranks=1):
    """sorted extents along `axis` (0, 1: lines; 2: the real-to-complex rows) served by the fused pipelines"""
    lo, hi = (16, 2048) if axis == 2 else (8, 1024)
    out = {1 << k for k in range(3, 12) if lo <= (1 << k) <= hi}
    out |= set(FUSED_EXTENTS_MIXED)
    if nranks > 1 and axis < 2:                       # slabs: axes 0 and 1 are cut into nranks pieces
        out = {e for e in out if e % nranks == 0}
    return sorted(out)


def next_fast_extent(n, axis=0, dtype=torch.double, nranks=1):
    """smallest fused-pipeline extent >= n along `axis` (ValueError beyond the largest one)"""
    for e in fused_extents(axis, dtype, nranks):
        if e >= n:
            return e
    raise ValueError('no fused-pipeline extent >= %d along axis %d' % (n, axis))


def ecut2shape_fast(energy_cutoff, box_vecs, dtype=torch.double, nranks=1):
    """Grid shape for an energy cutoff (eV) and lattice vectors (Angstrom): the reference's rule (system.py:74-89,
    1 + 2 ceil(k_cut / |b_i|) points per axis -- always odd) rounded UP per axis to the next extent the fused pipelines
    serve, so the cutoff is never lowered.  Returns (shape, reference_shape)."""
    import numpy as np
    A_per_b, eV_per_Ha = 5.29177210903e-11 * 1e10, 4.3597447222071e-18 / 1.602176634e-19       # CODATA 2018, as system.py:27-31
    bvs = np.asarray(box_vecs, dtype=float) / A_per_b
    kcut = np.sqrt(2.0 * float(energy_cutoff) / eV_per_Ha)
    ref = tuple(int(1 + 2 * np.ceil(kcut / (2 * np.pi / np.sqrt((bvs[i] ** 2).sum())))) for i in range(3))
    return tuple(next_fast_extent(r, i, dtype, nranks) for i, r in enumerate(ref)), ref


def engine_for(shape, device, dtype=torch.double):
    """One cached Engine per (shape, device, dtype)."""
    dev = torch.device(device)
    key = (tuple(int(s) for s in shape), dev.type, dev.index if dev.index is not None else 0, dtype)
    e = _ENGINES.get(key)
    if e is None:
        e = _ENGINES[key] = Engine(shape, dev, dtype=dtype)
    return e
