"""Slab-decomposed evaluation over the GPUs of one node: one process per GPU.

Rank r of P owns the real-space x-slab ``[n0/P, n1, n2]`` of chi / n / v_ext and gets back its slab of the
potential / gradient.  The engine (C ABI ``ofdft_dist_*``) does all local work -- z and y passes on x-slabs,
fused x passes on y-slabs, reading / writing the exchange buffers in place -- and this module only sequences the
steps (per chain and per kz chunk of the exchange) and performs the collectives through ``torch.distributed``: equal-split all-to-alls (the FFT transposes,
every array of a stage and chain in ONE message per peer) and two small all-reduces per evaluation.  The work is
two independent chains (density / Hartree / vW / PBE and the nonlocal KEDF) with separate exchange buffers: with
the ``nccl`` backend (= RCCL) the all-to-alls are asynchronous, device-to-device over xGMI, and one chain's
exchange is in flight while the other chain's kernels run; with ``gloo`` the buffers are staged through the host
(used for tests, including several ranks sharing one GPU).

The orchestration (`run_closure`, `run_potential`) is written against a tiny "stages" interface so that the
CPU-only test-suite can drive it with a numpy double under gloo; the product implementation of that
interface is `HipStages` (ctypes -> libofdft_hip.so), which has no CPU fallback.
"""
import contextlib
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _native as N
from .engine import Engine

NSUMS = 13   # 10 combine scalars + GGA sums (PBE exchange, correlation, kinetic GGA)
SUMSQ_SLOT = 15   # slot of sum chi^2 in the context's device-resident scalar block (16 doubles)


class SlabPlan:
    """Index bookkeeping of the decomposition (host logic only)."""

    def __init__(self, shape, nranks, rank):
        n0, n1, n2 = (int(s) for s in shape)
        if n0 % nranks or n1 % nranks:
            raise ValueError('slab decomposition needs n0 and n1 divisible by the number of ranks')
        self.shape = (n0, n1, n2)
        self.nranks, self.rank = int(nranks), int(rank)
        self.nxl, self.nyl = n0 // nranks, n1 // nranks
        self.nzc = n2 // 2 + 1
        self.local_shape = (self.nxl, n1, n2)                  # real-space slab of this rank
        self.x0, self.y0 = self.rank * self.nxl, self.rank * self.nyl
        self.chunk = self.nzc * self.nxl * self.nyl            # complex elements per (peer, array)

    def x_range(self, r=None):
        r = self.rank if r is None else r
        return slice(r * self.nxl, (r + 1) * self.nxl)

    def y_range(self, r=None):
        r = self.rank if r is None else r
        return slice(r * self.nyl, (r + 1) * self.nyl)

    def scatter(self, full):
        """this rank's slab of a global real array"""
        return full[self.x_range()]


def _group_timeout():
    """the timeout the default (world) process group was created with, or None when torch does not tell"""
    try:
        from torch.distributed.distributed_c10d import _get_default_group
        pg = _get_default_group()
        for attr in ('_timeout', 'timeout'):
            t = getattr(pg, attr, None)
            if t is not None:
                return t() if callable(t) else t
        opts = getattr(pg, 'options', None)
        return getattr(opts, '_timeout', None)
    except Exception:  # noqa: BLE001  (private API: absence is not an error)
        return None


class Comm:
    """The two collectives the path needs, over a torch.distributed group (or trivially for one rank).

    The evaluation is two independent chains.  A process group owns ONE communicator (under nccl = RCCL: one internal
    stream per device), so asynchronous all-to-alls issued on the same group execute in host-issue order, and chain 1's
    chunk queues behind chain 0's previously issued message.  `groups[c]` is the group chain c's all-to-alls are issued
    on.  DEFAULT: both chains on `group` -- the ordering every rank shares by construction.
    A second communicator for chain 1 (own stream: its exchanges no longer queue behind chain 0's) is OPT-IN
    (`chain1_group=` or OFDFT_COMM_TWO_GROUPS=1) until it has run on a multi-GPU node: two communicators on one device
    are only safe when every rank's all-to-all kernels of BOTH can be resident at once.  If a compute stream of one chain
    shares a hardware queue with the other chain's spinning all-to-all kernel on rank A while rank B is in the mirror
    state, the ranks wait for each other forever (the runtime maps streams onto 4 hardware queues by default; a slab rank
    has main + side + ipc + torch streams and one stream per communicator).  So the opt-in also REQUIRES
    GPU_MAX_HW_QUEUES >= 8 in the environment before HIP initialises (checked here; bench.py sets it for its workers),
    and every rank must issue the exchanges of both groups in the same order (`_run_exchanges` is deterministic).
    The small all-reduces stay on `groups[0]`."""

    def __init__(self, group=None, chain1_group=None, timeout=None):
        # (OFDFT_COMM_ONE_RANK=1: also route a ONE-rank group's exchanges through the backend -- on a one-GPU box that is the
        # only way to drive the whole staged evaluation through RCCL: tests/test_dist_gpu.py)
        import os
        self.active = dist.is_available() and dist.is_initialized() and (
            dist.get_world_size(group) > 1 or os.environ.get('OFDFT_COMM_ONE_RANK') == '1')
        self.group = group
        self.nranks = dist.get_world_size(group) if self.active else 1
        self.rank = dist.get_rank(group) if self.active else 0
        self.backend = dist.get_backend(group) if self.active else None
        # second group for chain 1 (opt-in, see above).  `dist.new_group` is collective over the WORLD, so it is created here
        # only when this Comm spans the world (every process constructs it); a caller that passes a sub-group creates the
        # second group itself (all world ranks calling new_group) and hands it in as `chain1_group`.
        self._own_group = None
        want_two = os.environ.get('OFDFT_COMM_TWO_GROUPS') == '1' and os.environ.get('OFDFT_COMM_ONE_GROUP') != '1'
        if (chain1_group is not None or want_two) and self.active and self.backend == 'nccl' and self.nranks > 1:
            if int(os.environ.get('GPU_MAX_HW_QUEUES', '4') or 4) < 8:
                raise RuntimeError('a second communicator for the nonlocal chain needs GPU_MAX_HW_QUEUES >= 8 in the environment '
                                   'before HIP initialises (streams that share a hardware queue can deadlock two concurrent '
                                   'all-to-alls across ranks); set it, or drop chain1_group / OFDFT_COMM_TWO_GROUPS')
        if chain1_group is None and want_two and self.active and group is None:
            kw = {}
            if timeout is None:          # the world group's bound holds for chain 1 too (torch's default would be 10 minutes)
                timeout = _group_timeout()
            if timeout is not None:
                kw['timeout'] = timeout
            chain1_group = self._own_group = dist.new_group(ranks=list(range(dist.get_world_size())), backend=self.backend, **kw)
        self.groups = (group, chain1_group if chain1_group is not None else group)
        self.issued = [0, 0]          # all-to-alls issued per chain group (tests assert which groups the chains use)

    def close(self):
        """release the communicator this object created for chain 1 (every rank calls it, in the same order)"""
        if self._own_group is not None and dist.is_initialized():
            g, self._own_group = self._own_group, None
            self.groups = (self.group, self.group)
            dist.destroy_process_group(g)

    def all_reduce_sum(self, vec, device):
        """vec: 1-D numpy fp64 -> summed over ranks (numpy)"""
        if not self.active:
            return vec
        t = torch.as_tensor(vec, dtype=torch.double, device=device if self.backend == 'nccl' else 'cpu').clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def all_reduce_dev(self, t):
        """in-place sum over ranks of a small fp64 DEVICE tensor; no host synchronisation under nccl"""
        if not self.active:
            return
        if self.backend == 'nccl':
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)

    def all_to_all(self, send_t, recv_t, chain=0):
        """equal-split all-to-all between flat byte tensors (device tensors; staged through host under gloo), issued on
        the group of `chain` (0: density / Hartree / vW / GGA and everything else, 1: the nonlocal KEDF chain).
        Returns None when complete on return, else a work handle whose .wait() orders the current stream after it."""
        if not self.active:
            recv_t.copy_(send_t)
            return None
        group = self.groups[1 if chain else 0]
        self.issued[1 if chain else 0] += 1
        if self.backend == 'nccl':
            return dist.all_to_all_single(recv_t, send_t, group=group, async_op=True)
        else:
            # gloo has no all-to-all: host-staged pairwise exchange (test / fallback transport only)
            hs = send_t.cpu()
            hr = torch.empty_like(hs)
            ins = list(hs.chunk(self.nranks))
            outs = list(hr.chunk(self.nranks))
            reqs = []
            for p in range(self.nranks):
                if p == self.rank:
                    outs[p].copy_(ins[p])
                else:
                    reqs.append(dist.isend(ins[p].contiguous(), p, group=group))
                    reqs.append(dist.irecv(outs[p], p, group=group))
            for r in reqs:
                r.wait()
            recv_t.copy_(hr)
            return None


class _NoExchange:
    """Timing aid (bench.py: scale_512.compute_only_ms): a view of a `Comm` whose all-to-alls move nothing -- the ranks' kernels, the
    step sequence, both streams and the small all-reduces run as in a real evaluation, on whatever the receive buffers hold.  What
    it times is a rank's LOCAL wall time per evaluation: the floor the exchange adds to.  The numbers it produces mean nothing."""

    def __init__(self, comm):
        self._c = comm
        self.active, self.nranks, self.rank, self.backend, self.group = comm.active, comm.nranks, comm.rank, comm.backend, comm.group

    def all_reduce_sum(self, vec, device):
        return self._c.all_reduce_sum(vec, device)

    def all_reduce_dev(self, t):
        return self._c.all_reduce_dev(t)

    def all_to_all(self, send_t, recv_t, chain=0):
        return None


class _RawDeviceBuffer:
    """Zero-copy view of engine-owned device memory for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr, count, typestr='|u1'):
        self.__cuda_array_interface__ = {'shape': (int(count),), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


class HipStages(Engine):
    """`stages` interface on top of the ofdft_dist_* C ABI (one slab per rank)."""

    def __init__(self, shape, device, nranks=1, rank=0, dtype=torch.double):
        self.plan = SlabPlan(shape, nranks, rank)
        super().__init__(shape, device, nranks=nranks, rank=rank, dtype=dtype)
        p = C.c_void_p(0)
        self._check(self.lib.ofdft_dist_scalars(self._ctx, C.byref(p)), 'ofdft_dist_scalars')
        # 16 device-resident doubles owned by the context: [0..12] local sums of an evaluation, [15] sum chi^2
        self.device_scalars = torch.as_tensor(_RawDeviceBuffer(p.value, 16, '<f8'), device=self.device)
        self._xbuf = {}

    def enable_collectives(self, comm):
        """Hand the engine the two collectives its per-geometry-step routines (stress, ionic potential, ion-electron forces
        and stress) call back into on a slab-decomposed context (``ofdft_set_collectives``): an equal-split all-to-all of
        engine-owned device buffers and an in-place sum of a small host vector -- both through `comm` (RCCL under the
        nccl backend; host-staged under gloo).  The callbacks run on the calling thread, whose current torch stream is
        the stream the engine call was given, so the collective is ordered with the engine's kernels on it."""
        if not comm.active:
            return self

        def a2a(user, send, recv, nbytes, stream):
            try:
                tot = int(nbytes) * comm.nranks
                key = (int(send), int(recv), tot)
                ex = self._xbuf.get(key)
                if ex is None:
                    ex = self._xbuf[key] = (torch.as_tensor(_RawDeviceBuffer(send, tot), device=self.device),
                                            torch.as_tensor(_RawDeviceBuffer(recv, tot), device=self.device))
                w = comm.all_to_all(ex[0], ex[1])
                if w is not None:
                    w.wait()
                return 0
            except Exception as e:  # noqa: BLE001  (nothing may propagate across the C ABI)
                self._cb_error = e
                return 1

        def allreduce(user, buf, count):
            try:
                v = np.ctypeslib.as_array(buf, shape=(int(count),))
                v[:] = comm.all_reduce_sum(v.copy(), self.device)
                return 0
            except Exception as e:  # noqa: BLE001
                self._cb_error = e
                return 1

        self._cb_error = None
        self._callbacks = (N.A2A_FN(a2a), N.ALLREDUCE_FN(allreduce))       # keep the thunks alive as long as the context
        self._check(self.lib.ofdft_set_collectives(self._ctx, self._callbacks[0], self._callbacks[1], None), 'ofdft_set_collectives')
        return self

    def attach_ipc(self, comm):
        """Set up the library's own exchange (``ofdft_ipc_*``): export this rank's arena (receive buffers + mailbox, one
        allocation), pass the 64-byte hipIpc handles and the object offsets around with ONE all-gather, map every peer's.  Afterwards `closure_ipc` runs an evaluation without
        any torch / RCCL call.  To be repeated after a `set_terms` that changes the exchange buffers."""
        if not comm.active:
            return False
        dev = self.device if comm.backend == 'nccl' else 'cpu'

        def agree(ok):          # every rank learns whether ALL ranks got through a step (no rank is left alone in a collective)
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=comm.group)
            return bool(int(t[0]))
        mine = np.zeros(64 + 40, dtype=np.uint8)           # hipIpc handle of this rank's arena + the five objects' offsets in it
        offs = (C.c_ulonglong * 5)()
        err = None
        try:
            self._check(self.lib.ofdft_ipc_export(self._ctx, mine[:64].ctypes.data_as(C.c_void_p), offs), 'ofdft_ipc_export')
            mine[64:] = np.frombuffer(bytes(offs), dtype=np.uint8)
        except RuntimeError as e:
            err = e
        if not agree(err is None):
            raise RuntimeError('ipc transport: exporting the exchange buffers failed on some rank (%r here)' % (err,))
        t = torch.from_numpy(mine.copy())
        allh = [torch.empty_like(t, device=dev) for _ in range(comm.nranks)]
        dist.all_gather(allh, t.to(dev), group=comm.group)
        try:
            for p, h in enumerate(allh):
                if p == comm.rank:
                    continue
                hh = np.ascontiguousarray(h.cpu().numpy())
                po = (C.c_ulonglong * 5).from_buffer_copy(hh[64:].tobytes())
                self._check(self.lib.ofdft_ipc_attach(self._ctx, p, hh[:64].ctypes.data_as(C.c_void_p), po), 'ofdft_ipc_attach')
        except RuntimeError as e:
            err = e
        if comm.backend == 'nccl':
            torch.cuda.synchronize(self.device)
        if not agree(err is None):        # (also the barrier: nobody starts writing before everybody has mapped)
            raise RuntimeError('ipc transport: mapping the peers\' buffers failed on some rank (%r here)' % (err,))
        self._ipc_terms = self._terms_key
        self._ipc_attached = True        # (DistEngine.close: peers hold this rank's arena open)
        return True

    def closure_ipc(self, chi, n_elec, vext):
        """chi -> (E_terms, mu, chi.grad) on this rank's slab, the exchange done inside the library (ofdft_dist_closure)"""
        out, v = torch.empty_like(chi), torch.empty_like(chi)
        E = (C.c_double * N.NTERMS)()
        mu = C.c_double(0.0)
        self._check(self.lib.ofdft_dist_closure(self._ctx, C.c_void_p(chi.data_ptr()), C.c_void_p(vext.data_ptr() if vext is not None else 0),
                                                float(n_elec), E, C.byref(mu), C.c_void_p(out.data_ptr()), C.c_void_p(v.data_ptr()),
                                                self._stream()), 'ofdft_dist_closure')
        return {nm: E[i] for i, nm in enumerate(N.TERM_ORDER)}, mu.value, out

    def sumsq(self, x, square=True, on_device=False):
        """local sum of x^2 (or x): returned as a float, or left in device_scalars[15] without a host sync"""
        x = self._grid_tensor(x, 'x')
        out = C.c_double(0.0)
        self._check(self.lib.ofdft_dist_sumsq(self._ctx, C.c_void_p(x.data_ptr()), 1 if square else 0,
                                              None if on_device else C.byref(out), self._stream()), 'ofdft_dist_sumsq')
        return None if on_device else out.value

    def begin(self, src, from_chi, cscale, nel, vext, v_out):
        """from_chi: False = src is the density, True = chi with the host scale `cscale`, 2 = chi with the scale formed
        on the device from the all-reduced device_scalars[15]"""
        self._keep = (src, vext, v_out)          # keep the tensors alive for the duration of the evaluation
        self._check(self.lib.ofdft_dist_begin(self._ctx, C.c_void_p(src.data_ptr()), int(from_chi), float(cscale),
                                              float(nel), C.c_void_p(vext.data_ptr() if vext is not None else 0),
                                              C.c_void_p(v_out.data_ptr() if v_out is not None else 0), self._stream()),
                    'ofdft_dist_begin')

    def stage(self, k, chain):
        """-> None or (send, recv) flat uint8 device tensors of nranks * bytes_per_peer bytes"""
        nbytes, sp, rp = C.c_ulonglong(0), C.c_void_p(0), C.c_void_p(0)
        self._check(self.lib.ofdft_dist_stage(self._ctx, int(k), int(chain), self._stream(), C.byref(nbytes), C.byref(sp),
                                              C.byref(rp)), 'ofdft_dist_stage')
        if nbytes.value == 0:
            return None
        tot = nbytes.value * self.plan.nranks
        key = (sp.value, rp.value, tot)
        ex = self._xbuf.get(key)
        if ex is None:           # the engine reuses its buffers: wrap each (pointer, size) once
            ex = self._xbuf[key] = (torch.as_tensor(_RawDeviceBuffer(sp.value, tot), device=self.device),
                                    torch.as_tensor(_RawDeviceBuffer(rp.value, tot), device=self.device))
        return ex

    @property
    def nchunks(self):
        """kz chunks the exchange buffers are cut into (ofdft_query(OFDFT_Q_XCHG_CHUNKS); 1 = the plain layout)"""
        return int(self.query(N.Q_XCHG_CHUNKS))

    def set_xchg_chunks(self, n):
        """0 = automatic, 1 = off, 2..16 = that many kz chunks (every rank the same value)"""
        self.set_option(N.OPT_XCHG_CHUNKS, int(n))
        self._xbuf.clear()
        self._ipc_terms = None          # the ipc transport's view of the buffers is stale
        return self

    def step(self, step, chain, chunk):
        """one (step, chunk) of a chain of the kz-chunked evaluation (ofdft_dist_step; steps 1..6, see include/ofdft_hip.h)
        -> None or (send, recv) flat uint8 device tensors over the chunk's region, nranks * bytes_per_peer bytes each"""
        nbytes, sp, rp = C.c_ulonglong(0), C.c_void_p(0), C.c_void_p(0)
        self._check(self.lib.ofdft_dist_step(self._ctx, int(step), int(chain), int(chunk), self._stream(), C.byref(nbytes), C.byref(sp),
                                             C.byref(rp)), 'ofdft_dist_step')
        if nbytes.value == 0:
            return None
        tot = nbytes.value * self.plan.nranks
        key = (sp.value, rp.value, tot)
        ex = self._xbuf.get(key)
        if ex is None:           # the engine reuses its buffers: wrap each (pointer, size) once
            ex = self._xbuf[key] = (torch.as_tensor(_RawDeviceBuffer(sp.value, tot), device=self.device),
                                    torch.as_tensor(_RawDeviceBuffer(rp.value, tot), device=self.device))
        return ex

    def finish(self, on_device=False):
        """the 11 local sums: as a numpy vector, or left in device_scalars[0:13] without a host sync"""
        if on_device:
            self._check(self.lib.ofdft_dist_finish(self._ctx, None, self._stream()), 'ofdft_dist_finish')
            return None
        sums = (C.c_double * NSUMS)()
        self._check(self.lib.ofdft_dist_finish(self._ctx, sums, self._stream()), 'ofdft_dist_finish')
        return np.array(list(sums), dtype=np.float64)

    def energies(self, global_sums):
        E = (C.c_double * N.NTERMS)()
        vn = C.c_double(0.0)
        g = (C.c_double * NSUMS)(*[float(x) for x in global_sums])
        self._check(self.lib.ofdft_dist_energies(self._ctx, g, E, C.byref(vn)), 'ofdft_dist_energies')
        return {nm: E[i] for i, nm in enumerate(N.TERM_ORDER)}, vn.value

    def chi_grad(self, chi, v, cscale, mu):
        out = torch.empty_like(chi)
        self._check(self.lib.ofdft_dist_chi_grad(self._ctx, C.c_void_p(chi.data_ptr()), C.c_void_p(v.data_ptr()),
                                                 C.c_void_p(out.data_ptr()), float(cscale), float(mu), self._stream()),
                    'ofdft_dist_chi_grad')
        return out

    def sync(self):
        torch.cuda.current_stream(self.device).synchronize()


def _run_exchanges(stages, comm):
    """Step / chain / chunk sequencing of one evaluation (the six steps of ofdft_dist_step, include/ofdft_hip.h).

    Two levels of overlap, both by ordering alone (an asynchronous all-to-all is ordered behind the kernels enqueued before
    it; `work.wait()` orders the current stream behind the exchange, the host never blocks):
      * between the chains: the nonlocal-KEDF chain is issued on its own stream, so its kernels and exchanges neither wait
        for nor delay the density / Hartree / vW / GGA chain's;
      * INSIDE a chain, by kz chunks: the exchange buffers are chunk-major, chunk k of a step is sent as soon
        as its kernels are enqueued, and the kernels of chunk k of the NEXT step wait only for that chunk -- chunk k crosses
        the fabric while chunk k + 1 is in its y pass and chunk k - 1 already in its fused x pass.
    With one chunk this is the plain stage-by-stage sequence."""
    K = int(getattr(stages, 'nchunks', 1))
    dev = getattr(stages, 'device', None)
    on_gpu = isinstance(dev, torch.device) and dev.type == 'cuda'
    side = None
    if on_gpu:
        side = getattr(stages, '_side_stream', None)
        if side is None:
            side = stages._side_stream = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))          # everything ofdft_dist_begin enqueued
    pending = [[None] * K, [None] * K]          # per chain and chunk: the exchange the chain's next step consumes
    for step in (1, 2, 3, 4, 5, 6):
        for chain in (0, 1):
            ctx = torch.cuda.stream(side) if (side is not None and chain == 1) else contextlib.nullcontext()
            with ctx:
                for k in range(K):
                    if pending[chain][k] is not None:
                        pending[chain][k].wait()
                        pending[chain][k] = None
                    ex = stages.step(step, chain, k)
                    if ex is not None:
                        pending[chain][k] = comm.all_to_all(ex[0], ex[1], chain)
    if side is not None:
        torch.cuda.current_stream(dev).wait_stream(side)           # the combine needs both chains


def _run_stages(stages, comm):
    _run_exchanges(stages, comm)
    local = stages.finish()
    return comm.all_reduce_sum(local, getattr(stages, 'device', 'cpu'))


def run_closure(stages, comm, chi, n_elec, vext, vol, npts_global, new_like):
    """The optimize_density closure (system.py:830-838) over slabs -> (E_terms, mu, grad slab)."""
    sc = getattr(stages, 'device_scalars', None)
    if sc is not None:
        # device-resident scalars: sum chi^2 and the closure scale never visit the host; the only host
        # synchronisation of the evaluation is the copy of the 11 all-reduced sums
        stages.sumsq(chi, True, on_device=True)
        comm.all_reduce_dev(sc[SUMSQ_SLOT:SUMSQ_SLOT + 1])
        v = new_like(chi)
        stages.begin(chi, 2, 0.0, n_elec, vext, v)
        _run_exchanges(stages, comm)
        stages.finish(on_device=True)
        comm.all_reduce_dev(sc[0:NSUMS])
        E_terms, vn = stages.energies(sc[0:NSUMS].cpu().numpy())
        mu = vn / n_elec                                    # system.py:851
        return E_terms, mu, stages.chi_grad(chi, v, 0.0, mu)
    s2 = comm.all_reduce_sum(np.array([stages.sumsq(chi, True)]), getattr(stages, 'device', 'cpu'))[0]
    ntilde = s2 / npts_global * vol                     # system.py:833
    cscale = n_elec / ntilde                            # system.py:834
    v = new_like(chi)
    stages.begin(chi, True, cscale, n_elec, vext, v)
    gsums = _run_stages(stages, comm)
    E_terms, vn = stages.energies(gsums)
    mu = vn / n_elec                                    # system.py:851
    return E_terms, mu, stages.chi_grad(chi, v, cscale, mu)


def run_potential(stages, comm, den, vext, vol, npts_global, new_like):
    """energy + dE/dn for a given density slab (functional_tools.py:9-31 semantics)."""
    nsum = comm.all_reduce_sum(np.array([stages.sumsq(den, False)]), getattr(stages, 'device', 'cpu'))[0]
    nel = nsum / npts_global * vol
    v = new_like(den)
    stages.begin(den, False, 1.0, nel, vext, v)
    gsums = _run_stages(stages, comm)
    E_terms, _ = stages.energies(gsums)
    return E_terms, v


class DistEngine:
    """User-facing slab-decomposed engine: same `set_cell` / `set_terms` / `energy_grad_chi` / `energy_potential`
    as `Engine`, on this rank's slab."""

    def __init__(self, shape, device, group=None, dtype=torch.double, transport='collective', xchg_chunks=None, chain1_group=None,
                 timeout=None):
        """dtype=torch.float32 runs the slab-decomposed hot path on the fp32 build (half the bytes on every link);
        stress and ion forces are then formed by the fp64 routines on fp64 slabs.
        transport: 'collective' = the host issues an all-to-all per stage through torch.distributed (RCCL under nccl);
        'ipc' = the library maps the peers' buffers (hipIpc) and moves the spectra itself, one call per evaluation.
        xchg_chunks: kz chunks the exchange is pipelined by inside each chain (both transports; None = automatic, 1 = off).
        chain1_group: second process group over the same ranks for the nonlocal chain's all-to-alls (OPT-IN, see `Comm`:
        needs GPU_MAX_HW_QUEUES >= 8; OFDFT_COMM_TWO_GROUPS=1 creates it when `group` is the world); timeout: of that group."""
        if transport not in ('collective', 'ipc'):
            raise ValueError("transport must be 'collective' or 'ipc'")
        self.transport = transport
        self.comm = Comm(group, chain1_group, timeout)
        self.stages = None
        try:
            self.stages = HipStages(shape, device, nranks=self.comm.nranks, rank=self.comm.rank, dtype=dtype)
            self.stages.enable_collectives(self.comm)
            if xchg_chunks is not None:             # kz chunks of the exchange (None: the engine's automatic choice)
                self.stages.set_xchg_chunks(xchg_chunks)
        except BaseException:
            # a failed set-up must not leak the communicator this object created (callers wrap the constructor in try / except
            # and would never reach close(): one leaked group per attempt, asymmetric group bookkeeping between the ranks)
            if self.stages is not None:
                with contextlib.suppress(Exception):
                    self.stages.close()
            self.comm.close()
            raise
        self.plan = self.stages.plan
        self.npts_global = int(np.prod(self.plan.shape))
        self._vol = None

    def set_cell(self, box_vecs):
        self.stages.set_cell(box_vecs)
        self._box_np = np.ascontiguousarray(torch.as_tensor(box_vecs).detach().cpu().numpy(), dtype=np.float64).reshape(9)
        self._box_c = self._box_np.ctypes.data_as(C.POINTER(C.c_double))
        self._vol = float(abs(np.linalg.det(np.asarray(torch.as_tensor(box_vecs).detach().cpu().numpy(), dtype=np.float64))))
        return self

    def set_terms(self, names, params=None):
        self.stages.set_terms(names, params)
        return self

    def energy_grad_chi(self, chi, n_elec, vext=None):
        chi = self.stages._grid_tensor(chi, 'chi')
        vext = self.stages._grid_tensor(vext, 'v_ext')
        if self.transport == 'ipc' and self.comm.active:
            if getattr(self.stages, '_ipc_terms', None) != self.stages._terms_key:
                self.stages.attach_ipc(self.comm)          # first use, or the term set (hence the buffers) changed
            return self.stages.closure_ipc(chi, n_elec, vext)
        return run_closure(self.stages, self.comm, chi, n_elec, vext, self._vol, self.npts_global, torch.empty_like)

    def compute_only(self, chi, n_elec, vext=None):
        """one evaluation's LOCAL work with the exchange skipped (timing only: see `_NoExchange`); the collective sequencing"""
        chi = self.stages._grid_tensor(chi, 'chi')
        vext = self.stages._grid_tensor(vext, 'v_ext')
        return run_closure(self.stages, _NoExchange(self.comm), chi, n_elec, vext, self._vol, self.npts_global, torch.empty_like)

    def energy_potential(self, den, vext=None):
        den = self.stages._grid_tensor(den, 'den')
        vext = self.stages._grid_tensor(vext, 'v_ext')
        return run_potential(self.stages, self.comm, den, vext, self._vol, self.npts_global, torch.empty_like)

    def query(self, what):
        return self.stages.query(what)

    # ---- once-per-geometry-step quantities (stress, ionic potential, ion-electron forces and stress): slab-decomposed like
    # the hot path.  Each rank works on its x-slab (PME spreading / gathering on its own planes, real-space sums over its
    # points) and on its y-slab of every spectrum; the transforms go through one all-to-all each, the reduced numbers
    # through one small all-reduce each (the engine calls back into `Comm`, see HipStages.enable_collectives).  These
    # routines are fp64 work: an fp32 engine hands them to an fp64 slab engine of the same decomposition.
    def _f64_stages(self):
        if self.stages.dtype == torch.double:
            return self.stages
        if getattr(self, '_f64', None) is None:
            self._f64 = HipStages(self.plan.shape, self.stages.device, nranks=self.comm.nranks, rank=self.comm.rank,
                                  dtype=torch.double).enable_collectives(self.comm)
        self._f64.set_cell(self._box_np.reshape(3, 3))
        return self._f64

    def gather(self, slab):
        """this rank's x-slab -> the full grid on every rank (diagnostics; the routines below do not need it)"""
        slab = self.stages._grid_tensor(slab, 'slab')
        if not self.comm.active:
            return slab
        if self.comm.backend == 'nccl':
            full = torch.empty(self.plan.shape, dtype=slab.dtype, device=slab.device)
            dist.all_gather_into_tensor(full, slab, group=self.comm.group)
            return full
        parts = [torch.empty(self.plan.local_shape, dtype=slab.dtype) for _ in range(self.comm.nranks)]
        dist.all_gather(parts, slab.cpu(), group=self.comm.group)
        return torch.cat(parts).to(slab.device)

    def stress(self, den_slab, names, params=None):
        """per-term stress tensors of the full system (see Engine.stress) from this rank's density slab"""
        eng = self._f64_stages()
        den = self.stages._grid_tensor(den_slab, 'den').double()
        # on an fp64 engine this is the hot path's own context: the stress term set (often a subset, `params=None` = defaults)
        # must not become what later energy_grad_chi / energy_potential calls compute -- the active set is put back afterwards
        # (same mask and parameter bytes: no change of the exchange buffers, so the ipc transport keeps its mappings)
        saved = eng._terms_key if eng is self.stages else None
        try:
            return eng.set_terms(names, params).stress(den)
        finally:
            if saved is not None and eng._terms_key != saved:
                vals = np.frombuffer(saved[1], dtype=np.float64)
                eng._check(eng.lib.ofdft_set_terms(eng._ctx, saved[0], vals.ctypes.data_as(C.POINTER(C.c_double)), len(vals)),
                           'ofdft_set_terms')
                eng._terms_key = saved

    def ion_electron_forces(self, den_slab, species, pme_order=None):
        """ion-electron forces of the full system (see ions.ion_electron_forces) from this rank's density slab"""
        from .ions import ion_electron_forces
        den = self.stages._grid_tensor(den_slab, 'den').double()
        return ion_electron_forces(self._f64_stages(), self._box_np.reshape(3, 3), den, species, pme_order)

    def ion_electron_stress(self, den_slab, species, pme_order=None):
        from .ions import ion_electron_stress
        den = self.stages._grid_tensor(den_slab, 'den').double()
        return ion_electron_stress(self._f64_stages(), self._box_np.reshape(3, 3), den, species, pme_order)

    def ionic_potential(self, species, pme_order=None):
        """this rank's x-slab of v_ext built from the ions (PME: each rank spreads onto its own planes)"""
        from .ions import ionic_potential
        return ionic_potential(self._f64_stages(), self._box_np.reshape(3, 3), species,
                               pme_order).to(self.stages.dtype).contiguous()

    def close(self, sync_peers=False):
        """sync_peers=True (every rank must then call close, in the same order): with the ipc transport attached, every rank first
        lets go of the peers' arenas, the ranks meet, and only then is any arena freed (ofdft_ipc_detach) -- for jobs that go on to
        create another slab engine in the same processes.  The default frees at once and never waits for a peer: a rank that
        closes because ANOTHER rank failed must not be left in a collective."""
        if sync_peers and self.comm.active and getattr(self.stages, '_ipc_attached', False) and dist.is_initialized():
            with contextlib.suppress(Exception):
                self.stages.lib.ofdft_ipc_detach(self.stages._ctx)
                t = torch.zeros(1, dtype=torch.int32, device=self.stages.device if self.comm.backend == 'nccl' else 'cpu')
                dist.all_reduce(t, group=self.comm.group)
            self.stages._ipc_attached = False
        self.stages.close()
        if getattr(self, '_f64', None) is not None:
            self._f64.close()
        self.comm.close()
