#!/usr/bin/env python3
"""bench.py -- energy+gradient evaluations/s of the OFDFT hot path on MI355X.

A "step" is one density-optimisation closure evaluation chi -> (E, dE/dchi) (reference
system.py:830-838) on the BASELINE.json workload: 256^3 grid, fp64, IonElectron + Hartree +
WGC99(+TF+vW) + PBE, synthetic density (the converged fcc-Al 32^3 fixture tiled 8^3 times plus a
seeded low-|k| perturbation, SURVEY.md §8d option A).  Inputs are resident in HBM before the timed
region.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
HIP-event timed in a separate profiling pass) and `cpu_baseline` (the oracle's op-for-op torch-CPU
restatement of the reference path, timed on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from professad_amd import synth  # noqa: E402
from professad_amd.engine import Engine  # noqa: E402
from professad_amd.distributed import DistEngine  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
XGMI_GBS_PER_LINK_DIRECTION = 76.8      # xGMI is point-to-point, 7 links x ~153.6 GB/s (both directions together) per GPU
CFG3 = ['ion_electron', 'hartree', 'tf', 'vw', 'wgc99_nl', 'pbe_x', 'pbe_c']
CFG2 = ['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'lda_x', 'pz_c']


def algorithmic_bytes(n, cfg, word=8):
    """SURVEY.md §8d byte model: n_fft (R + 5C) + n_pw R (fp32: half of everything)."""
    R = float(word) * n ** 3
    Cc = 2.0 * word * n * n * (n // 2 + 1)
    n_fft, n_pw = (23, 25) if cfg == 'cfg3' else (6, 10)
    return n_fft * (R + 5 * Cc) + n_pw * R, R, Cc


def class_alg_bytes(cfg, n, word, n_ypass, n_yfwd_fused=0):
    """ALGORITHMIC bytes per evaluation of every kernel class of the default (z-fused, split-derivative) pipeline: what
    the class must read + write once, whatever the launch granularity (a class may run as several x- or kz-chunked
    launches).  R = one real grid, C = one half spectrum; DESIGN.md section 4 lists the arrays behind each entry.
    Classes that move no grid-sized data (second-level reductions, scalar kernels, table generation once per cell) have
    no entry."""
    R = float(word) * n ** 3
    Cc = 2.0 * word * n * n * (n // 2 + 1)
    t = {'sum': R,                       # chi
         'chi_grad': 3 * R,              # chi, v -> grad
         'cpass_y': n_ypass * 2 * Cc, 'cpass_x': 2 * Cc, 'zfwd': R + Cc, 'zinv': R + Cc,
         'xfused_lap': 2 * Cc,
         # persistent small-grid kernel (cfg2 term set on a cubic 16^3 / 32^3 / 64^3 grid): chi three times over (phase A per
         # spectrum), three spectra written + read + rewritten + read, three real results written + read, v_ext, v twice, grad
         'resident': (3 + 6 + 1 + 2 + 1) * R + 4 * 3 * Cc}
    if cfg == 'cfg3':
        t.update({'zf_density': 2 * R + 2 * Cc,         # chi -> n^, (sqrt n)^, D_c n
                  # D_b n (out of place) and D_b G_b (in place); on one GPU the y-forward of n^ rides in the first launch (1C -> 2C:
                  # one y pass less in `n_ypass`, one more C written here -- counted by the engine, OFDFT_Q_YFWD_FUSED)
                  'yderiv': (4 + n_yfwd_fused) * Cc,
                  'xfused_n': 3 * Cc,                   # n^ -> vH^, i f_a n^
                  'xfused_div': 2 * Cc,
                  'xfused_wgc': 2 * 6 * Cc,             # two 3 -> 3 launches; their (w0,K1 | K2) kernel tables are NOT algorithmic
                                                        # bytes (k-dependent kernels earn none) -- see `table_MB_per_eval`
                  'zpbe': 4 * Cc + 3 * R,               # A, B in/out; D_c n, chi in; df/dn - 2 D_c G_c out
                  'zf_powers': R + 6 * Cc,
                  'zi_wgc': 6 * Cc + 2 * R,             # six result spectra + chi -> v_part
                  'zi_combine': 4 * Cc + 5 * R})        # vH, lap, D_a G_a, D_b G_b; chi, v_ext, df/dn, v_part -> v
    else:
        t.update({'zf_density': R + 2 * Cc, 'zf_powers': R + Cc, 'xfused_n': 2 * Cc, 'xfused_lind': 2 * Cc,
                  'zi_combine': 3 * Cc + 3 * R})        # vH, lap, K*n^beta; chi, v_ext -> v
    return t


def eval_roofline_block(model_bytes, ms_per_step, world, measured_hbm, class_bytes, extra=None):
    """Whole-evaluation figures of the bench line.  The ROOFLINE FRACTION is formed from bytes that really cross the HBM
    interface -- rocprofv3's FETCH_SIZE / WRITE_SIZE summed over an evaluation when a PMC file of THIS build is committed,
    else the per-class algorithmic bytes of the launches executed (equal to the counters within 1-2 %) -- divided by the
    measured time per step.  SURVEY 8d's byte model counts the x-forward and x-inverse of a transform pair as separate HBM
    passes, which the fused x pass does not make: model bytes / time may exceed the peak, so it is reported as a
    model-equivalent rate WITHOUT a fraction."""
    sec = ms_per_step * 1e-3
    if measured_hbm:
        hbm, basis = float(measured_hbm), 'rocprofv3 FETCH_SIZE/WRITE_SIZE of this build (profiles/pmc_traffic_*.json), all kernels of one evaluation'
    else:
        hbm, basis = float(class_bytes) / world, 'per-class algorithmic bytes of the launches executed (no PMC file stamped for this build)'
    ach = hbm / sec / 1e9 if hbm else None
    out = {'hbm_bytes_per_eval_per_gpu': hbm or None, 'hbm_bytes_basis': basis if hbm else None,
           'achieved_GBs_per_gpu': round(ach, 1) if ach else None, 'peak_GBs': HBM_PEAK_GBS,
           'frac': round(ach / HBM_PEAK_GBS, 4) if ach else None,
           'model_bytes_per_eval': model_bytes,
           'model_equivalent_GBs_per_gpu': round(model_bytes / world / sec / 1e9, 1),
           'measured_hbm_bytes_per_eval': measured_hbm}
    out.update(extra or {})
    return out


def source_stamp():
    """sha256 (16 hex digits) over the native sources the library is built from: ties a committed PMC file to a build."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, 'professad_amd', 'csrc')
    for f in sorted(os.listdir(csrc)) + [os.path.join('..', '..', 'include', 'ofdft_hip.h')]:
        if f.endswith(('.h', '.hip', '.inc')):
            with open(os.path.join(csrc, f), 'rb') as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def make_inputs(n, rank=0):
    """chi, v_ext, box, N_e for an n^3 grid (professad_amd.synth.bench_inputs; the same recipe the golden generator pins
    against the reference, tests/golden/bench_scalars.json)."""
    return synth.bench_inputs(n, os.path.join(ROOT, 'tests', 'golden'), rank)


def _time_cpu_closure(n, reps, budget_s):
    from oracle import refpath as rp
    box, chi, vext, n_elec, _ = make_inputs(n)
    tb, tc, tv = torch.as_tensor(box), torch.as_tensor(chi), torch.as_tensor(vext)
    table = rp.term_table(tv)
    fns = [table[k] for k in ('ion_electron', 'hartree', 'wgc99', 'pbe_x', 'pbe_c')]
    t0 = time.perf_counter()
    rp.closure(tb, tc, n_elec, fns)                    # warm-up: includes WGC99 kernel generation
    first = time.perf_counter() - t0
    times = []
    while len(times) < reps and (sum(times) + first) < budget_s:
        t0 = time.perf_counter()
        rp.closure(tb, tc, n_elec, fns)
        times.append(time.perf_counter() - t0)
    return (min(times) if times else first), first, len(times)


def host_cores():
    """CPU threads this process may really use: the scheduler affinity, capped by the cgroup CPU quota when one is set
    (a one-GPU box grants a share of a bigger host); OFDFT_CPU_THREADS overrides."""
    if os.environ.get('OFDFT_CPU_THREADS'):
        return int(os.environ['OFDFT_CPU_THREADS'])
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            quota, period = fh.read().split()[:2]
        if quota != 'max':
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(sample_n, full_n, budget_s=45.0):
    """The oracle's op-for-op restatement of the reference path (autograd through torch.fft on the host cores),
    timed on a bounded sample of the same workload: the 128^3 sample always; the full grid too when the sample
    predicts it fits the budget (then the full-grid figure is reported, un-scaled)."""
    torch.set_num_threads(host_cores())
    best, first, cnt = _time_cpu_closure(sample_n, 5, 30.0)
    res = {'n': sample_n, 'best': best, 'first': first, 'count': cnt, 'cores': torch.get_num_threads(), 'full': None}
    ratio = (full_n / sample_n) ** 3 * 1.6            # grid points x cache/log-N penalty seen in the survey
    if full_n > sample_n and (first + 2 * best) * ratio < budget_s:
        fb, ff, fc = _time_cpu_closure(full_n, 2, budget_s)
        res['full'] = {'n': full_n, 'best': fb, 'first': ff, 'count': fc}
    return res


def launch_workers(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks (one process per GPU, RCCL) from this parent,
    which makes no GPU call itself (torch.cuda.device_count() does not initialise the device).  The workers are
    ordinary children, never an exec of a process that touched the GPU; rank 0 prints the JSON line."""
    import socket
    import subprocess
    share = os.environ.get('OFDFT_BENCH_SHARE_GPU') == '1'        # rehearsal on a one-GPU box (gloo, every rank on cuda:0)
    ndev = torch.cuda.device_count()
    if ndev < n and not share:
        sys.stderr.write('bench.py: --gpus %d asked for, %d visible (set OFDFT_BENCH_SHARE_GPU=1 OFDFT_BENCH_BACKEND=gloo '
                         'to rehearse on one GPU)\n' % (n, ndev))
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    from professad_amd import _build
    _build.build(verbose=False)                    # once, before the ranks start (they only load the library)
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0 and rc == 0:
                rc = r
                for q in live:                      # a failed rank leaves the others in a collective: end exactly those
                    q.terminate()
    sys.exit(rc)


def scale_512_block(dist, device, backend, rank, world, tdtype, names, box256, chi256, vext256, n_elec256, E256, transports, steps=5):
    """The north star's strong-scaling pair, measured inside the driver's own `bench.py --gpus N` command: the 512^3
    slab-decomposed evaluation on the N GPUs of the job against the SAME 512^3 problem on one GPU (rank 0 alone; the other ranks
    idle at a barrier).  Inputs = the 256^3 bench inputs tiled 2 x 2 x 2 on the device (periodic: the energy must be 8 x the
    256^3 energy to round-off -- checked).  Both transports when both work.  Per-link rate = bytes one rank sends to one peer
    per evaluation / time per evaluation: what the links sustain if the exchange were spread over the whole evaluation (a lower bound of
    the rate while an exchange is in flight)."""
    base = int(np.asarray(chi256).shape[0])          # 256 in the driver's run (a smaller base only in the rehearsal test)
    n = 2 * base
    word = 8 if tdtype == torch.double else 4
    box = torch.as_tensor(box256) * 2.0
    nel = n_elec256 * 8.0
    xs256 = None
    out = {'grid': [n, n, n], 'steps': steps, 'ms_Ngpu': {}, 'inputs': 'the %d^3 bench inputs tiled 2x2x2 on the device' % base}

    def fence():
        torch.cuda.synchronize(device)
        dist.barrier()
        torch.cuda.synchronize(device)

    def agree(ok):          # every rank learns whether ALL ranks got through a set-up step: nobody is left alone in a collective
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t[0]))

    tol = 1e-10 if tdtype == torch.double else 5e-6
    chi_d = torch.as_tensor(chi256, dtype=tdtype, device=device)
    vext_d = torch.as_tensor(vext256, dtype=tdtype, device=device)
    def note(msg):          # progress on stderr (stdout stays ONE line): a long block must not look hung
        if rank == 0:
            sys.stderr.write('bench.py: scale_512 [%6.1f s] %s\n' % (time.perf_counter() - t_block, msg))
            sys.stderr.flush()
    t_block = time.perf_counter()
    for tr in transports:
        note('transport %s: set-up' % tr)
        eng, err = None, None
        try:        # everything that can fail on ONE rank only (memory, ipc attach is guarded inside) happens before the ranks meet
            eng = DistEngine((n, n, n), device, dtype=tdtype, transport=tr,
                             xchg_chunks=int(os.environ['OFDFT_XCHG_CHUNKS']) if os.environ.get('OFDFT_XCHG_CHUNKS') else None)
            eng.set_cell(box).set_terms(names)
            xs = eng.plan.x_range()
            idx = torch.arange(xs.start, xs.stop, device=device) % base          # this rank's x planes of the tiled grid
            chi = chi_d[idx].repeat(1, 2, 2).contiguous()
            vext = vext_d[idx].repeat(1, 2, 2).contiguous()
        except Exception as e:  # noqa: BLE001
            err = e
        if not agree(err is None):
            out.setdefault('errors', {})[tr] = 'set-up failed on some rank (%r on rank %d)' % (err, rank)
            if eng is not None:
                eng.close()
            continue
        note('transport %s: warm-up' % tr)
        for _ in range(2):
            E, mu, g = eng.energy_grad_chi(chi, nel, vext)
        fence()
        note('transport %s: %d timed evaluations' % (tr, steps))
        t0 = time.perf_counter()
        for _ in range(steps):
            E, mu, g = eng.energy_grad_chi(chi, nel, vext)
        fence()
        note('transport %s: timed' % tr)
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.double, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        if 'compute_only_ms' not in out:
            # a rank's LOCAL wall time per evaluation (kernels, launches, both streams, the small all-reduces; the all-to-alls
            # skipped): the floor the exchange adds to -- measured here, on this node, not taken from a one-GPU emulation
            try:
                note('local compute without the exchange')
                for _ in range(2):
                    eng.compute_only(chi, nel, vext)
                fence()
                t1 = time.perf_counter()
                for _ in range(steps):
                    eng.compute_only(chi, nel, vext)
                fence()
                tc = torch.tensor([time.perf_counter() - t1], dtype=torch.double, device=device if backend == 'nccl' else 'cpu')
                dist.all_reduce(tc, op=dist.ReduceOp.MAX)
                out['compute_only_ms'] = round(float(tc.item()) / steps * 1e3, 4)
            except Exception as e:  # noqa: BLE001
                out.setdefault('errors', {})['compute_only'] = repr(e)[:300]
        out['ms_Ngpu'][tr] = round(float(tt[0]) / steps * 1e3, 4)
        out['xchg_chunks'] = eng.stages.nchunks
        out['rel_dE_vs_8x_256'] = abs(sum(E.values()) - 8.0 * E256) / abs(8.0 * E256)
        eng.close(sync_peers=True)          # (every rank is here: peers let go of each other's arenas before any is freed)
        del chi, vext, g
    if rank == 0:          # the one-GPU leg: this rank alone on the whole 512^3 grid (the others wait at the barrier below)
        try:
            note('the same problem on one GPU')
            one = Engine((n, n, n), device, dtype=tdtype).set_cell(box).set_terms(names)
            chi = chi_d.repeat(2, 2, 2).contiguous()
            vext = vext_d.repeat(2, 2, 2).contiguous()
            for _ in range(2):
                one.energy_grad_chi(chi, nel, vext)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for _ in range(steps):
                E1, _, _ = one.energy_grad_chi(chi, nel, vext)
            torch.cuda.synchronize(device)
            out['ms_1gpu'] = round((time.perf_counter() - t0) / steps * 1e3, 4)
            out['rel_dE_1gpu_vs_8x_256'] = abs(sum(E1.values()) - 8.0 * E256) / abs(8.0 * E256)
            one.close()
            del chi, vext
        except Exception as e:  # noqa: BLE001  (rank 0 alone: it must still reach the barrier)
            out.setdefault('errors', {})['1gpu'] = repr(e)[:300]
    dist.barrier()
    if rank == 0:          # the tiled problem's energy is 8 x the 256^3 energy: asserted, not only reported
        out['ok'] = bool(out['ms_Ngpu']) and 'ms_1gpu' in out and max(out.get('rel_dE_vs_8x_256', 1.0), out.get('rel_dE_1gpu_vs_8x_256', 1.0)) < tol
        out['tol'] = tol
    if rank == 0 and out['ms_Ngpu'] and 'ms_1gpu' in out:
        best = min(out['ms_Ngpu'], key=out['ms_Ngpu'].get)
        out['speedup'] = round(out['ms_1gpu'] / out['ms_Ngpu'][best], 3)
        out['speedup_transport'] = best
        Cc = 2.0 * word * n * n * (n // 2 + 1)
        per_link = 19 * (Cc / world) / world                       # bytes one rank sends to ONE peer per evaluation (19 spectra)
        out['MB_per_link_per_eval'] = round(per_link / 1e6, 1)
        out['GBs_per_link_direction'] = {tr: round(per_link / (ms * 1e-3) / 1e9, 1) for tr, ms in out['ms_Ngpu'].items()}
        # what this decomposition CAN give on this node: a rank cannot finish before its own kernels (compute_only_ms, measured
        # above) nor before its bytes have crossed its slowest link (xGMI: 7 links x ~153.6 GB/s bidirectional = ~76.8 GB/s per
        # link and direction) -- with perfect overlap the evaluation takes the larger of the two
        link_ms = per_link / XGMI_GBS_PER_LINK_DIRECTION / 1e6
        out['bound'] = {'link_ms_at_%.1f_GBs_per_link_direction' % XGMI_GBS_PER_LINK_DIRECTION: round(link_ms, 3),
                        'compute_only_ms': out.get('compute_only_ms'),
                        'expected_speedup_at_most': round(out['ms_1gpu'] / max(link_ms, out.get('compute_only_ms') or 0.0), 2),
                        'binding': 'links' if link_ms > (out.get('compute_only_ms') or 0.0) else 'kernels'}
        out['target'] = 'north star: >= 6x at 8 GPUs'
    return out


def grad_stats(g, dist=None, x0=0, npts=None):
    """sum, L2 norm and the eight fixed probes of chi.grad (tests/golden/cases.py: probe_stats) from the device tensor; on
    slabs (`dist` given): g is this rank's x-slab starting at plane x0 of a grid of npts points -- one small all-reduce"""
    import torch
    flat = g.reshape(-1).double()
    npts = npts or flat.numel()
    idx = [int(i * 2654435761 % npts) for i in range(8)]
    off = x0 * (flat.numel() // g.shape[0])
    mine = [(k, i - off) for k, i in enumerate(idx) if 0 <= i - off < flat.numel()]
    acc = torch.zeros(10, dtype=torch.double, device=flat.device)
    acc[0] = flat.sum()
    acc[1] = (flat * flat).sum()
    if mine:
        acc[torch.as_tensor([2 + k for k, _ in mine], device=flat.device)] = flat[torch.as_tensor([j for _, j in mine], device=flat.device)]
    if dist is not None:
        dist.all_reduce(acc)
    acc = acc.cpu()
    return {'sum': float(acc[0]), 'l2': float(acc[1].sqrt()), 'probes': [float(x) for x in acc[2:]], 'probe_idx': idx}


def reference_check(n, cfg, dtype, E, mu, grad=None):
    """The bench workload pinned to the reference (tests/golden/bench_scalars.json, written by make_golden.py --bench
    from the reference's own closure on these inputs): energy, mu and -- when `grad` (grad_stats of the timed call's
    chi.grad) is given -- the L2 norm, the sum and eight probes of chi.grad (system.py:850-853).  Outside the timed region."""
    fn = os.path.join(ROOT, 'tests', 'golden', 'bench_scalars.json')
    if cfg != 'cfg3' or not os.path.exists(fn):
        return None
    with open(fn) as fh:
        ref = json.load(fh).get('cfg3_%d' % n)
    if not ref:
        return None
    tol = 1e-10 if dtype == 'f64' else 5e-6
    dE = abs(E - ref['E']) / abs(ref['E'])
    dmu = abs(mu - ref['mu']) / max(abs(ref['mu']), 1e-300)
    out = {'E_ref_Ha': ref['E'], 'rel_dE': dE, 'mu_ref': ref['mu'], 'rel_dmu': dmu, 'tol': tol,
           'ok': bool(dE < tol and dmu < max(tol, 1e-9)), 'source': 'tests/golden/bench_scalars.json (reference closure, system.py:830-838)'}
    if grad is not None and ref.get('grad'):
        rg = ref['grad']
        gtol = 1e-9 if dtype == 'f64' else 5e-4          # of the gradient's scale (max-norm bar of the parity tests)
        scale = max(abs(p) for p in rg['probes'])
        assert list(grad['probe_idx']) == list(rg['probe_idx'])
        d_l2 = abs(grad['l2'] - rg['l2']) / rg['l2']
        d_sum = abs(grad['sum'] - rg['sum']) / (rg['l2'] * float(n) ** 1.5)       # |sum| <= l2 sqrt(N^3)
        d_probe = max(abs(a - b) for a, b in zip(grad['probes'], rg['probes'])) / scale
        out.update(grad_rel_dl2=d_l2, grad_rel_dsum=d_sum, grad_probe_max_rel=d_probe, grad_tol=gtol)
        out['ok'] = bool(out['ok'] and d_l2 < gtol and d_sum < gtol and d_probe < gtol)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--grid', type=int, default=256, help='n of the n^3 grid (512: the 1-GPU / 8-GPU pair of the slab target)')
    ap.add_argument('--cfg', default='cfg3', choices=['cfg2', 'cfg3'])
    ap.add_argument('--dtype', default='f64', choices=['f64', 'f32'],
                    help='f64: the reference precision (BASELINE metric); f32: the fp32 build (config 5)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-grid', type=int, default=128)
    a = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        launch_workers(a.gpus, sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != a.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d ranks; running %d\n' % (a.gpus, world, world))
    dist = None
    # rehearsal switches for a one-GPU box (never set by the driver): OFDFT_BENCH_BACKEND=gloo stages the exchange
    # through the host, OFDFT_BENCH_SHARE_GPU=1 puts every rank on cuda:0
    backend = os.environ.get('OFDFT_BENCH_BACKEND', 'nccl')
    if os.environ.get('OFDFT_BENCH_SHARE_GPU') == '1':
        local_rank = 0
    if world > 1:
        # A slab rank runs three compute streams, two RCCL communicators (one per chain) or two communication streams of the ipc
        # transport: more streams than the runtime's default of four hardware queues per process, and streams that share a queue
        # serialise -- which would couple the chains' exchanges again.  Must be set before the first HIP call of the process.
        # (rehearsals with the ranks SHARING one GPU keep the default: four processes x eight queues oversubscribe the card's hardware
        # queues and the scheduler time-slices them -- a 128^3 evaluation on four ranks took 87 ms instead of 2 ms)
        if os.environ.get('OFDFT_BENCH_SHARE_GPU') != '1':
            os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        # (a collective that a failed rank never joins must not hold the job for the backend's default ten minutes)
        import datetime
        tmo = datetime.timedelta(seconds=int(os.environ.get('OFDFT_BENCH_COLL_TIMEOUT_S', '300')))
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank), timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)

    n = a.grid
    tdtype = torch.double if a.dtype == 'f64' else torch.float32
    word = 8 if a.dtype == 'f64' else 4
    names = CFG3 if a.cfg == 'cfg3' else CFG2
    # ONE n^3 system for the whole job: on N > 1 GPUs it is slab-decomposed (rank r owns x-slab r) and every
    # 3-D FFT is transposed with an RCCL all-to-all -- strong scaling of the BASELINE workload.
    box, chi_full, vext_full, n_elec, src = make_inputs(n, 0)
    chi_h, vext_h = chi_full, vext_full
    transport, transport_probe = None, None
    if world > 1:
        # Two transports for the FFT transposes: 'ipc' = the library maps the peers' buffers (hipIpc) and moves the spectra
        # itself, one C call per evaluation; 'collective' = the host issues an RCCL all-to-all per stage.  Unless
        # OFDFT_BENCH_TRANSPORT names one, both are tried on a few untimed evaluations (a transport that raises on any rank,
        # or whose energies differ from the other's, is out) and the faster one runs the timed region; both times are reported.
        forced = os.environ.get('OFDFT_BENCH_TRANSPORT')
        xs = None
        cands, transport_probe = {}, {}
        for tr in ([forced] if forced else ['ipc', 'collective']):
            ok, e_tot, ms = True, None, None
            try:
                cand = DistEngine((n, n, n), device, dtype=tdtype, transport=tr,
                                  xchg_chunks=int(os.environ['OFDFT_XCHG_CHUNKS']) if os.environ.get('OFDFT_XCHG_CHUNKS') else None)
                cand.set_cell(torch.as_tensor(box)).set_terms(names)
                xs = cand.plan.x_range()
                c_chi = torch.as_tensor(np.ascontiguousarray(chi_full[xs]), dtype=tdtype, device=device)
                c_vext = torch.as_tensor(np.ascontiguousarray(vext_full[xs]), dtype=tdtype, device=device)
                for _ in range(2):
                    Ep, _, _ = cand.energy_grad_chi(c_chi, n_elec, c_vext)
                torch.cuda.synchronize(device)          # (no barrier here: a rank that raised above must not leave the others in one)
                t0 = time.perf_counter()
                for _ in range(3):
                    Ep, _, _ = cand.energy_grad_chi(c_chi, n_elec, c_vext)
                torch.cuda.synchronize(device)
                ms = (time.perf_counter() - t0) / 3 * 1e3
                e_tot = sum(Ep.values())
            except Exception as e:  # noqa: BLE001
                ok = False
                sys.stderr.write('bench.py: transport %s failed on rank %d: %r\n' % (tr, rank, e))
            st = torch.tensor([1.0 if ok else 0.0, ms or 0.0], dtype=torch.double, device=device if backend == 'nccl' else 'cpu')
            mn = st.clone()
            dist.all_reduce(mn, op=dist.ReduceOp.MIN)
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
            if bool(mn[0] > 0.5):
                cands[tr] = (cand, float(st[1]), e_tot)
                transport_probe[tr] = {'ms_per_eval': round(float(st[1]), 4), 'energy_Ha': e_tot}
            else:
                transport_probe[tr] = {'failed': True}
                if ok:
                    cand.close()
        if not cands:
            sys.stderr.write('bench.py: no slab transport works on this node\n')
            sys.exit(4)
        if len(cands) == 2:
            ea, eb = cands['ipc'][2], cands['collective'][2]
            # they run the same kernels in the same order: any difference is a transport fault -- possibly a rank-local one, so
            # the verdict is shared (MAX over ranks) before anybody drops the transport: all ranks keep or drop it together
            bad = torch.tensor([1.0 if abs(ea - eb) > 1e-10 * abs(eb) else 0.0], dtype=torch.double,
                               device=device if backend == 'nccl' else 'cpu')
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if bool(bad[0] > 0.5):
                transport_probe['ipc']['disagrees_with_collective'] = True
                cands.pop('ipc')[0].close(sync_peers=True)
        transport = min(cands, key=lambda k: cands[k][1])
        for k in list(cands):
            if k != transport:
                cands.pop(k)[0].close(sync_peers=True)          # (the same engines on every rank, in the same order)
        eng = cands[transport][0]
        chi_h, vext_h = np.ascontiguousarray(chi_h[xs]), np.ascontiguousarray(vext_h[xs])
        raw = eng.stages
    else:
        eng = Engine((n, n, n), device, dtype=tdtype).set_cell(torch.as_tensor(box)).set_terms(names)
        raw = eng
    chi = torch.as_tensor(chi_h, dtype=tdtype, device=device)
    vext = torch.as_tensor(vext_h, dtype=tdtype, device=device)
    default_options = True
    for env, opt in (('OFDFT_SIDE_STREAM', 1), ('OFDFT_XCHUNKS', 2), ('OFDFT_XCHUNK_MASK', 3), ('OFDFT_SPLIT_COMBINE', 4),
                     ('OFDFT_GGA_SPLIT', 6), ('OFDFT_XWAVE', 8), ('OFDFT_MIXED_RADIX', 9), ('OFDFT_YBATCH', 14)):          # A/B switches (never set by the driver)
        if os.environ.get(env):
            raw.set_option(opt, int(os.environ[env]))
            default_options = False

    def step():
        return eng.energy_grad_chi(chi, n_elec, vext)

    for _ in range(a.warmup):
        E, mu, g = step()

    def fence():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        E, mu, g = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.double, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / a.steps * 1e3
    evals_per_s = a.steps / dt                    # whole-job rate: all ranks work on the same n^3 system
    n_fft = int(eng.query(0))
    n_launch = int(eng.query(4))
    n_ypass = float(eng.query(5))          # whole-spectrum y line passes actually executed
    n_yfwd = int(eng.query(10))            # y-forward transforms that rode inside a yderiv launch
    dev_ms = raw.query(3) or None          # begin .. end of the last evaluation on this rank's stream (HIP events); the
                                           # persistent small-grid kernel's calls are not bracketed by events (reads 0)

    # ---- per-kernel HIP-event profile (separate pass, not inside the timed region; events are recorded on the streams the
    # kernels are launched on).  Durations are measured with the chains serialised on ONE stream: with the side streams
    # on, kernels of different chains share the GPU and a launch's begin-to-end time is not that kernel's own cost
    raw.set_option(1, 0)
    raw.set_profiling(True)
    nprof = 3
    for _ in range(nprof):
        step()
    prof = raw.profile()
    raw.set_profiling(False)
    if os.environ.get('OFDFT_SIDE_STREAM') != '0':
        raw.set_option(1, 1)
    if world > 1 and ('ypass_send' in prof or 'ypass_recv' in prof):
        # slab-decomposed path: the y passes read / write the exchange buffers; one class for the roofline
        ys = [prof.pop(k) for k in ('ypass_send', 'ypass_recv') if k in prof]
        prof['cpass_y'] = (sum(v[0] for v in ys), sum(v[1] for v in ys))
    tot_ms = sum(v[0] for v in prof.values()) or 1.0
    cab = class_alg_bytes(a.cfg, n, word, n_ypass, n_yfwd) if default_options else {}
    # rocprofv3 FETCH_SIZE / WRITE_SIZE of the same command, committed under profiles/ by tools/profile.sh and stamped
    # with the hash of the sources it profiled: used only when it belongs to THIS build (else traffic = null)
    pmc, pmc_name, stamp = None, None, source_stamp()
    if world == 1:
        import glob
        # pmc_traffic_rNN[_<grid>][_f32][_cfg2].json: the default workload has no tag; other single-GPU workloads that were profiled
        # (tools/profile.sh <dir> --grid 512, ... --dtype f32 --grid 1024 --cfg cfg2) carry theirs
        suffix = ('' if n == 256 else '_%d' % n) + ('' if a.dtype == 'f64' else '_f32') + ('' if a.cfg == 'cfg3' else '_' + a.cfg)
        for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'pmc_traffic_r[0-9][0-9]%s.json' % suffix)), reverse=True):
            with open(fn) as fh:
                cand = json.load(fh)
            if cand.get('source_stamp') == stamp:
                pmc, pmc_name = cand, os.path.basename(fn)
                break
    kernels = {}
    for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]):
        ent = {'ms_per_eval': round(v[0] / nprof, 4), 'launches_per_eval': v[1] // nprof, 'share': round(v[0] / tot_ms, 4)}
        if cab.get(k):
            gbs = cab[k] / world / (v[0] / nprof * 1e-3) / 1e9       # per GPU: a rank holds 1/world of every array
            ent.update(alg_MB_per_eval=round(cab[k] / world / 1e6, 1), achieved_GBs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
        if pmc and k in pmc.get('kernels', {}):
            pk = pmc['kernels'][k]
            ent['traffic_MB_per_eval'] = round(pk['launches'] * (pk['read_MB'] + pk['write_MB']) / pmc['evaluations'], 1)
            if k == 'xfused_wgc' and 'alg_MB_per_eval' in ent:      # what the kernel tables cost on top of the spectra
                ent['table_MB_per_eval'] = round(ent['traffic_MB_per_eval'] - ent['alg_MB_per_eval'], 1)
        kernels[k] = ent
    dom = max((k for k in prof if cab.get(k)), key=lambda k: prof[k][0], default=None)
    roofline = None
    if dom:
        launches = prof[dom][1] / nprof
        class_ms = prof[dom][0] / nprof
        per_launch = cab[dom] / world / launches
        ach = kernels[dom]['achieved_GBs']
        traffic = None
        if pmc and dom in pmc.get('kernels', {}):
            pk = pmc['kernels'][dom]
            traffic = int(round((pk['read_MB'] + pk['write_MB']) * 1e6))
        roofline = {'bound': 'hbm', 'kernel': dom, 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': traffic,
                    'launches_per_eval': launches, 'avg_launch_ms': round(class_ms / launches, 5),
                    'alg_bytes_per_launch': per_launch,
                    'traffic_source': ('profiles/%s (source stamp %s): %s' % (pmc_name, stamp, pmc['source'])) if traffic else None,
                    'note': 'the class with the largest share of the evaluation among all kernel classes; launch durations '
                            'from a profiling pass with the chains serialised on one stream (the setting of the committed '
                            'rocprofv3 summary); the timed region overlaps independent chains on side streams'}
    alg, R, Cc = algorithmic_bytes(n, a.cfg, word)
    measured_hbm = None
    if pmc:       # rocprofv3 FETCH_SIZE / WRITE_SIZE of every kernel of the committed profile, per evaluation
        measured_hbm = sum(k['launches'] * (k['read_MB'] + k['write_MB']) for k in pmc['kernels'].values()) / pmc['evaluations'] * 1e6
    class_bytes = sum(cab[k] for k in prof if cab.get(k))        # algorithmic bytes of the kernel classes that ran
    # the box's own streaming ceiling beside the 8 TB/s spec figure: device-to-device copy of 512 MiB, read + write
    cp_a = torch.empty(64 * 1024 * 1024, dtype=torch.double, device=device)
    cp_b = torch.empty_like(cp_a)
    cp_b.copy_(cp_a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        cp_b.copy_(cp_a)
    ev1.record()
    torch.cuda.synchronize(device)
    copy_gbs = 10 * 2 * cp_a.numel() * 8 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del cp_a, cp_b

    E_tot = sum(E.values())
    out = {
        'metric': 'energy+grad evals/sec', 'value': round(evals_per_s, 3), 'unit': 'evals/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(ms_per_step, 4),
        'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None, 'dtype': a.dtype, 'data': 'synthetic',
        'config': {'workload': '%d^3 grid, %s, IonElectron+Hartree+WGC99(+TF+vW)+PBE closure chi->(E,dE/dchi)' % (n, a.dtype)
                   if a.cfg == 'cfg3' else '%d^3 grid, %s, IonElectron+Hartree+WT(+TF+vW)+PZ-LDA closure' % (n, a.dtype),
                   'grid': [n, n, n], 'terms': names, 'density': src,
                   'parallelism': 'single GPU' if world == 1 else
                   ('x-slab decomposition over %d GPUs, 6 exchanges as RCCL all-to-alls pipelined by kz chunks inside each of two overlapped chains + 2 small all-reduces per evaluation' % world)
                   if transport == 'collective' else
                   ('x-slab decomposition over %d GPUs, library-issued peer copies over hipIpc mappings (6 exchanges pipelined by kz chunks inside each of two overlapped chains) + 2 mailbox reductions per evaluation, no collective call' % world)},
        'roofline': roofline,
        'eval_roofline': eval_roofline_block(alg, ms_per_step, world, measured_hbm, class_bytes, {
            'measured_copy_GBs': round(copy_gbs, 1), 'ffts_executed': n_fft, 'kernel_launches': n_launch,
            'device_ms_last_eval': round(dev_ms, 4) if dev_ms is not None else None}),
        'kernels': kernels,
        'energy_Ha': E_tot, 'mu': mu, 'source_stamp': stamp,
    }
    if transport_probe:
        out['transport'] = {'used': transport, 'probe': transport_probe}
    if world > 1:
        # what crosses the fabric per evaluation and rank: every transposed spectrum leaves (world-1)/world
        # of its slab; with the ipc transport the profiling pass also times the scatter kernels (all links driven at once) and
        # the epoch waits (peer skew + delivery)
        ntr = 19 if a.cfg == 'cfg3' else 6
        sent = ntr * (Cc / world) * (world - 1) / world
        ex = {'spectra_transposed_per_eval': ntr, 'sent_MB_per_rank_per_eval': round(sent / 1e6, 1),
              'MB_per_link_per_eval': round(sent / (world - 1) / 1e6, 1)}
        if 'ipc_scatter' in kernels:
            ms = kernels['ipc_scatter']['ms_per_eval']
            ex.update(scatter_ms_per_eval=ms, scatter_GBs_per_rank=round(sent / (ms * 1e-3) / 1e9, 1) if ms else None,
                      scatter_GBs_per_link=round(sent / (world - 1) / (ms * 1e-3) / 1e9, 1) if ms else None,
                      wait_ms_per_eval=kernels.get('ipc_sync', {}).get('ms_per_eval'))
        out['exchange'] = ex
    # (chi.grad of the timed call itself; on slabs every rank contributes its planes' share of the statistics)
    gst = grad_stats(g) if world == 1 else grad_stats(g, dist, eng.plan.x_range().start, n ** 3)
    chk = reference_check(n, a.cfg, a.dtype, E_tot, mu, gst)
    out['reference_check'] = chk
    if world > 1 and rank == 0 and os.environ.get('OFDFT_BENCH_NO_PARITY') != '1':
        # the slab-decomposed result against ONE engine on the whole grid of this rank's GPU (outside the timed region)
        one = Engine((n, n, n), device, dtype=tdtype).set_cell(torch.as_tensor(box)).set_terms(names)
        E1, mu1, g1 = one.energy_grad_chi(torch.as_tensor(chi_full, dtype=tdtype, device=device), n_elec,
                                          torch.as_tensor(vext_full, dtype=tdtype, device=device))
        gs = g1[eng.plan.x_range()]
        out['parity_vs_single_gpu'] = {
            'rel_dE': abs(E_tot - sum(E1.values())) / abs(sum(E1.values())), 'rel_dmu': abs(mu - mu1) / abs(mu1),
            'grad_slab_max_rel': float((g - gs).abs().max() / gs.abs().max())}
        one.close()
        del g1, gs
    if world > 1:
        out['exchange']['xchg_chunks'] = eng.stages.nchunks
    if world > 1 and (n == 256 or os.environ.get('OFDFT_BENCH_SCALE_ANY_GRID') == '1') and a.cfg == 'cfg3' and os.environ.get('OFDFT_BENCH_NO_SCALE512') != '1':
        # the 512^3 one-GPU / N-GPU pair of the north star, inside this very command (the driver passes no --grid)
        working = [tr for tr, v in (transport_probe or {}).items() if not v.get('failed') and not v.get('disagrees_with_collective')]
        eng.close(sync_peers=True)
        del g
        if rank == 0:       # the primary line survives whatever happens in the optional block (a copy on stderr; stdout stays ONE line)
            sys.stderr.write('bench.py: 256^3 line before the optional scale_512 block: %s\n' % json.dumps(out))
            sys.stderr.flush()
        try:
            sc = scale_512_block(dist, device, backend, rank, world, tdtype, names, box, chi_full, vext_full, n_elec, E_tot, working)
        except Exception as e:  # noqa: BLE001  (a failure every rank shares, e.g. memory: the 256^3 line above must survive it)
            sc = {'error': repr(e)[:300]}
        if rank == 0:
            out['scale_512'] = sc
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(a.cpu_sample_grid, n)
        if cb['full']:
            f = cb['full']
            out['cpu_baseline'] = {
                'value': round(1.0 / f['best'], 5), 'unit': 'evals/s', 'cores': cb['cores'], 'kind': 'port',
                'sample': '%d^3 grid (the full workload), same terms, best of %d after 1 warm-up (%.2f s/eval; first call '
                          '%.1f s incl. WGC99 kernel generation); %d^3 pre-sample: %.3f s/eval'
                          % (f['n'], max(f['count'], 1), f['best'], f['first'], cb['n'], cb['best'])}
        else:
            scale = (a.cpu_sample_grid / n) ** 3
            out['cpu_baseline'] = {
                'value': round(scale / cb['best'], 5), 'unit': 'evals/s', 'cores': cb['cores'], 'kind': 'port',
                'sample': '%d^3 grid, same terms, best of %d after 1 warm-up (%.2f s/eval; first call %.1f s incl. WGC99 '
                          'kernel generation); value = measured %d^3 rate x (%d/%d)^3 grid-point scaling to %d^3'
                          % (cb['n'], cb['count'], cb['best'], cb['first'], cb['n'], cb['n'], n, n),
                'measured_evals_per_s_on_sample': round(1.0 / cb['best'], 4)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if rank == 0 and chk is not None and not chk['ok']:
        sys.stderr.write('bench.py: energy / mu / chi.grad differ from the reference pin beyond %g: %r\n' % (chk['tol'], chk))
        sys.exit(3)


if __name__ == '__main__':
    main()
