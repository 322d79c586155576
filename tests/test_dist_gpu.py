"""Slab-decomposed (multi-rank) evaluation against the single-GPU engine: several gloo ranks sharing cuda:0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _gpu_count():
    import torch
    return torch.cuda.device_count()          # does not initialise the GPU in this (parent) process


def _check_worker_results(res, dtype):
    if dtype == 'f32':      # fp32 build: slabs against one fp32 engine differ by fp32 round-off only (tests/test_gpu_f32.py)
        for cfg, w in res.items():
            if cfg == 'chunks':
                assert max(w['dE'], w['dE2'], w['dmu'], w['dg'], w['dv']) == 0.0 and w['ffts'] == w['ffts_ref'], w
            elif cfg == 'opt':
                assert w['dE'] < 2e-5 and abs(w['ffts'] - w['ffts_ref']) <= 1, w
            elif cfg == 'ions':
                assert w['dE'] < 1e-6 and w['dE2'] < 1e-9, w          # fp64 routines on both sides; v_ext narrowed to fp32
            elif cfg == 'stress':
                assert w['dE'] < 1e-12, w                             # fp64 routine on the same widened density
            else:
                assert w['dE'] < 5e-6 and w['dE2'] < 5e-6 and w['dmu'] < 5e-6 and w['dg'] < 5e-4 and w['dv'] < 5e-4, (cfg, w)
                assert w['ffts'] == w['ffts_ref'], (cfg, w)
        return
    for cfg, w in res.items():
        if cfg == 'chunks':   # kz-chunked exchange against the unchunked sequence on the same ranks: identical numbers
            assert w['dE'] == 0.0 and w['dE2'] == 0.0 and w['dmu'] == 0.0 and w['dg'] == 0.0 and w['dv'] == 0.0, w
            assert w['ffts'] == w['ffts_ref'], ('chunk count not honoured', w)
            continue
        if cfg == 'opt':      # 8 outer L-BFGS steps over slabs vs one GPU: same path up to the optimiser's sensitivity to
            assert w['dE'] < 1e-7 and w['dg'] < 1e-3 and abs(w['ffts'] - w['ffts_ref']) <= 1, w      # round-off
            continue
        assert w['dE'] < 1e-12 and w['dE2'] < 1e-12 and w['dmu'] < 1e-12, (cfg, w)
        # energy_potential(den) forms N_e = sum(den) dV itself: the ranks' partial sums and one GPU's add up in different orders
        # (a last-bit difference of N_e), and the Lindhard factor's direct form (functionals.py:617-628) amplifies a relative
        # change of eta by ~3 eta^2 / |f| at the large eta this rough random density still populates: 1e-12 of max |v| was met by
        # luck of the summation order, not by construction (1.4e-12 seen after the round-5 sum kernel); the parity bar is 5e-10
        assert w['dg'] < 1e-12 and w['dv'] < (5e-12 if cfg == 'cfg2' else 1e-12), (cfg, w)
        assert w['ffts'] == w['ffts_ref'], (cfg, w)


def _run_workers(world, shape, dtype, out, extra_env=None, timeout=240):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0', **(extra_env or {}))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), shape, out, dtype], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:               # exactly the processes started here
                q.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors='replace')[-2000:])
    assert all(p.returncode == 0 for p in procs), '\n----\n'.join(logs)
    return json.load(open(out))


@pytest.mark.skipif(_gpu_count() < 2, reason='the RCCL transport needs one GPU per rank (>= 2 GPUs on the box)')
@pytest.mark.parametrize('shape,dtype', [('64x64x64', 'f64'), ('256x256x256', 'f64'), ('64x64x64', 'f32')])
def test_slab_decomposed_over_rccl_matches_single_gpu(shape, dtype, tmp_path):
    """The product transport: one process per GPU, `nccl` (= RCCL) all-to-alls on the engine's own exchange buffers
    (raw-pointer tensors), two overlapped chains, device-resident scalars -- against one engine on rank 0's GPU.
    Runs wherever the box has >= 2 GPUs (a one-GPU box cannot host two RCCL ranks)."""
    world = 2 if _gpu_count() < 4 else 4
    # 'collective2': the opt-in second RCCL communicator for the nonlocal chain (Comm: OFDFT_COMM_TWO_GROUPS=1 with eight hardware
    # queues per process) -- its first run on real peers happens here, wherever a box with >= 2 GPUs runs this suite
    for transport in ('collective', 'ipc', 'collective2'):          # RCCL all-to-alls issued by the host; peer copies issued by the library
        env = {'OFDFT_TEST_BACKEND': 'nccl', 'OFDFT_TEST_TRANSPORT': transport.rstrip('2')}
        if transport == 'collective2':
            env.update(OFDFT_COMM_TWO_GROUPS='1', GPU_MAX_HW_QUEUES='8')
        if shape == '64x64x64' and world == 2:       # ... and the kz-chunked exchange against the unchunked one on the same GPUs
            env['OFDFT_TEST_XCHG_CHUNKS'] = '2'      # (256^3 runs chunked by the automatic choice)
        res = _run_workers(world, shape, dtype, str(tmp_path / ('res_%s.json' % transport)), env, timeout=600)
        _check_worker_results(res, dtype)


# (world 4 = the most ranks a -m gpu test can start: the slab path takes power-of-two rank counts, and the GPU box admits 6
# processes on its card at once, the test runner being one of them -- 8 real processes on one card are not possible here.  The
# node size of BASELINE configs 4 / 5, 8 ranks, is covered by the 8-rank emulator below (kernels, layouts, step order) and by
# tests/test_dist_cpu.py (Comm / group creation / step sequencing under gloo with 8 processes))
@pytest.mark.parametrize('world,shape,dtype', [(2, '32x32x32', 'f64'), (4, '16x64x32', 'f64'), (2, '64x32x128', 'f64'), (4, '32x32x32', 'f32'),
                                               (2, '48x96x120', 'f64'), (4, '96x48x120', 'f32')])
def test_slab_decomposed_with_the_library_own_exchange(world, shape, dtype, tmp_path):
    """transport='ipc': the ranks map each other's receive buffers and mailboxes through hipIpc and the library moves the
    spectra itself (peer copies + epoch stamps + bounded waits), one C call per evaluation and no collective in it -- here with
    the ranks sharing one GPU, which exercises the very same mappings, copies and flags as separate GPUs do"""
    res = _run_workers(world, shape, dtype, str(tmp_path / 'res.json'), {'OFDFT_TEST_TRANSPORT': 'ipc'})
    _check_worker_results(res, dtype)


@pytest.mark.parametrize('world,shape,dtype', [(1, '16x16x32', 'f64'), (2, '32x32x32', 'f64'), (4, '16x64x32', 'f64'),
                                               (2, '64x32x128', 'f64'), (2, '32x32x32', 'f32'), (4, '16x64x32', 'f32'),
                                               (2, '48x48x96', 'f64'), (2, '48x96x120', 'f32')])
def test_slab_decomposed_matches_single_gpu(world, shape, dtype, tmp_path):
    res = _run_workers(world, shape, dtype, str(tmp_path / 'res.json'))
    _check_worker_results(res, dtype)


@pytest.mark.parametrize('world,shape,dtype,chunks,transport', [
    (2, '64x64x64', 'f64', 4, 'collective'), (2, '64x64x64', 'f64', 2, 'ipc'), (4, '128x128x64', 'f64', 4, 'ipc'),
    (4, '128x128x32', 'f64', 2, 'collective'), (2, '64x64x128', 'f32', 4, 'ipc'), (2, '64x64x64', 'f32', 3, 'collective')])
def test_kz_chunked_exchange_with_real_ranks(world, shape, dtype, chunks, transport, tmp_path):
    """SURVEY.md 8e / round-2 verdict item 1: the exchange of every step cut into kz chunks (chunk-major buffers, one message
    per chunk, the next step consuming chunk by chunk; professad_amd.distributed._run_exchanges, ofdft_dist_closure) -- with
    2 and 4 processes sharing the GPU, both transports: bitwise the unchunked sequence, and the single-GPU engine to 1e-12,
    with the same FFT count"""
    res = _run_workers(world, shape, dtype, str(tmp_path / 'res.json'),
                       {'OFDFT_TEST_TRANSPORT': transport, 'OFDFT_TEST_XCHG_CHUNKS': str(chunks)}, timeout=400)
    assert 'chunks' in res
    _check_worker_results(res, dtype)


@pytest.mark.parametrize('nranks,shape,chunks', [(2, (64, 64, 64), 4), (4, (128, 128, 32), 2), (8, (256, 256, 256), 4),
                                                  (8, (256, 256, 64), 3), (2, (64, 64, 48), 3)])
def test_kz_chunked_exchange_emulated_is_bitwise_the_unchunked_one(nranks, shape, chunks):
    """the same in one process with emulated ranks (every rank count a one-GPU box cannot host as processes; uneven chunk
    sizes; a mixed-radix z extent): chunked == unchunked bit for bit, FFT count unchanged, single-GPU engine to 1e-12"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    import cases
    from local_ranks import LocalRanks
    from professad_amd import synth
    from professad_amd.engine import Engine
    from professad_amd.functionals import NativeTerms
    dev = torch.device('cuda:0')
    box = torch.as_tensor(cases.make_cell(('tri', shape[0] / 16.0)))
    den = synth.random_density(shape, seed=41)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.double, device=dev)  # noqa: E731
    chi = t(np.sqrt(den) * (1 + 0.1 * np.random.default_rng(43).random(shape)))
    vext = t(synth.random_potential(shape, seed=42))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box.numpy()))) + 0.3)
    for names, params in ((NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe']).names, None),
                          (['ion_electron', 'hartree', 'vw', 'tf', 'wt_nl', 'gga_k', 'pbe_x'], {'wt_alpha': 1.1, 'wt_beta': 0.6, 'ggak_kind': 1.0,
                                                                                               'ggak_beta': 0.25, 'ggak_lambda': 0.4})):
        ref = Engine(shape, dev).set_cell(box).set_terms(names, params)
        Er, mur, gr = ref.energy_grad_chi(chi, n_elec, vext)
        nfft = ref.query(0)
        ref.close()
        out = {}
        for K in (1, chunks):
            loc = LocalRanks(shape, dev, nranks).set_cell(box).set_terms(names, params).set_xchg_chunks(K)
            assert loc.st[0].nchunks == K
            out[K] = loc.closure(chi, n_elec, vext) + (loc.st[0].query(0),)
            out[(K, 2)] = loc.closure(chi * 1.01, n_elec, vext)          # a second evaluation on the same contexts
            loc.close()
        for key in (chunks, (chunks, 2)):
            a, b = out[key], out[1 if key == chunks else (1, 2)]
            assert all(a[0][k] == b[0][k] for k in a[0]) and a[1] == b[1] and torch.equal(a[2], b[2]), (names, key)
        E, mu, g, nf = out[chunks]
        assert nf == nfft
        for k in Er:
            assert abs(E[k] - Er[k]) <= 1e-12 * max(1.0, abs(Er[k])), (k, E[k], Er[k])
        assert abs(mu - mur) <= 1e-12 * max(1.0, abs(mur))
        assert float((g - gr).abs().max()) <= 1e-12 * float(gr.abs().max())


def test_ipc_transport_fails_an_evaluation_on_all_ranks_and_recovers(tmp_path):
    """round-2 advice (ipc wait): a rank that is later than the transport's patience must not leave the others with an error
    and itself with garbage.  Two ranks sharing the GPU, patience 0.4 s, rank 1 two seconds late for the second evaluation:
    both ranks raise for THAT evaluation (the waiter posts the evaluation number into every rank's abort word; the late rank
    finds it), and the third evaluation is bitwise the first on both ranks (tests/ipc_abort_worker.py)"""
    port = _free_port()
    out = str(tmp_path / 'res.json')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'ipc_abort_worker.py'), out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            o, _ = p.communicate()
        logs.append(o.decode(errors='replace')[-2000:])
    assert all(p.returncode == 0 for p in procs), '\n----\n'.join(logs)
    res = [json.load(open('%s.%d' % (out, r))) for r in range(2)]
    for r in res:
        assert r['second'].startswith('raised') and 'ipc transport' in r['second'], res
        assert r['third_equals_first'], res
    assert 'no delivery from rank 1' in res[0]['second'] and 'aborted by another rank' in res[1]['second'], res
    assert res[0]['dE_vs_single_gpu'] < 1e-12


def test_whole_stage_entry_point_equals_the_step_sequence():
    """ofdft_dist_stage (the one-chunk, whole-stage form of the ABI) against ofdft_dist_step on the same emulated ranks: identical
    numbers; and it refuses a context whose exchange is cut into chunks"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    import cases
    from local_ranks import LocalRanks
    from professad_amd import synth
    from professad_amd.functionals import NativeTerms
    dev = torch.device('cuda:0')
    shape = (64, 64, 32)
    box = torch.as_tensor(cases.make_cell(('tri', 4.0)))
    den = synth.random_density(shape, seed=41)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.double, device=dev)  # noqa: E731
    chi, vext = t(np.sqrt(den)), t(synth.random_potential(shape, seed=42))
    names = NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe']).names
    out = {}
    for legacy in (False, True):
        loc = LocalRanks(shape, dev, 2).set_cell(box).set_terms(names).set_xchg_chunks(1)
        loc.legacy_stage_api = legacy
        out[legacy] = loc.closure(chi, 7.3, vext)
        out[(legacy, 2)] = loc.closure(chi * 1.02, 7.3, vext)
        loc.close()
    for key in (False, (False, 2)):
        a, b = out[key], out[True if key is False else (True, 2)]
        assert all(a[0][k] == b[0][k] for k in a[0]) and a[1] == b[1] and torch.equal(a[2], b[2])
    loc = LocalRanks(shape, dev, 2).set_cell(box).set_terms(names).set_xchg_chunks(2)
    assert loc.st[0].nchunks == 2
    loc.legacy_stage_api = True
    with pytest.raises(RuntimeError, match='ofdft_dist_step'):
        loc.closure(chi, 7.3, vext)
    loc.close()


def test_eight_rank_geometry_with_the_laplacian_dependent_pauli_gaussian():
    """PGSL0.25 + Hartree + PBE over 8 emulated slab ranks (lap n and df/dL cross the exchange beside the GGA spectra)"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    import cases
    from local_ranks import LocalRanks
    from professad_amd import synth
    from professad_amd.engine import Engine
    dev = torch.device('cuda:0')
    shape = (64, 32, 32)
    box = torch.as_tensor(cases.make_cell(('tri', 2.0)))
    den = synth.smooth_density(shape, seed=41)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.double, device=dev)  # noqa: E731
    chi = t(np.sqrt(den) * (1 + 0.05 * np.random.default_rng(43).random(shape)))
    vext = t(synth.random_potential(shape, seed=42))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box.numpy()))) + 0.3)
    names = ['ion_electron', 'hartree', 'vw', 'gga_k', 'pbe_x', 'pbe_c']
    params = {'ggak_kind': 1.0, 'ggak_mu': 40 / 27, 'ggak_beta': 0.25, 'ggak_lambda': 0.4, 'ggak_sigma': 0.2}
    ref = Engine(shape, dev).set_cell(box).set_terms(names, params)
    Er, mur, gr = ref.energy_grad_chi(chi, n_elec, vext)
    ref.close()
    for nranks in (2, 8):
        loc = LocalRanks(shape, dev, nranks).set_cell(box).set_terms(names, params)
        E, mu, g = loc.closure(chi, n_elec, vext)
        loc.close()
        for k in Er:
            assert abs(E[k] - Er[k]) <= 1e-12 * max(1.0, abs(Er[k])), (nranks, k, E[k], Er[k])
        assert abs(mu - mur) <= 1e-12 * max(1.0, abs(mur))
        assert float((g - gr).abs().max()) <= 1e-12 * float(gr.abs().max())


@pytest.mark.parametrize('nranks,shape', [(8, (32, 64, 16)), (8, (64, 64, 64)), (8, (256, 256, 256)),
                                          (2, (48, 96, 120)), (4, (120, 144, 250)), (8, (240, 240, 240))])       # extents with factors 3 / 5
def test_eight_rank_geometry_in_one_process(nranks, shape):
    """the slab geometry of an 8-GPU job (kernels, pack / un-pack, stage order), emulated with 8 contexts on one GPU"""
    import numpy as np
    import torch
    sys.path.insert(0, os.path.join(HERE, 'golden'))
    import cases
    from local_ranks import LocalRanks
    from professad_amd import synth
    from professad_amd.engine import Engine
    from professad_amd.functionals import NativeTerms
    dev = torch.device('cuda:0')
    box = torch.as_tensor(cases.make_cell(('tri', shape[0] / 16.0)))
    den = synth.random_density(shape, seed=41)
    t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.double, device=dev)  # noqa: E731
    chi = t(np.sqrt(den) * (1 + 0.1 * np.random.default_rng(43).random(shape)))
    vext = t(synth.random_potential(shape, seed=42))
    n_elec = float(np.floor(den.mean() * abs(np.linalg.det(box.numpy()))) + 0.3)
    names = NativeTerms(['ion_electron', 'hartree', 'wgc99', 'pbe']).names
    ref = Engine(shape, dev).set_cell(box).set_terms(names)
    Er, mur, gr = ref.energy_grad_chi(chi, n_elec, vext)
    ref.close()
    loc = LocalRanks(shape, dev, nranks).set_cell(box).set_terms(names)
    E, mu, g = loc.closure(chi, n_elec, vext)
    loc.close()
    for k in Er:
        assert abs(E[k] - Er[k]) <= 1e-12 * max(1.0, abs(Er[k])), (k, E[k], Er[k])
    assert abs(mu - mur) <= 1e-12 * max(1.0, abs(mur))
    assert float((g - gr).abs().max()) <= 1e-12 * float(gr.abs().max())


def test_rccl_calls_on_engine_owned_memory_with_one_rank():
    """the torch.distributed (nccl = RCCL) calls of the collective transport on the tensors it passes them -- zero-copy views of
    engine-owned device memory, async all-to-all on a side stream, in-place all-reduce of the device scalars -- in a one-rank group
    (tests/nccl_one_rank_worker.py): what can be exercised of the RCCL branch on a one-GPU box"""
    import subprocess
    env = dict(os.environ, MASTER_PORT=str(29500 + os.getpid() % 400))
    r = subprocess.run([sys.executable, os.path.join(HERE, 'nccl_one_rank_worker.py')], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize('shape,dtype', [('32x32x32', 'f64'), ('64x64x64', 'f32')])
def test_whole_staged_evaluation_through_rccl_with_one_rank(shape, dtype, tmp_path):
    """the complete slab-decomposed worker (closure + potential for three term sets, stress, ionic potential, forces) with backend
    nccl and ONE rank whose collectives are forced through the backend (OFDFT_COMM_ONE_RANK=1): the staged ABI, the side-stream
    sequencing and every small reduction (RCCL all-reduces in place on the context's device scalars, host vectors of the per-step
    routines) -- against the single-GPU engine.  (A one-rank context has nothing to transpose, so the all-to-all itself is
    covered by tests/nccl_one_rank_worker.py and, with real peers, by the >= 2-GPU test above.)"""
    res = _run_workers(1, shape, dtype, str(tmp_path / 'res.json'), {'OFDFT_TEST_BACKEND': 'nccl', 'OFDFT_COMM_ONE_RANK': '1'},
                       timeout=400)
    _check_worker_results(res, dtype)


@pytest.mark.parametrize('ranks', [2, 4])
def test_bench_scale_pair_rehearsal_ranks_sharing_the_gpu(ranks):
    """`bench.py --gpus N` end to end (self-launched workers, transport probe, timed region, then the one-GPU / N-GPU pair of
    the north star appended by `scale_512_block`) at a reduced grid: 64^3 bench inputs, the pair on 128^3.  The driver's
    round-end run on a real multi-GPU node is the same command with nccl and one GPU per rank."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, OFDFT_BENCH_SHARE_GPU='1', OFDFT_BENCH_BACKEND='gloo', OFDFT_BENCH_SCALE_ANY_GRID='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(ranks), '--grid', '64', '--steps', '3', '--warmup', '1',
                        '--no-cpu-baseline'], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors='replace')[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == ranks and line['reference_check']['ok'], line
    sc = line['scale_512']
    assert sc.get('ok') and sc['grid'] == [128, 128, 128] and 'errors' not in sc, sc
    assert 'collective' in sc['ms_Ngpu'] and sc['ms_1gpu'] > 0 and sc['speedup'] > 0, sc
    # the line explains itself: a rank's local wall time with the exchange skipped, the link time of its bytes, and the
    # speed-up the larger of the two allows (round-4 verdict item 3)
    b = sc['bound']
    assert sc['compute_only_ms'] > 0 and b['compute_only_ms'] == sc['compute_only_ms'] and b['binding'] in ('links', 'kernels'), sc
    assert b['expected_speedup_at_most'] > 0 and any(k.startswith('link_ms_at_') for k in b), sc
    # the timed workload's chi.grad is pinned on slabs too (every rank contributes its planes)
    assert line['reference_check']['grad_probe_max_rel'] < 1e-9 and line['reference_check']['grad_rel_dl2'] < 1e-9, line['reference_check']
    assert b'256^3 line before the optional scale_512 block' in p.stderr
