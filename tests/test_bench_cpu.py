"""Host-side checks of the measurement tooling (no GPU): the per-class byte model, the rocprofv3 classifier, the
reference pin of the bench workload and the worker launcher's refusal paths."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

import bench  # noqa: E402
import rocprof_summary  # noqa: E402


def test_every_kernel_class_of_the_committed_profile_is_classified_and_has_bytes():
    """the kernel names of the round-1 rocprofv3 table: each engine kernel maps to a class, each class that moves
    grid-sized data has an algorithmic-bytes entry, and the per-evaluation sum is DESIGN.md's ~96C + 12R"""
    names = []
    with open(os.path.join(ROOT, 'profiles', 'r01_final_rocprof_serialised.md')) as fh:
        for line in fh:
            if line.startswith('| ') and not line.startswith('| kernel') and not line.startswith('|---'):
                names.append(line.split('|')[1].strip())
    assert len(names) > 15
    n, word = 256, 8
    cab = bench.class_alg_bytes('cfg3', n, word, 19.0)
    small = {'reduce', 'wgc_table'}
    seen = set()
    for nm in names:
        if any(f in nm for f in rocprof_summary.FOREIGN):
            continue
        c = rocprof_summary.klass(nm)
        assert c is not None, nm
        assert c in cab or c in small, (nm, c)
        seen.add(c)
    assert {'cpass_y', 'xfused_wgc', 'zpbe', 'zi_wgc', 'yderiv', 'xfused_div', 'zf_powers', 'zi_combine'} <= seen
    R, C = 8.0 * n ** 3, 16.0 * n * n * (n // 2 + 1)
    total = sum(v for k, v in cab.items() if k in seen)
    assert abs(total - (96 * C + 16 * R)) < 0.08 * total          # what the fused pipeline really moves (~14.2-14.5 GB)
    assert 14.0e9 < total < 15.5e9


def test_algorithmic_bytes_model_is_the_survey_contract():
    alg, R, C = bench.algorithmic_bytes(256, 'cfg3', 8)
    assert abs(alg - (23 * (R + 5 * C) + 25 * R)) < 1.0 and abs(alg - 22.0e9) < 0.05e9


def test_bench_workload_is_pinned_to_the_reference():
    with open(os.path.join(ROOT, 'tests', 'golden', 'bench_scalars.json')) as fh:
        ref = json.load(fh)
    assert 'cfg3_256' in ref and 'cfg3_128' in ref
    r = ref['cfg3_256']
    ok = bench.reference_check(256, 'cfg3', 'f64', r['E'] * (1 + 2e-12), r['mu'])
    bad = bench.reference_check(256, 'cfg3', 'f64', r['E'] * (1 + 2e-9), r['mu'])
    assert ok['ok'] and not bad['ok']
    assert bench.reference_check(96, 'cfg3', 'f64', 1.0, 1.0) is None         # no pin for that grid: reported as null
    # the pinned inputs are the bench's own: same recipe, same checksum
    import cases
    box, chi, vext, n_elec, _ = bench.make_inputs(128)
    assert abs(cases.checksum(chi[:8, :8, :8], vext[:8, :8, :8]) - ref['cfg3_128']['input_checksum']) < 1e-12
    assert n_elec == ref['cfg3_128']['n_elec']


def test_oracle_reproduces_the_bench_pin_at_64():
    """the CPU restatement (cpu_baseline leg) on the bench recipe equals the reference's closure energy"""
    import torch
    from oracle import refpath as rp
    with open(os.path.join(ROOT, 'tests', 'golden', 'bench_scalars.json')) as fh:
        ref = json.load(fh)['cfg3_64']
    box, chi, vext, n_elec, _ = bench.make_inputs(64)
    table = rp.term_table(torch.as_tensor(vext))
    fns = [table[k] for k in ('ion_electron', 'hartree', 'wgc99', 'pbe_x', 'pbe_c')]
    out = rp.closure(torch.as_tensor(box), torch.as_tensor(chi), n_elec, fns)
    E = float(out[0])
    assert abs(E - ref['E']) <= 1e-11 * abs(ref['E'])


def test_source_stamp_is_stable_and_tracks_the_sources():
    a, b = bench.source_stamp(), bench.source_stamp()
    assert a == b and len(a) == 16


def test_multi_gpu_launch_refuses_without_enough_devices():
    """`python bench.py --gpus 2` on a box without two GPUs ends with a message and code 2 (no silent 1-GPU run)"""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'OFDFT_BENCH_SHARE_GPU')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('this box has two GPUs: the launch itself is covered by the -m gpu tier')
    assert p.returncode == 2 and '--gpus 2' in p.stderr, (p.returncode, p.stderr[-500:])


def test_host_cores_honours_override(monkeypatch):
    monkeypatch.setenv('OFDFT_CPU_THREADS', '3')
    assert bench.host_cores() == 3
    monkeypatch.delenv('OFDFT_CPU_THREADS')
    assert 1 <= bench.host_cores() <= (os.cpu_count() or 1)
