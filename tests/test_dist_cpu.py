"""CPU (gloo, world_size 2 and 4) test of the multi-rank host logic: SlabPlan, Comm and the stage orchestration of
professad_amd.distributed, driven with the numpy stage double and checked against the oracle."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from professad_amd.distributed import SlabPlan

HERE = os.path.dirname(os.path.abspath(__file__))


def test_slab_plan_bookkeeping():
    p = SlabPlan((16, 32, 8), 4, 2)
    assert (p.nxl, p.nyl, p.nzc) == (4, 8, 5) and p.local_shape == (4, 32, 8)
    assert p.x_range() == slice(8, 12) and p.y_range(1) == slice(8, 16) and p.chunk == 5 * 4 * 8
    full = np.arange(16 * 32 * 8).reshape(16, 32, 8)
    assert np.array_equal(np.concatenate([SlabPlan((16, 32, 8), 4, r).scatter(full) for r in range(4)]), full)
    with pytest.raises(ValueError):
        SlabPlan((18, 32, 8), 4, 0)


@pytest.mark.parametrize('world,shape,chunks,two_groups', [(2, '8x12x10', 1, 0), (4, '16x8x9', 1, 1), (2, '8x12x10', 3, 1), (4, '16x8x9', 4, 0),
                                                          (8, '16x8x9', 2, 0), (8, '8x16x6', 1, 1)])
def test_multi_rank_host_logic_under_gloo(world, shape, chunks, two_groups, tmp_path):
    """chunks > 1: the exchange of every step travels as that many kz chunks, each its own all-to-all, consumed chunk by chunk
    by the next step (the sequencing of professad_amd.distributed._run_exchanges).  two_groups: the opt-in
    second process group for the nonlocal chain (Comm).  world 8 = the node size of BASELINE configs 4 / 5."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'res.json')
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   OMP_NUM_THREADS='1', OFDFT_COMM_TWO_GROUPS=str(two_groups))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_cpu_worker.py'), shape, out, str(chunks)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = [p.communicate(timeout=240)[0].decode(errors='replace')[-1500:] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n----\n'.join(logs)
    res = json.load(open(out))
    assert all(v < 1e-12 for v in res.values()), res
