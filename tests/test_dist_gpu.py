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


@pytest.mark.parametrize('world,shape', [(1, '16x16x32'), (2, '32x32x32'), (4, '16x64x32'), (2, '64x32x128')])
def test_slab_decomposed_matches_single_gpu(world, shape, tmp_path):
    port = _free_port()
    out = str(tmp_path / 'res.json')
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, 'dist_worker.py'), shape, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        o, _ = p.communicate(timeout=240)
        logs.append(o.decode(errors='replace')[-2000:])
    assert all(p.returncode == 0 for p in procs), '\n----\n'.join(logs)
    res = json.load(open(out))
    for cfg, w in res.items():
        assert w['dE'] < 1e-12 and w['dE2'] < 1e-12 and w['dmu'] < 1e-12, (cfg, w)
        assert w['dg'] < 1e-12 and w['dv'] < 1e-12, (cfg, w)
        assert w['ffts'] == w['ffts_ref'], (cfg, w)
