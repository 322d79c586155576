#!/usr/bin/env python3
"""Build experiment variants of the fp64 library side by side for A/B runs on one box:
    python tools/build_variants.py name1="-DFLAG=1" name2="-DOTHER=2 -DX=3"   ->  build_ab/lib_<name>.so
(select one with OFDFT_LIB=build_ab/lib_<name>.so; tools/ab_multi2.sh runs several against the default)."""
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from professad_amd import _build  # noqa: E402

os.makedirs(os.path.join(ROOT, 'build_ab'), exist_ok=True)
variants = dict(a.split('=', 1) for a in sys.argv[1:])
res = {}


def one(name, flags):
    try:
        _build.build(out=os.path.join(ROOT, 'build_ab', 'lib_%s.so' % name), extra_flags=flags.split(), verbose=False)
        res[name] = 'ok'
    except Exception as e:  # noqa: BLE001
        res[name] = repr(e)


ts = [threading.Thread(target=one, args=kv) for kv in variants.items()]
for t in ts:
    t.start()
for t in ts:
    t.join()
print(res)
