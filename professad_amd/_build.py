"""Build recipe for the HIP engine (gfx950 only).  `python -m professad_amd._build` rebuilds in-tree."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'libofdft_hip.so')
SOURCES = ['engine.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'ofdft_hip.h')]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=True, extra_flags=(), out=None):
    """Compile csrc/*.hip into lib/libofdft_hip.so with hipcc (cross-compiles without a GPU).
    `extra_flags` / `out` build an experiment variant (A/B runs select it with OFDFT_LIB=<path>)."""
    if out is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [hipcc, '-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-shared',
           '-ffp-contract=on', '-Wall', '-Wno-unused-function',
           '-I', os.path.join(PKG, '..', 'include')]
    cmd += list(extra_flags)
    cmd += [os.path.join(CSRC, s) for s in SOURCES] + ['-o', out or LIB]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return out or LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
