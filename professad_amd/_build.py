"""Build recipe for the HIP engine (gfx950 only).  `python -m professad_amd._build` rebuilds in-tree."""
import contextlib
import fcntl
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIBDIR = os.path.join(PKG, 'lib')
LIB = os.path.join(LIBDIR, 'libofdft_hip.so')            # fp64 (the reference's precision)
LIB_F32 = os.path.join(LIBDIR, 'libofdft_hip_f32.so')    # same sources with -DOFDFT_REAL_F32 (BASELINE config 5)
SOURCES = ['engine.hip', 'lines.hip', 'xpass_a.hip', 'xpass_b.hip', 'zfused.hip', 'resident.hip']     # separately compiled (engine_ctx.h)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'ofdft_hip.h')]


def _stale(lib=None):
    lib = lib or LIB
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def _flags(extra_flags):
    return ['-O3', '-std=c++17', '--offload-arch=gfx950', '-fPIC', '-ffp-contract=on', '-Wall', '-Wno-unused-function',
            '-I', os.path.join(PKG, '..', 'include')] + list(extra_flags)


def _compile_and_link(extra_flags, out, verbose=False):
    """every source to its own object, side by side (hipcc cross-compiles gfx950 without a GPU), then one link"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    tag = '%s.%d' % (os.path.basename(out), os.getpid())
    objdir = os.path.join(LIBDIR, '.obj')
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(objdir, '%s.%s.o' % (tag, src))
        objs.append(obj)
        cmd = [hipcc] + _flags(extra_flags) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        jobs.append((cmd, subprocess.Popen(cmd)))
    err = None
    for cmd, p in jobs:
        if p.wait() != 0:
            err = err or subprocess.CalledProcessError(p.returncode, cmd)
    try:
        if err:
            raise err
        link = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', out]
        if verbose:
            print(' '.join(link), flush=True)
        subprocess.run(link, check=True)
    finally:
        for o in objs:
            with contextlib.suppress(OSError):
                os.remove(o)
    return out


@contextlib.contextmanager
def _build_lock():
    """One builder at a time per checkout (ranks of a multi-process job may all find the library missing at once)."""
    os.makedirs(LIBDIR, exist_ok=True)
    with open(os.path.join(LIBDIR, '.build.lock'), 'w') as fh:
        fcntl.flock(fh, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(fh, fcntl.LOCK_UN)


def build(force=False, verbose=True, extra_flags=(), out=None):
    """Compile csrc/*.hip with hipcc (cross-compiles without a GPU) into lib/libofdft_hip.so (fp64) and
    lib/libofdft_hip_f32.so (fp32 build of the same sources): every source of both libraries compiles side by side.
    `extra_flags` / `out` build ONE experiment variant instead (A/B runs select it with OFDFT_LIB=<path>).
    The link writes to a temporary name and the finished file is renamed into place under a file lock, so a concurrent
    loader never maps a half-written library and concurrent builders do not clobber each other's output."""
    os.makedirs(LIBDIR, exist_ok=True)
    if out is not None:
        tmp = '%s.tmp.%d' % (out, os.getpid())
        _compile_and_link(extra_flags, tmp, verbose)
        os.replace(tmp, out)
        return out
    import threading
    with _build_lock():
        # fp32 build: unsuffixed floating literals are fp32 too (no f64 promotion of `0.5 * x` in the fused kernels: +10 %);
        # fp64 constants that must stay exact are spelled with long-double literals / integer operands in the sources
        todo = [(lib, flags) for lib, flags in ((LIB, []), (LIB_F32, ['-DOFDFT_REAL_F32', '-cl-single-precision-constant']))
                if force or _stale(lib)]          # re-checked under the lock: another process may just have built it
        errs = []

        def one(lib, flags):
            tmp = '%s.tmp.%d' % (lib, os.getpid())
            try:
                _compile_and_link(list(extra_flags) + flags, tmp, verbose)
                os.replace(tmp, lib)
            except Exception as e:  # noqa: BLE001
                errs.append(e)
                with contextlib.suppress(OSError):
                    os.remove(tmp)
        threads = [threading.Thread(target=one, args=t) for t in todo]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errs:
            raise errs[0]
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
