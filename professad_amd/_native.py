"""ctypes binding of libofdft_hip.so (include/ofdft_hip.h).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

# torch first, always: PyTorch-ROCm bundles its own libamdhip64/libhsa-runtime64 (same SONAMEs as
# /opt/rocm).  Loading our library before torch binds the process to the system runtime and the
# two then disagree about the device ("no ROCm-capable device"); loading torch first makes the
# whole process -- torch tensors, streams and this engine -- share ONE HIP runtime.
import torch  # noqa: F401,E402

from . import _build

_LIBS = {}

# mirror of the #defines in include/ofdft_hip.h
OK, EINVAL, EHIP, ESTATE, ENOMEM = 0, -1, -2, -3, -4
F64, F32 = 0, 1
TERM_BITS = {
    'ion_electron': 1 << 0, 'hartree': 1 << 1, 'tf': 1 << 2, 'vw': 1 << 3, 'wt_nl': 1 << 4, 'wgc99_nl': 1 << 5,
    'lda_x': 1 << 6, 'pz_c': 1 << 7, 'pw_c': 1 << 8, 'chachiyo_c': 1 << 9, 'pbe_x': 1 << 10, 'pbe_c': 1 << 11,
    'gga_k': 1 << 12, 'vwgtf': 1 << 13,
}
TERM_ORDER = ['ion_electron', 'hartree', 'tf', 'vw', 'wt_nl', 'wgc99_nl', 'lda_x', 'pz_c', 'pw_c', 'chachiyo_c',
              'pbe_x', 'pbe_c', 'gga_k', 'vwgtf']
NTERMS = 14
NPARAMS = 13
Q_FFT_COUNT, Q_WORKSPACE_BYTES, Q_FAST_PATH, Q_KERNEL_MS, Q_LAUNCH_COUNT, Q_YPASS_COUNT, Q_GRAPH_REPLAYS, Q_RESIDENT_EVALS = 0, 1, 2, 3, 4, 5, 6, 7
Q_RESIDENT_FALLBACKS = 8
Q_XCHG_CHUNKS = 9
Q_YFWD_FUSED = 10
OPT_GRAPH = 7
OPT_XWAVE = 8
OPT_MIXED_RADIX = 9
OPT_RESIDENT = 10
OPT_TEST_FAULT = 11
OPT_XCHG_CHUNKS = 12
OPT_IPC_WAIT_MS = 13
OPT_YBATCH = 14
OPT_BS_FUSED = 15
OPT_WGC_FOLD = 28

EXPORTS = ['ofdft_create', 'ofdft_destroy', 'ofdft_last_error', 'ofdft_set_cell', 'ofdft_set_terms',
           'ofdft_energy_potential', 'ofdft_energy_grad_chi', 'ofdft_rfftn', 'ofdft_irfftn', 'ofdft_debug_math', 'ofdft_query',
           'ofdft_create_dist', 'ofdft_dist_sumsq', 'ofdft_dist_begin', 'ofdft_dist_stage', 'ofdft_dist_step', 'ofdft_dist_finish', 'ofdft_dist_scalars',
           'ofdft_dist_energies', 'ofdft_dist_chi_grad', 'ofdft_ipc_export', 'ofdft_ipc_attach', 'ofdft_ipc_detach', 'ofdft_dist_closure', 'ofdft_ionic_potential', 'ofdft_ion_electron_forces', 'ofdft_stress', 'ofdft_ion_electron_stress', 'ofdft_ion_ion', 'ofdft_lbfgs_create', 'ofdft_lbfgs_destroy', 'ofdft_lbfgs_last_error', 'ofdft_lbfgs_reset', 'ofdft_lbfgs_direction', 'ofdft_lbfgs_abs_step', 'ofdft_lbfgs_dots',
           'ofdft_lbfgs_commit', 'ofdft_lbfgs_update', 'ofdft_set_option', 'ofdft_set_collectives', 'ofdft_set_profiling', 'ofdft_profile_count', 'ofdft_profile_get']


# callback types of ofdft_set_collectives (include/ofdft_hip.h)
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_ulonglong, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)


class NativeLibraryError(RuntimeError):
    pass


def lib_path(dtype=F64):
    return _build.LIB_F32 if dtype == F32 else _build.LIB


def load(dtype=F64):
    """Load (building if stale and hipcc is present) the HIP engine of one precision: libofdft_hip.so (fp64) or
    libofdft_hip_f32.so (the fp32 build of the same sources, same symbols).  Raises if it cannot."""
    if dtype in _LIBS:
        return _LIBS[dtype]
    env = 'OFDFT_LIB_F32' if dtype == F32 else 'OFDFT_LIB'    # an experiment build for A/B measurements
    path = os.environ.get(env) or lib_path(dtype)
    if os.environ.get(env):
        pass
    elif not os.path.exists(path) or (os.path.exists('/opt/rocm/bin/hipcc') and _build._stale(path)):
        try:
            _build.build(verbose=False)
        except Exception as e:  # noqa: BLE001
            raise NativeLibraryError('%s is missing and could not be built: %r' % (os.path.basename(path), e))
    try:
        lib = C.CDLL(path)
    except OSError as e:
        raise NativeLibraryError('cannot load %s: %s (the HIP engine is required; there is no CPU fallback)'
                                 % (path, e))
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.c_int
    lib.ofdft_create.argtypes = [C.POINTER(vp), ip, ip, ip, ip, ip]
    lib.ofdft_create.restype = ip
    lib.ofdft_destroy.argtypes = [vp]
    lib.ofdft_destroy.restype = None
    lib.ofdft_last_error.argtypes = [vp]
    lib.ofdft_last_error.restype = C.c_char_p
    lib.ofdft_set_cell.argtypes = [vp, dp]
    lib.ofdft_set_cell.restype = ip
    lib.ofdft_set_terms.argtypes = [vp, C.c_uint32, dp, ip]
    lib.ofdft_set_terms.restype = ip
    lib.ofdft_energy_potential.argtypes = [vp, vp, vp, dp, vp, vp]
    lib.ofdft_energy_potential.restype = ip
    lib.ofdft_energy_grad_chi.argtypes = [vp, vp, vp, C.c_double, dp, dp, vp, vp]
    lib.ofdft_energy_grad_chi.restype = ip
    lib.ofdft_rfftn.argtypes = [vp, vp, vp, vp]
    lib.ofdft_rfftn.restype = ip
    lib.ofdft_irfftn.argtypes = [vp, vp, vp, vp]
    lib.ofdft_irfftn.restype = ip
    lib.ofdft_debug_math.argtypes = [vp, ip, vp, vp, C.c_longlong, vp]
    lib.ofdft_debug_math.restype = ip
    lib.ofdft_query.argtypes = [vp, ip, dp]
    lib.ofdft_query.restype = ip
    lib.ofdft_create_dist.argtypes = [C.POINTER(vp), ip, ip, ip, ip, ip, ip, ip]
    lib.ofdft_create_dist.restype = ip
    lib.ofdft_dist_sumsq.argtypes = [vp, vp, ip, dp, vp]
    lib.ofdft_dist_sumsq.restype = ip
    lib.ofdft_dist_begin.argtypes = [vp, vp, ip, C.c_double, C.c_double, vp, vp, vp]
    lib.ofdft_dist_begin.restype = ip
    lib.ofdft_dist_stage.argtypes = [vp, ip, ip, vp, C.POINTER(C.c_ulonglong), C.POINTER(vp), C.POINTER(vp)]
    lib.ofdft_dist_stage.restype = ip
    lib.ofdft_dist_step.argtypes = [vp, ip, ip, ip, vp, C.POINTER(C.c_ulonglong), C.POINTER(vp), C.POINTER(vp)]
    lib.ofdft_dist_step.restype = ip
    lib.ofdft_dist_scalars.argtypes = [vp, C.POINTER(vp)]
    lib.ofdft_dist_scalars.restype = ip
    lib.ofdft_dist_finish.argtypes = [vp, dp, vp]
    lib.ofdft_dist_finish.restype = ip
    lib.ofdft_dist_energies.argtypes = [vp, dp, dp, dp]
    lib.ofdft_dist_energies.restype = ip
    lib.ofdft_dist_chi_grad.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, vp]
    lib.ofdft_dist_chi_grad.restype = ip
    lib.ofdft_ipc_export.argtypes = [vp, vp, C.POINTER(C.c_ulonglong)]
    lib.ofdft_ipc_export.restype = ip
    lib.ofdft_ipc_attach.argtypes = [vp, ip, vp, C.POINTER(C.c_ulonglong)]
    lib.ofdft_ipc_attach.restype = ip
    lib.ofdft_ipc_detach.argtypes = [vp]
    lib.ofdft_ipc_detach.restype = ip
    lib.ofdft_dist_closure.argtypes = [vp, vp, vp, C.c_double, dp, dp, vp, vp, vp]
    lib.ofdft_dist_closure.restype = ip
    lib.ofdft_ionic_potential.argtypes = [vp, dp, ip, dp, dp, ip, C.c_double, ip, vp, ip, vp]
    lib.ofdft_ionic_potential.restype = ip
    lib.ofdft_ion_electron_forces.argtypes = [vp, vp, dp, ip, dp, dp, ip, C.c_double, ip, dp, vp]
    lib.ofdft_ion_electron_forces.restype = ip
    lib.ofdft_stress.argtypes = [vp, vp, dp, vp]
    lib.ofdft_stress.restype = ip
    lib.ofdft_ion_electron_stress.argtypes = [vp, vp, dp, ip, dp, dp, ip, C.c_double, ip, dp, vp]
    lib.ofdft_ion_electron_stress.restype = ip
    lib.ofdft_ion_ion.argtypes = [vp, dp, dp, ip, C.c_double, dp, dp, dp, vp]
    lib.ofdft_ion_ion.restype = ip
    lib.ofdft_lbfgs_create.argtypes = [C.POINTER(vp), C.c_longlong, ip, ip]
    lib.ofdft_lbfgs_create.restype = ip
    lib.ofdft_lbfgs_destroy.argtypes = [vp]
    lib.ofdft_lbfgs_destroy.restype = None
    lib.ofdft_lbfgs_last_error.argtypes = [vp]
    lib.ofdft_lbfgs_last_error.restype = C.c_char_p
    lib.ofdft_lbfgs_reset.argtypes = [vp]
    lib.ofdft_lbfgs_reset.restype = ip
    lib.ofdft_lbfgs_dots.argtypes = [vp, vp, dp, C.POINTER(ip), vp]
    lib.ofdft_lbfgs_dots.restype = ip
    lib.ofdft_lbfgs_commit.argtypes = [vp, ip]
    lib.ofdft_lbfgs_commit.restype = ip
    lib.ofdft_lbfgs_update.argtypes = [vp, dp, dp, C.c_double, C.c_double, vp, vp, dp, vp]
    lib.ofdft_lbfgs_update.restype = ip
    lib.ofdft_lbfgs_direction.argtypes = [vp, dp, ip, ip, dp, dp, dp, dp, C.POINTER(ip), C.POINTER(ip)]
    lib.ofdft_lbfgs_direction.restype = ip
    lib.ofdft_lbfgs_abs_step.argtypes = [vp, dp]
    lib.ofdft_lbfgs_abs_step.restype = ip
    lib.ofdft_set_collectives.argtypes = [vp, A2A_FN, ALLREDUCE_FN, vp]
    lib.ofdft_set_collectives.restype = ip
    lib.ofdft_set_option.argtypes = [vp, ip, C.c_double]
    lib.ofdft_set_option.restype = ip
    lib.ofdft_set_profiling.argtypes = [vp, ip]
    lib.ofdft_set_profiling.restype = ip
    lib.ofdft_profile_count.argtypes = [vp]
    lib.ofdft_profile_count.restype = ip
    lib.ofdft_profile_get.argtypes = [vp, ip, C.c_char_p, ip, dp, C.POINTER(C.c_longlong)]
    lib.ofdft_profile_get.restype = ip
    _LIBS[dtype] = lib
    return lib
