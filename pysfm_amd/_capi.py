"""ctypes binding of libpysfm_ba.so (include/pysfm_ba.h).

There is no CPU fallback: if the HIP library is missing, or no MI355X is
visible, every entry point raises.  Build the library with
``make -C pysfm_amd/csrc`` or ``python -c "import __graft_entry__ as g; g.build()"``.
"""
import ctypes as C
import os

import numpy as np

LIB_NAME = 'libpysfm_ba.so'
TRIAL_PARTIALS = 2048          # BA_TRIAL_PARTIALS of include/pysfm_ba.h
LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), LIB_NAME)

BA_OK = 0
BA_ERR_INVALID_ARG, BA_ERR_NO_DEVICE, BA_ERR_HIP, BA_ERR_STATE, BA_ERR_SINGULAR, BA_ERR_NOMEM = \
    -1, -2, -3, -4, -5, -6
SENSOR_GAUSS, SENSOR_CAUCHY, SENSOR_HUBER, SENSOR_TABLE = 0, 1, 2, 3
PARAMS_CUR, PARAMS_TRIAL = 0, 1
KERNEL_IDS = ('cost', 'linearize', 'point_invert', 'schur_init', 'schur_pairs', 'backsub',
              'update', 'flatten', 'band_solve', 'eval', 'camera_blocks', 'triangulate',
              'bcr_assemble', 'bcr_eliminate', 'bcr_backsolve', 'dense_solve', 'border_schur', 'border_solve', 'bcr_refine', 'pcg_solve')
K_COUNT = len(KERNEL_IDS)
INFO_KEYS = ('points_permuted', 'obs_permuted', 'groups', 'mfma_groups', 'point_groups', 'max_track_len',
             'half_bandwidth', 'schur_mfma', 'schur_groups', 'lds_window_rows', 'pair_units', 'schur_kernel',
             'mfma_points_per_batch_cap', 'mfma_k_rows', 'cameras_permuted', 'caller_half_bandwidth', 'border_cameras', 'linearizations_reused', 'solves_refined', 'packed_store')      # BA_INFO_*
SOLVE_KINDS = ('none', 'bcr', 'bcr_wide', 'band', 'dense_cholesky', 'bcr_lu', 'bcr_big', 'band_lu', 'pcg')
SOLVE_TIMED_OUT = 0x7f000001            # BA_SOLVE_TIMED_OUT of include/pysfm_ba.h
SOLVE_STALLED = 0x7f000002              # BA_SOLVE_STALLED: conjugate gradients out of iterations
DIST_INFO_KEYS = ('on', 'cams_per_node', 'nodes', 'nodes_per_rank', 'node_lo', 'node_hi', 'cam_lo', 'cam_hi',
                  'exchange1_doubles', 'exchange2_doubles', 'exchange3_doubles', 'separators')      # ba_dist_info

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)
_h = C.c_void_p

RESIDENT_MAX_TRIALS = 1000       # BA_RESIDENT_MAX_TRIALS
RESIDENT_DONE, RESIDENT_LOG_FULL, RESIDENT_NOT_POSITIVE_DEFINITE, RESIDENT_SINGULAR_POINT, RESIDENT_TIMED_OUT = 0, 1, 2, 3, 4


class ResidentLog(C.Structure):
    """ba_resident_log of include/pysfm_ba.h: what ba_lm_resident did, trial by trial."""
    _fields_ = [('ntrials', C.c_int32), ('nsteps', C.c_int32), ('converged', C.c_int32), ('in_step', C.c_int32),
                ('exit_reason', C.c_int32), ('exit_info', C.c_int32), ('accepted', C.c_int32), ('have_cost0', C.c_int32),
                ('damping', C.c_double), ('cost0', C.c_double), ('cur_cost', C.c_double), ('reserved', C.c_double),
                ('trial_damping', C.c_double * RESIDENT_MAX_TRIALS), ('trial_cost', C.c_double * RESIDENT_MAX_TRIALS),
                ('trial_accepted', C.c_int32 * RESIDENT_MAX_TRIALS)]


# name -> (restype, argtypes); one entry per function declared in include/pysfm_ba.h
PROTOTYPES = {
    'ba_create': (C.c_int, [C.c_int, C.POINTER(_h)]),
    'ba_destroy': (C.c_int, [_h]),
    'ba_last_error': (C.c_char_p, [_h]),
    'ba_set_stream': (C.c_int, [_h, C.c_void_p]),
    'ba_synchronize': (C.c_int, [_h]),
    'ba_set_option': (C.c_int, [_h, C.c_char_p, C.c_char_p]),
    'ba_debug_poison': (C.c_int, [_h]),
    'ba_set_problem': (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int64, _ip, _ip, _dp, _dp, _ip, _bp]),
    'ba_problem_info': (C.c_int, [_h, C.POINTER(C.c_int64), C.c_int32]),
    'ba_order_cameras': (C.c_int, [C.c_int32, C.c_int32, _ip, _ip, _ip, _ip]),
    'ba_set_camera_layout': (C.c_int, [_h, _ip, C.c_int32]),
    'ba_get_camera_layout': (C.c_int, [_h, _ip, C.POINTER(C.c_int32)]),
    'ba_set_pattern_lists': (C.c_int, [_h, C.c_int32, _ip, _ip]),
    'ba_pcg_info': (C.c_int, [_h, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    'ba_plan_camera_layout': (C.c_int, [C.c_int32, C.c_int32, _ip, _ip, _ip, C.c_int32, _ip, _ip, _ip]),
    'ba_set_sensor': (C.c_int, [_h, C.c_int, _dp, C.c_int]),
    'ba_set_params': (C.c_int, [_h, C.c_int, _dp, _dp, _dp]),
    'ba_get_params': (C.c_int, [_h, C.c_int, _dp, _dp, _dp]),
    'ba_swap_params': (C.c_int, [_h]),
    'ba_cost': (C.c_int, [_h, C.c_int, _dp]),
    'ba_eval_observations': (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp]),
    'ba_eval_sensor': (C.c_int, [_h, C.c_int64, _dp, _dp, _dp]),
    'ba_linearize': (C.c_int, [_h, C.c_int, C.c_int]),
    'ba_get_blocks': (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp]),
    'ba_schur': (C.c_int, [_h, C.c_int, C.c_double, C.c_double]),
    'ba_get_reduced': (C.c_int, [_h, _dp, _dp]),
    'ba_get_point_inverses': (C.c_int, [_h, _dp]),
    'ba_reduced_device_ptrs': (C.c_int, [_h, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    'ba_bind_reduced_buffers': (C.c_int, [_h, C.c_void_p, C.c_void_p]),
    'ba_reduced_layout': (C.c_int, [_h, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    'ba_solve_reduced': (C.c_int, [_h, _bp, C.POINTER(C.c_int32)]),
    'ba_last_solve_kind': (C.c_int, [_h]),
    'ba_get_solution': (C.c_int, [_h, _dp]),
    'ba_set_solution': (C.c_int, [_h, _dp]),
    'ba_flatten_reduced': (C.c_int, [_h, _ip, C.c_int32, C.c_void_p, C.c_void_p]),
    'ba_backsubstitute': (C.c_int, [_h, C.c_int, _dp, _dp]),
    'ba_apply_update': (C.c_int, [_h, C.c_int, C.c_int, _dp, _dp]),
    'ba_lm_trial': (C.c_int, [_h, C.c_double, C.c_double, _bp, _dp, C.POINTER(C.c_int32)]),
    'ba_bind_trial_result': (C.c_int, [_h, C.c_void_p]),
    'ba_lm_resident_fits': (C.c_int, [_h]),
    'ba_lm_resident_trace': (C.c_int, [_h, C.POINTER(C.c_int64)]),
    'ba_lm_resident_debug': (C.c_int, [_h, _dp, _dp, _dp]),
    'ba_lm_resident': (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double,
                                 _bp, C.POINTER(ResidentLog)]),
    'ba_lm_resident_begin': (C.c_int, [_h, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_double, C.c_double, C.c_double, _bp]),
    'ba_lm_resident_end': (C.c_int, [_h, C.POINTER(ResidentLog)]),
    'ba_set_dense_visibility': (C.c_int, [_h, C.c_int32]),
    'ba_measure_copy_bandwidth': (C.c_int, [_h, C.c_int64, C.c_int32, C.POINTER(C.c_double)]),
    'ba_set_min_half_bandwidth': (C.c_int, [_h, C.c_int32]),
    'ba_comm_load': (C.c_int, [C.c_char_p]),
    'ba_comm_unique_id': (C.c_int, [C.c_void_p]),
    'ba_comm_init': (C.c_int, [_h, C.c_void_p, C.c_int32, C.c_int32]),
    'ba_comm_destroy': (C.c_int, [_h]),
    'ba_comm_allreduce_reduced': (C.c_int, [_h]),
    'ba_comm_allreduce_sum': (C.c_int, [_h, C.POINTER(C.c_double), C.c_int32]),
    'ba_lm_trial_begin': (C.c_int, [_h, C.c_double, C.c_double]),
    'ba_lm_trial_end': (C.c_int, [_h, _bp, C.POINTER(C.c_int32)]),
    'ba_lm_trial_finish': (C.c_int, [_h]),
    'ba_dist_plan': (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    'ba_dist_enable': (C.c_int, [_h, C.c_int32, C.c_int32]),
    'ba_dist_info': (C.c_int, [_h, C.POINTER(C.c_int64), C.c_int32]),
    'ba_dist_bind_exchange': (C.c_int, [_h, C.c_void_p, C.c_int64]),
    'ba_dist_stage': (C.c_int, [_h, C.c_int32, _bp, C.POINTER(C.c_int64)]),
    'ba_triangulate': (C.c_int, [_h, C.c_int, C.c_double, _dp]),
    'ba_enable_timing': (C.c_int, [_h, C.c_int]),
    'ba_set_timing_mask': (C.c_int, [_h, C.c_uint64]),
    'ba_set_timing_stride': (C.c_int, [_h, C.c_int32]),
    'ba_get_timings': (C.c_int, [_h, _dp, C.POINTER(C.c_int64), C.c_int]),
    'ba_kernel_name': (C.c_char_p, [C.c_int]),
    'ba_version': (C.c_char_p, []),
}


class HipLibraryMissing(RuntimeError):
    """libpysfm_ba.so is not built / not loadable.  There is no CPU path."""


class HipDeviceError(RuntimeError):
    """The HIP library reported an error (see message)."""


_lib = None


def load():
    """Load libpysfm_ba.so and declare every prototype.  Raises HipLibraryMissing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            '%s not found. pysfm_amd has no CPU fallback: build the HIP library with '
            '`make -C pysfm_amd/csrc` (hipcc --offload-arch=gfx950).' % LIB_PATH)
    # torch bundles its own HIP runtime (SONAME libamdhip64.so.7).  Import it first so
    # that libpysfm_ba.so binds to the SAME runtime instance: two HIP runtimes in one
    # process cannot share streams, device pointers or even see the GPU together.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise HipLibraryMissing('cannot load %s: %s' % (LIB_PATH, e))
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise HipLibraryMissing('%s does not export %s (stale build?)' % (LIB_PATH, name))
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def dptr(a):
    return None if a is None else a.ctypes.data_as(_dp)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_ip)


def bptr(a):
    return None if a is None else a.ctypes.data_as(_bp)


def f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)
