"""ctypes binding of libomds_hip.so (C-ABI declared in include/omds.h).

This is exactly the binding a maintainer of the reference would add to call the MI355X path
(see INTEGRATION.md).  There is no fallback: if the shared library is missing or a call fails,
an exception is raised."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMDS_LIB: another build of the same library (diagnostic / experiment builds under csrc/), never a different backend
LIB_PATH = os.environ.get("OMDS_LIB") or os.path.join(_HERE, "csrc", "libomds_hip.so")

OMDS_MAX_DOF = 7
SWEEP_HIST_LOG_BINS, SWEEP_HIST_RATIO_BINS = 32, 128
SWEEP_HIST_WORDS = 8 + 2 * SWEEP_HIST_LOG_BINS + SWEEP_HIST_RATIO_BINS    # omds.h: OMDS_SWEEP_HIST_WORDS
# float* / int32_t* arguments are declared as plain addresses: numpy's ``a.ctypes.data_as(POINTER(c_float))`` goes through
# ``ctypes.cast`` (~28 us per call: 0.8 ms of a planner iteration through the facade), ``a.ctypes.data`` is an attribute read
F32P = C.c_void_p
I32P = C.c_void_p


class OmdsConfig(C.Structure):
    _fields_ = [("n_dof", C.c_int32), ("n_traj", C.c_int32), ("horizon", C.c_int32), ("n_kernel_max", C.c_int32),
                ("max_obs", C.c_int32), ("n_closest", C.c_int32), ("device", C.c_int32), ("flags", C.c_int32)]


class OmdsParams(C.Structure):
    _fields_ = [("dt", C.c_float), ("dst_thr", C.c_float), ("lin_thr", C.c_float), ("lvel", C.c_float * 5),
                ("ln", C.c_float * 5), ("ltau", C.c_float * 5), ("goal_act_cut", C.c_float),
                ("norm_clamp", C.c_float), ("coll_slow", C.c_float), ("coll_repulse", C.c_float),
                ("softmax_k", C.c_float), ("rbf_p", C.c_float), ("ignored_links", C.c_uint32),
                ("variant", C.c_uint32), ("cost_terms", C.c_uint32)]


FLAG_UNFUSED_STEP = 1
FLAG_TAIL_FORWARD = 4      # the fp32 step's tail keeps its own forward (omds.h: OMDS_FLAG_TAIL_FORWARD)
FLAG_DENSE_PASS1 = 8       # k_pass1 multiplies every k-chunk (omds.h: OMDS_FLAG_DENSE_PASS1; same bits as the per-tile compaction)
FLAG_TWO_KERNEL_STEP = 2   # keep few-obstacle scenes on k_pass1 + k_tail (omds.h: OMDS_FLAG_TWO_KERNEL_STEP)   # omds_config.flags
# omds_params.variant / cost_terms bits (include/omds.h)
VARIANT_KVAL_TIMES_ACT = 1
VARIANT_NO_BASE_MASK = 2
COST_GOAL, COST_COLLISION, COST_JOINT_LIMITS, COST_STAGNATION, COST_FK, COST_ALL = 1, 2, 4, 8, 16, 31


class OmdsError(RuntimeError):
    pass


# name -> (restype, argtypes); every symbol include/omds.h declares
SIGNATURES = {
    "omds_version": (C.c_int, []),
    "omds_default_params": (None, [C.POINTER(OmdsParams)]),
    "omds_create": (C.c_int, [C.POINTER(OmdsConfig), C.POINTER(C.c_void_p)]),
    "omds_destroy": (None, [C.c_void_p]),
    "omds_last_error": (C.c_char_p, [C.c_void_p]),
    "omds_set_mlp": (C.c_int, [C.c_void_p, C.c_int, I32P, C.POINTER(F32P), C.POINTER(F32P), C.c_int, C.c_float]),
    "omds_set_mlp_ex": (C.c_int, [C.c_void_p, C.c_int, I32P, I32P, C.POINTER(F32P), C.POINTER(F32P), C.c_int, C.c_float,
                                  C.c_int, I32P]),
    "omds_set_obstacles": (C.c_int, [C.c_void_p, F32P, C.c_int]),
    "omds_set_ds": (C.c_int, [C.c_void_p, F32P]),
    "omds_set_ds_matrix": (C.c_int, [C.c_void_p, F32P, F32P]),
    "omds_set_ds_seds": (C.c_int, [C.c_void_p, F32P, C.c_int, F32P, F32P, F32P, F32P, F32P, F32P, C.c_float, C.c_float]),
    "omds_set_params": (C.c_int, [C.c_void_p, C.POINTER(OmdsParams)]),
    "omds_set_cost": (C.c_int, [C.c_void_p, F32P, F32P, F32P]),
    "omds_set_policy_samples": (C.c_int, [C.c_void_p, F32P, F32P, F32P, C.c_int]),
    "omds_sample_policy": (C.c_int, [C.c_void_p, F32P, F32P, F32P, C.c_float, C.c_float, C.c_float, C.c_int,
                                     C.c_uint64, C.c_int64]),
    "omds_get_policy_samples": (C.c_int, [C.c_void_p, F32P, F32P, F32P]),
    "omds_propagate": (C.c_int, [C.c_void_p, F32P, C.c_int]),
    "omds_get_rollouts": (C.c_int, [C.c_void_p, F32P, F32P, F32P, F32P, F32P, F32P, F32P]),
    "omds_get_rollout_rows": (C.c_int, [C.c_void_p, I32P, C.c_int, F32P, F32P, F32P, F32P, F32P, F32P, F32P]),
    "omds_dist_grad": (C.c_int, [C.c_void_p, F32P, C.c_int, F32P, F32P, F32P, I32P]),
    "omds_mlp_forward_vjp": (C.c_int, [C.c_void_p, F32P, C.c_int, F32P, F32P, I32P]),
    "omds_mlp_jacobian": (C.c_int, [C.c_void_p, F32P, C.c_int, I32P, C.c_int, F32P, F32P]),
    "omds_cost": (C.c_int, [C.c_void_p, F32P]),
    "omds_cost_eval": (C.c_int, [C.c_void_p, F32P, F32P, C.c_int, F32P]),
    "omds_weighted_update": (C.c_int, [C.c_void_p, C.c_float, C.c_float, F32P, F32P, F32P, I32P, F32P]),
    "omds_weighted_update_eval": (C.c_int, [C.c_void_p, F32P, F32P, F32P, C.c_float, C.c_float, F32P, F32P, F32P, I32P, F32P]),
    "omds_get_qdot": (C.c_int, [C.c_void_p, C.c_int, F32P]),
    "omds_kernel_candidates": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, F32P, F32P, C.c_int, C.c_int,
                                         F32P, I32P, I32P]),
    "omds_cost_sum": (C.c_int, [C.c_void_p, F32P]),
    "omds_red_count": (C.c_int, [C.c_void_p]),
    "omds_local_sums": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_int, F32P]),
    "omds_apply_update": (C.c_int, [C.c_int, C.c_int, C.c_int, F32P, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                    F32P, F32P, F32P, I32P]),
    "omds_comm_probe": (C.c_int, []),
    "omds_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "omds_comm_last_error": (C.c_char_p, []),
    "omds_comm_init_rank": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]),
    "omds_comm_destroy": (C.c_int, [C.c_void_p]),
    "omds_comm_info": (C.c_int, [C.c_void_p, I32P, I32P]),
    "omds_comm_active": (C.c_int, [C.c_void_p]),
    "omds_weighted_update_sharded": (C.c_int, [C.c_void_p, C.c_float, C.c_float, F32P, F32P, F32P, I32P, F32P, F32P,
                                               F32P]),
    "omds_set_screening": (C.c_int, [C.c_void_p, C.c_int, C.c_float]),
    "omds_set_screening_audit": (C.c_int, [C.c_void_p, C.c_int]),
    "omds_screen_audit_stats": (C.c_int, [C.c_void_p, I32P, C.POINTER(C.c_double), F32P, I32P, C.POINTER(C.c_int64)]),
    "omds_screen_fallback_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "omds_screen_order_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), I32P, C.c_int]),
    "omds_pass1_skip_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int64)]),
    "omds_set_screening_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "omds_screen_sweep_hist": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int]),
    "omds_screen_sweep_stats": (C.c_int, [C.c_void_p, I32P, C.POINTER(C.c_int64), F32P]),
    "omds_screen_mindist": (C.c_int, [C.c_void_p, F32P, C.c_int, F32P]),
    "omds_screen_stats": (C.c_int, [C.c_void_p, I32P, F32P, F32P, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "omds_trainer_create": (C.c_int, [C.c_int, C.c_int, I32P, C.c_int, C.POINTER(C.c_void_p)]),
    "omds_trainer_destroy": (None, [C.c_void_p]),
    "omds_trainer_last_error": (C.c_char_p, [C.c_void_p]),
    "omds_trainer_set_weights": (C.c_int, [C.c_void_p, C.POINTER(F32P), C.POINTER(F32P)]),
    "omds_trainer_get_weights": (C.c_int, [C.c_void_p, C.POINTER(F32P), C.POINTER(F32P)]),
    "omds_trainer_set_data": (C.c_int, [C.c_void_p, F32P, F32P, C.c_int]),
    "omds_trainer_step": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, F32P]),
    "omds_trainer_set_val_data": (C.c_int, [C.c_void_p, F32P, F32P, C.c_int]),
    "omds_trainer_eval": (C.c_int, [C.c_void_p, C.c_int, F32P, F32P]),
    "omds_trainer_get_optimizer_state": (C.c_int, [C.c_void_p, C.POINTER(F32P), C.POINTER(F32P), C.POINTER(F32P), C.POINTER(F32P),
                                                   C.POINTER(C.c_int64)]),
    "omds_trainer_set_optimizer_state": (C.c_int, [C.c_void_p, C.POINTER(F32P), C.POINTER(F32P), C.POINTER(F32P), C.POINTER(F32P), C.c_int64]),
    "omds_prof_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "omds_prof_reset": (C.c_int, [C.c_void_p]),
    "omds_prof_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "omds_prof_read_ex": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                    C.POINTER(C.c_char_p)]),
    "omds_sync": (C.c_int, [C.c_void_p]),
    "omds_device_count": (C.c_int, [C.POINTER(C.c_int32)]),
}

# include/omds_test.h: exported by libomds_hip_test.so only (the product library does not have them)
TEST_HOOK_SIGNATURES = {
    "omds_screen_debug_corrupt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_float]),
    "omds_debug_force_tile_rows": (C.c_int, [C.c_int, C.c_int]),
    "omds_debug_trainer_general_gemm": (C.c_int, [C.c_int]),
    "omds_test_pack_mlp": (C.c_int, [C.c_int, C.c_int, I32P, I32P, C.POINTER(F32P), C.POINTER(F32P), C.c_int, C.c_float, C.c_int, I32P,
                                     C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]),
}
TEST_LIB_PATH = os.path.join(_HERE, "csrc", "libomds_hip_test.so")

ABI_VERSION = 500      # omds_version() of the library this binding was written against
_libs = {}             # path -> bound CDLL


def _autobuild(path=LIB_PATH):
    """`make` in csrc/, serialised across processes (torchrun starts one rank per GPU at once) by an exclusive
    lock; the Makefile links to a temporary name and renames, so no rank can dlopen a half-written library."""
    import fcntl
    import subprocess
    csrc = os.path.join(_HERE, "csrc")
    with open(os.path.join(csrc, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if os.path.exists(path):     # another rank built it while this one waited
                return
            r = subprocess.run(["make", "-C", csrc, "-j4"], capture_output=True, text=True)
            if r.returncode != 0:
                raise OmdsError(f"building {path} failed (make exit {r.returncode}):\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}")
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def load(path=None, extra_signatures=None):
    """dlopen the in-tree library (or another build of it at ``path``) and bind every entry point.  Fails loudly when it
    is missing.  A handle created by one loaded library must only be passed to that library (``Engine`` keeps its own)."""
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path) and os.environ.get("OMDS_NO_AUTOBUILD") != "1":
        # the library is built in-tree by __graft_entry__.build(); if a checkout arrives without it, build it
        # here (hipcc, gfx950) -- still no fallback of any kind: without the library nothing runs
        _autobuild(path)
    if not os.path.exists(path):
        raise OmdsError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        f"or `make -C optimalmodulationds_amd/csrc` (there is no CPU fallback)")
    lib = C.CDLL(path)
    try:
        lib.omds_version.restype = C.c_int
        have = int(lib.omds_version())
    except AttributeError:
        have = -1
    if have != ABI_VERSION:     # a stale build of an older checkout: say so instead of failing on a missing symbol
        raise OmdsError(f"{path} is ABI version {have}, this package needs {ABI_VERSION}: rebuild it with "
                        f"`make -C optimalmodulationds_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`)")
    for name, (res, args) in dict(SIGNATURES, **(extra_signatures or {})).items():
        fn = getattr(lib, name)      # AttributeError = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib


def load_test_hooks():
    """libomds_hip_test.so: the product's objects plus the two hooks of include/omds_test.h (damage the screening inputs, force
    a tile shape).  For tests only: ``Engine(..., lib=load_test_hooks())``."""
    return load(TEST_LIB_PATH, TEST_HOOK_SIGNATURES)


def f32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if shape is not None:
        a = a.reshape(shape)
    return a


def fptr(a):
    """Address of a C-contiguous float32 array (None -> NULL).  The argtypes are plain addresses, so ctypes neither checks the
    element type nor keeps ``a`` alive: pass a NAMED array that outlives the call -- never an inline temporary such as
    ``fptr(f32(x))`` or ``fptr(arr[::2].copy())`` -- and convert with ``f32`` first."""
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], "fptr: float32, C-contiguous arrays only"
    return a.ctypes.data


def iptr(a):
    """Address of a C-contiguous int32 array (None -> NULL); the same named-array rule as ``fptr``."""
    if a is None:
        return None
    assert a.dtype == np.int32 and a.flags["C_CONTIGUOUS"], "iptr: int32, C-contiguous arrays only"
    return a.ctypes.data


def check(ctx, rc, lib=None):
    if rc != 0:
        msg = (lib or load()).omds_last_error(ctx)
        raise OmdsError(f"omds error {rc}: {msg.decode() if msg else '?'}")


def device_count() -> int:
    """HIP devices visible to this process (omds_device_count; 0 on a box without a GPU)."""
    n = C.c_int32(0)
    rc = load().omds_device_count(C.byref(n))
    if rc != 0:
        raise OmdsError(f"omds_device_count failed ({rc})")
    return int(n.value)


def comm_probe() -> str:
    """'' when RCCL can be loaded in this process, else the loader's message (omds_comm_probe)."""
    lib = load()
    return "" if lib.omds_comm_probe() == 0 else (lib.omds_comm_last_error() or b"RCCL not available").decode()


def default_params() -> OmdsParams:
    p = OmdsParams()
    load().omds_default_params(C.byref(p))
    return p
