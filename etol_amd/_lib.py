"""ctypes binding of libemi355x.so (the C ABI declared in include/emi355x.h).

There is no CPU fallback: if the shared library is missing, or it cannot find
a gfx950 device when a context is created, the caller gets an exception.
"""
import ctypes as C
import os

# torch ships its own libamdhip64 (same soname as /opt/rocm's).  Import torch
# first so that one HIP runtime serves both torch and libemi355x in this process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libemi355x.so")

EMI_OK = 0
STATUS = {0: "EMI_OK", 1: "EMI_ERR_ARG", 2: "EMI_ERR_STATE", 3: "EMI_ERR_HIP",
          4: "EMI_ERR_NO_DEVICE", 5: "EMI_ERR_UNSUPPORTED", 6: "EMI_ERR_COMM"}

MODEL_POINTMASS2D, MODEL_QUADROTOR2D, MODEL_FIXEDWING12 = 0, 1, 2
MODEL_SOURCE = 100   # installed by emi_set_model_source
PATH_ELLIPSE, PATH_DISC, PATH_TRACK = 0, 1, 2
PATH_REC = 8
EVAL_NODES, EVAL_DEFECT, EVAL_ALL, EVAL_NOJAC = 1, 2, 3, 4


class EmiError(RuntimeError):
    pass


class Layout(C.Structure):
    _fields_ = [("model", C.c_int), ("ns", C.c_int), ("nc", C.c_int), ("np", C.c_int),
                ("M", C.c_int), ("B", C.c_int), ("nres", C.c_int), ("nvals", C.c_int),
                ("nhess", C.c_int), ("real_bytes", C.c_int), ("px", C.c_int), ("py", C.c_int),
                ("t0", C.c_double), ("tf", C.c_double)]


class PassPlan(C.Structure):
    """emi_pass_plan_t: what the default dispatch does with a batch (include/emi355x.h, emi_plan_pass)"""
    _fields_ = [(n, C.c_int) for n in ("one_launch", "sw", "ksplit", "ring_stages", "cpart", "cx", "mfma_workgroups", "store_mode",
                                       "block_order", "tiles16", "piece", "tail", "k_tile", "column_tiles", "k_halves")]


_P = C.c_void_p
_D = C.POINTER(C.c_double)
_I = C.POINTER(C.c_int)

# every symbol include/emi355x.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "emi_abi_version": (C.c_int, []),
    "emi_status_string": (C.c_char_p, [C.c_int]),
    "emi_device_count": (C.c_int, [_I]),
    "emi_lgl": (C.c_int, [C.c_int, _D, _D, _D]),
    "emi_model_dims": (C.c_int, [C.c_int, _I, _I, _I]),
    "emi_edge_ellipse": (C.c_int, [C.c_double] * 4 + [_D]),
    "emi_track_centres": (C.c_int, [C.c_int, _D, _D, _D, C.c_int, _D, _D, _D]),
    "emi_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "emi_create_f32": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "emi_destroy": (C.c_int, [_P]),
    "emi_last_error": (C.c_char_p, [_P]),
    "emi_set_stream": (C.c_int, [_P, _P]),
    "emi_get_stream": (C.c_int, [_P, C.POINTER(_P)]),
    "emi_synchronize": (C.c_int, [_P]),
    "emi_set_mesh": (C.c_int, [_P, C.c_int, _D, _D, _D, C.c_double, C.c_double]),
    "emi_set_model": (C.c_int, [_P, C.c_int, _D, C.c_int, C.c_int]),
    "emi_set_model_source": (C.c_int, [_P, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, _I, C.c_int, _D, C.c_int, C.c_int]),
    "emi_check_model_source": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]),
    "emi_kkt_factor": (C.c_int, [_P, _D, _D, C.POINTER(C.c_ubyte), C.c_double, C.POINTER(C.c_int)]),
    "emi_kkt_solve": (C.c_int, [_P, _D, C.c_int]),
    "emi_kkt_last_regularisation": (C.c_int, [_P, _D, _D]),
    "emi_kkt_factor_batch": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_D), C.POINTER(_D), C.POINTER(C.POINTER(C.c_ubyte)), _D, _I]),
    "emi_kkt_solve_batch": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_D)]),
    "emi_kkt_is_schur": (C.c_int, [_P]),
    "emi_kkt_solve_refined": (C.c_int, [_P, _D, C.c_double, C.c_int, _D, _I, _I, _I]),
    "emi_kkt_solve_refined_batch": (C.c_int, [C.c_int, C.POINTER(_P), C.POINTER(_D), _D, C.c_int, _D, _I, _I, _I]),
    "emi_kkt_lowrank": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), _D, _D, C.POINTER(C.c_int)]),
    "emi_set_batch": (C.c_int, [_P, C.c_int]),
    "emi_set_path": (C.c_int, [_P, C.c_int, C.c_int, _D, C.c_int, C.c_int]),
    "emi_set_tracks": (C.c_int, [_P, C.c_int, C.c_int, _D, _D]),
    "emi_get_layout": (C.c_int, [_P, C.POINTER(Layout)]),
    "emi_jac_structure": (C.c_int, [_P, _I, _I]),
    "emi_dev_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "emi_dev_free": (C.c_int, [_P, _P]),
    "emi_h2d": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "emi_d2h": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "emi_eval_dev": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_uint]),
    "emi_eval_host": (C.c_int, [_P, _D, _D, _D, _D, _D, C.c_uint]),
    "emi_hess_dev": (C.c_int, [_P, _P, _P, _P, _P, C.c_double, _P]),
    "emi_hess_host": (C.c_int, [_P, _D, _D, _D, _D, C.c_double, _D]),
    "emi_timer_start": (C.c_int, [_P]),
    "emi_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "emi_profile_enable": (C.c_int, [_P, C.c_int]),
    "emi_profile_read": (C.c_int, [_P, C.POINTER(C.c_float), _I, C.POINTER(C.c_float), _I, C.POINTER(C.c_float), _I]),
    "emi_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "emi_last_path": (C.c_int, [_P, _I]),
    "emi_plan_pass": (C.c_int, [_P, C.c_int, C.POINTER(PassPlan)]),
    "emi_last_defect_kernel": (C.c_char_p, [_P]),
    "emi_debug_pass_roles": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]),
    "emi_debug_tile_order": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "emi_set_delays": (C.c_int, [_P, C.c_int, C.c_int, C.c_double]),
    "emi_get_delays": (C.c_int, [_P, _I, _I, _I]),
    "emi_delay_matrix": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]),
    "emi_debug_tile_order2": (C.c_int, [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "emi_comm_unique_id": (C.c_int, [_P]),
    "emi_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(_P)]),
    "emi_comm_gather": (C.c_int, [_P, _P, _P, C.c_size_t, C.c_int, _P]),
    "emi_comm_destroy": (C.c_int, [_P]),
    "emi_comm_last_error": (C.c_char_p, [_P]),
}

_lib = None


def load():
    """Load libemi355x.so (once) and type every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmiError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(or `make lib`) first; there is no CPU fallback")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError here = ABI mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, ctx=None, what=""):
    if status == EMI_OK:
        return
    lib = load()
    msg = lib.emi_status_string(status).decode()
    if ctx:
        detail = lib.emi_last_error(ctx).decode()
        if detail:
            msg += ": " + detail
    raise EmiError(f"{what or 'libemi355x'}: {STATUS.get(status, status)} ({msg})")
