"""ctypes binding of libgan_mpc_amd.so (include/gan_mpc_amd.h).

The library is built in-tree by ``__graft_entry__.build()`` (``make -C gan_mpc_amd/csrc``).  There is
no CPU fallback: if the shared object is missing, or no HIP device is visible when a context is
created, the caller gets an exception.
"""

import ctypes as C
import os

GMPC_MAX_LAYERS = 8
_HERE = os.path.dirname(os.path.abspath(__file__))
# GMPC_LIB: an alternative build of the same library (A/B timing of kernel variants); default: the in-tree one
LIB_PATH = os.environ.get("GMPC_LIB") or os.path.join(_HERE, "libgan_mpc_amd.so")


class GmpcError(RuntimeError):
    pass


class Shape(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("m", C.c_int), ("T", C.c_int),
        ("dyn_layers", C.c_int), ("dyn_dims", C.c_int * (GMPC_MAX_LAYERS + 1)),
        ("cost_layers", C.c_int), ("cost_dims", C.c_int * (GMPC_MAX_LAYERS + 1)),
        ("lstm_features", C.c_int),
        ("head_layers", C.c_int), ("head_dims", C.c_int * (GMPC_MAX_LAYERS + 1)),
        ("dyn_lstm_features", C.c_int), ("x_size", C.c_int),
    ]


class IlqrOpts(C.Structure):
    _fields_ = [
        ("maxiter", C.c_int),
        ("grad_norm_threshold", C.c_float),
        ("relative_grad_norm_threshold", C.c_float),
        ("obj_step_threshold", C.c_float),
        ("inputs_step_threshold", C.c_float),
        ("make_psd", C.c_int),
        ("psd_delta", C.c_float),
        ("alpha_0", C.c_float),
        ("alpha_min", C.c_float),
    ]


class ExpertShape(C.Structure):
    _fields_ = [
        ("lstm_features", C.c_int), ("head_layers", C.c_int),
        ("head_dims_x", C.c_int * (GMPC_MAX_LAYERS + 1)),
        ("head_dims_u", C.c_int * (GMPC_MAX_LAYERS + 1)),
    ]


class Leaf(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("offset", C.c_long), ("rows", C.c_int), ("cols", C.c_int),
                ("ld", C.c_int)]


_P = C.c_void_p
# name -> (restype, argtypes); exactly the entry points declared in include/gan_mpc_amd.h
SIGNATURES = {
    "gmpc_last_error": (C.c_char_p, []),
    "gmpc_version": (C.c_char_p, []),
    "gmpc_param_count": (C.c_long, [C.POINTER(Shape), C.c_int]),
    "gmpc_pack_layout": (C.c_int, [C.POINTER(Shape), C.c_int, C.POINTER(Leaf), C.c_int]),
    "gmpc_get_cost": (C.c_int, [_P, C.c_int, _P, _P, _P, C.c_int, _P, _P]),
    "gmpc_predict": (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    "gmpc_comm_unique_id": (C.c_int, [C.c_char_p]),
    "gmpc_comm_init": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p]),
    "gmpc_comm_world": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gmpc_allreduce_grads": (C.c_int, [_P, _P, C.c_long, _P]),
    "gmpc_create": (C.c_int, [C.POINTER(Shape), C.c_int, C.c_int, C.POINTER(_P)]),
    "gmpc_destroy": (C.c_int, [_P]),
    "gmpc_set_params": (C.c_int, [_P, _P, _P, _P, _P]),
    "gmpc_rollout_cost": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "gmpc_lqr_backward": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gmpc_lqr_backward_after_rollout": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "gmpc_ilqr_solve": (C.c_int, [_P, C.c_int, _P, _P, _P, C.POINTER(IlqrOpts), _P, _P, _P, _P, _P,
                                  _P, _P]),
    "gmpc_bilevel_grad": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_float, _P, _P, _P]),
    "gmpc_upper_loss": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P]),
    "gmpc_expert_rollout": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(ExpertShape), _P, _P, _P, _P, _P]),
    "gmpc_expert_param_count": (C.c_long, [C.c_int, C.POINTER(ExpertShape)]),
    "gmpc_dynamics_loss_grad": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, C.c_double, C.c_int, _P, _P, _P]),
    "gmpc_polyak": (C.c_int, [_P, C.c_long, _P, _P, C.c_double, _P, _P]),
    "gmpc_critic_loss_grad": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "gmpc_critic_score_vjp": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "gmpc_adam_clip_step": (C.c_int, [_P, C.c_long, _P, _P, _P, _P, C.c_float, C.c_int, C.c_double,
                                      C.c_double, C.c_double, C.c_double, C.c_double, _P]),
    "gmpc_bgemm_tn": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_float, C.c_float,
                                _P]),
    "gmpc_linesearch_candidates": (C.c_long, [_P]),
    "gmpc_linesearch_stats": (C.c_int, [_P, C.POINTER(C.c_long), C.c_int]),
    "gmpc_profile_enable": (C.c_int, [_P, C.c_int]),
    "gmpc_set_linearize_event": (C.c_int, [_P, _P]),
    "gmpc_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "gmpc_profile_kernel_name": (C.c_char_p, [_P, C.c_int]),
    "gmpc_debug_buffer": (_P, [_P, C.c_int]),
    "gmpc_debug_buffer_count": (C.c_long, [_P, C.c_int]),
}

_lib = None


def load():
    """Load the shared object and bind every declared symbol (raises if one is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GmpcError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (make -C gan_mpc_amd/csrc).  gan_mpc_amd has no CPU fallback.")
    # The process must hold ONE HIP runtime: torch ships its own libamdhip64 and owns the device
    # buffers this library is handed, so it is loaded first and our shared object binds to that copy.
    # (Loaded the other way round, /opt/rocm's runtime comes in as a second instance that sees no
    # device and could not use torch's pointers anyway.)
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise GmpcError(f"libgan_mpc_amd error {rc}: {load().gmpc_last_error().decode()}")
