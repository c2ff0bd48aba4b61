"""ctypes binding of oracle/libgan_mpc_step.so, the plain-C restatement of one step (TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it)."""

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GMPC_STEP_LIB: another build of the same library (the sanitizer build of tests/test_oracle_c.py)
LIB_PATH = os.environ.get("GMPC_STEP_LIB") or os.path.join(_HERE, "libgan_mpc_step.so")
_lib = None
_F = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_I = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: make -C oracle (or __graft_entry__.build())")
        _lib = C.CDLL(LIB_PATH)
        _lib.gmpc_c_threads.restype = C.c_int
        _lib.gmpc_c_trajectories.restype = C.c_int
        _lib.gmpc_c_critic_loss_grad.restype = C.c_int
    return _lib


def _flat(layers):
    return np.ascontiguousarray(np.concatenate([t.reshape(-1) for W, b in layers for t in (W, b)]), np.float32)


def _dims(layers):
    return np.asarray([layers[0][0].shape[0]] + [W.shape[1] for W, _ in layers], np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def threads():
    return load().gmpc_c_threads()


def trajectories(dyn, cmlp, mpc_w, goal, x0, U, backward=True):
    """rollout + costs (+ backward pass) -> dict(X, costs[, K, k, grad, adjoints])"""
    lib = load()
    B, T, m = U.shape
    n = x0.shape[1]
    f32 = lambda a: np.ascontiguousarray(a, np.float32)
    out = dict(X=np.empty((B, T + 1, n), np.float32), costs=np.empty((B, T + 1), np.float32))
    if backward:
        out.update(K=np.empty((B, T, m, n), np.float32), k=np.empty((B, T, m), np.float32),
                   grad=np.empty((B, T, m), np.float32), adjoints=np.empty((B, T + 1, n), np.float32))
    dd, cd = _dims(dyn), _dims(cmlp)
    df, cf, w, x0, U, goal = _flat(dyn), _flat(cmlp), f32(mpc_w), f32(x0), f32(U), f32(goal)
    rc = lib.gmpc_c_trajectories(B, n, m, T, len(dyn), _p(dd), _p(df), len(cmlp), _p(cd), _p(cf), _p(w), _p(x0),
                                 _p(U), _p(goal), _p(out["X"]), _p(out["costs"]), _p(out.get("K")), _p(out.get("k")),
                                 _p(out.get("grad")), _p(out.get("adjoints")))
    if rc != 0:
        raise RuntimeError(f"gmpc_c_trajectories failed ({rc})")
    return out


def critic_flat(cr):
    return np.ascontiguousarray(np.concatenate([cr["Wx"].reshape(-1), cr["Wh"].reshape(-1), cr["b"].reshape(-1)]
                                               + [t.reshape(-1) for W, b in cr["head"] for t in (W, b)]), np.float32)


def critic_loss_grad(cr_flat, n, F, head_dims, xseq, label):
    """-> (loss SUM, gradient SUM over the sequences) in the flat critic layout"""
    lib = load()
    Bc, T1, _ = xseq.shape
    hd = np.asarray(head_dims, np.int32)
    xs, lab = np.ascontiguousarray(xseq, np.float32), np.ascontiguousarray(label, np.float32)
    g = np.empty(cr_flat.size, np.float32)
    loss = C.c_float()
    rc = lib.gmpc_c_critic_loss_grad(Bc, T1, n, F, len(hd) - 1, _p(hd), _p(cr_flat), _p(xs), _p(lab), C.byref(loss),
                                     _p(g))
    if rc != 0:
        raise RuntimeError(f"gmpc_c_critic_loss_grad failed ({rc})")
    return loss.value, g


def adam_clip(p, grad, m, v, grad_scale, step, lr, max_norm=100.0, b1=0.9, b2=0.999, eps=1e-8):
    lib = load()
    lib.gmpc_c_adam_clip(C.c_long(p.size), _p(p), _p(grad), _p(m), _p(v), C.c_float(grad_scale), C.c_int(step),
                         C.c_double(lr), C.c_double(max_norm), C.c_double(b1), C.c_double(b2), C.c_double(eps))
