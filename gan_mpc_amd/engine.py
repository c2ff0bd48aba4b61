"""Thin host wrapper around one ``gmpc_ctx``: torch tensors own the device memory and the stream,
every method is a single call through the C ABI (include/gan_mpc_amd.h).

torch is plumbing only here (device buffers, ``torch.cuda.current_stream``); no arithmetic on the
hot path is done by torch.
"""

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import IlqrOpts, Shape

# trajax iLQR keywords exactly as the reference passes them (policy/eval.py:10-20)
TRAJAX_iLQR_KWARGS = {
    "maxiter": 100,
    "grad_norm_threshold": 1e-4,
    "relative_grad_norm_threshold": 0.0,
    "obj_step_threshold": 0.0,
    "inputs_step_threshold": 0.0,
    "make_psd": False,
    "psd_delta": 0.0,
    "alpha_0": 1.0,
    "alpha_min": 0.00005,
}


def make_shape(n, m, T, dyn_dims, cost_dims, lstm_features=0, head_dims=None, dyn_lstm=0, x_size=0):
    """dyn_lstm = F > 0: the LSTM dynamics variant -- n = x_size + 2F is the size of xc = [x, c, h],
    dyn_dims the relu tail [F, hidden..., x_size] (see include/gan_mpc_amd.h)."""
    s = Shape()
    s.n, s.m, s.T = int(n), int(m), int(T)
    s.dyn_lstm_features, s.x_size = int(dyn_lstm), int(x_size)
    s.dyn_layers = len(dyn_dims) - 1
    s.cost_layers = len(cost_dims) - 1
    for i, d in enumerate(dyn_dims):
        s.dyn_dims[i] = int(d)
    for i, d in enumerate(cost_dims):
        s.cost_dims[i] = int(d)
    s.lstm_features = int(lstm_features)
    if lstm_features:
        head_dims = list(head_dims or [lstm_features, 1])
        s.head_layers = len(head_dims) - 1
        for i, d in enumerate(head_dims):
            s.head_dims[i] = int(d)
    return s


def make_expert_shape(lstm_features, head_dims_x, head_dims_u):
    """gmpc_expert_shape: y width (= lstm_features, or the first dense width of the MLP variant),
    hidden widths, then n / m."""
    es = _lib.ExpertShape()
    es.lstm_features = int(lstm_features)
    assert len(head_dims_x) == len(head_dims_u)
    es.head_layers = len(head_dims_x) - 1
    for i, (dx, du) in enumerate(zip(head_dims_x, head_dims_u)):
        es.head_dims_x[i], es.head_dims_u[i] = int(dx), int(du)
    return es


def make_opts(kwargs=None):
    kw = dict(TRAJAX_iLQR_KWARGS)
    if kwargs:
        kw.update(kwargs)
    o = IlqrOpts()
    o.maxiter = int(kw["maxiter"])
    o.grad_norm_threshold = float(kw["grad_norm_threshold"])
    o.relative_grad_norm_threshold = float(kw["relative_grad_norm_threshold"])
    o.obj_step_threshold = float(kw["obj_step_threshold"])
    o.inputs_step_threshold = float(kw["inputs_step_threshold"])
    o.make_psd = int(bool(kw["make_psd"]))
    o.psd_delta = float(kw["psd_delta"])
    o.alpha_0 = float(kw["alpha_0"])
    o.alpha_min = float(kw["alpha_min"])
    return o


def pack_layout(shape, which):
    """[(flax tree path, offset, rows, cols, ld), ...] of a flat parameter vector (gmpc_pack_layout):
    which = 0 dyn, 1 cost, 2 critic, 3 the training vector [mpc_weights | cost | dynamics | critic]."""
    lib = _lib.load()
    count = lib.gmpc_pack_layout(C.byref(shape), int(which), None, 0)
    if count < 0:
        _lib.check(count)
    leaves = (_lib.Leaf * count)()
    _lib.check(min(0, lib.gmpc_pack_layout(C.byref(shape), int(which), leaves, count)))
    return [(lf.name.decode(), lf.offset, lf.rows, lf.cols, lf.ld) for lf in leaves]


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype in (torch.float32, torch.int32) and t.is_contiguous(), \
        "C-ABI buffers must be contiguous fp32/int32 device tensors"
    return C.c_void_p(t.data_ptr())


class Engine:
    """One context on one GPU.  All tensor arguments are contiguous fp32 CUDA(HIP) tensors."""

    def __init__(self, n, m, T, dyn_dims, cost_dims, max_batch, lstm_features=0, head_dims=None,
                 device=None, dyn_lstm=0, x_size=0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GmpcError("no HIP device visible to torch; gan_mpc_amd has no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        self.shape = make_shape(n, m, T, dyn_dims, cost_dims, lstm_features, head_dims, dyn_lstm, x_size)
        self.n, self.m, self.T = int(n), int(m), int(T)
        self.nx = int(x_size) if dyn_lstm else int(n)      # x part of xc: goals / critic / expert columns
        self.max_batch = int(max_batch)
        # the step-major ("large-state") pipeline: n > 64, or more than 32 controls; it keeps ONE step of Jacobians
        self.big = self.n > 64 or self.m > 32
        self.dyn_count = self.lib.gmpc_param_count(C.byref(self.shape), 0)
        self.cost_count = self.lib.gmpc_param_count(C.byref(self.shape), 1)
        self.critic_count = self.lib.gmpc_param_count(C.byref(self.shape), 2) if lstm_features else 0
        ctx = C.c_void_p()
        _lib.check(self.lib.gmpc_create(C.byref(self.shape), self.max_batch, self.device.index,
                                        C.byref(ctx)))
        self.ctx = ctx
        self._bound = None

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.gmpc_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.device)

    def to_dev(self, a, dtype=torch.float32):
        if torch.is_tensor(a):
            return a.to(device=self.device, dtype=dtype).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(self.device).contiguous()

    def set_params(self, mpc_w, dyn, cost):
        assert mpc_w.numel() == 3 and dyn.numel() == self.dyn_count and cost.numel() == self.cost_count
        self._bound = (mpc_w, dyn, cost)  # keep the tensors alive: the ctx holds raw pointers
        _lib.check(self.lib.gmpc_set_params(self.ctx, _ptr(mpc_w), _ptr(dyn), _ptr(cost),
                                            self._stream()))

    def rollout_cost(self, x0, U, goal, X=None, costs=None):
        B = x0.shape[0]
        X = self.new(B, self.T + 1, self.n) if X is None else X
        costs = self.new(B, self.T + 1) if costs is None else costs
        _lib.check(self.lib.gmpc_rollout_cost(self.ctx, B, _ptr(x0), _ptr(U), _ptr(goal), _ptr(X),
                                              _ptr(costs), self._stream()))
        return X, costs

    def get_cost(self, x, u, goal_row, terminal):
        """cost_model.get_cost for B independent (x, u) pairs: the staging branch against goal_row
        (B, n), or (terminal) the terminal branch of an arbitrary state -> (B,)."""
        B = x.shape[0]
        cost = self.new(B)
        _lib.check(self.lib.gmpc_get_cost(self.ctx, B, _ptr(x), _ptr(u), _ptr(goal_row),
                                          int(bool(terminal)), _ptr(cost), self._stream()))
        return cost

    def predict(self, x, u):
        """dynamics_model.predict for B independent (x, u) pairs -> next_x (B, n)."""
        B = x.shape[0]
        nxt = self.new(B, self.n)
        _lib.check(self.lib.gmpc_predict(self.ctx, B, _ptr(x), _ptr(u), _ptr(nxt), self._stream()))
        return nxt

    def lqr_backward(self, X, U, goal, after_rollout=False, out=None):
        B = X.shape[0]
        n, m, T = self.n, self.m, self.T
        if out is None:
            out = dict(K=self.new(B, T, m, n), k=self.new(B, T, m), grad=self.new(B, T, m),
                       adjoints=self.new(B, T + 1, n))
            if not self.big:   # the large-state pass is step-major and never holds all T Jacobians
                out["AB"] = self.new(B, T, n, n + m)
        fn = self.lib.gmpc_lqr_backward_after_rollout if after_rollout else self.lib.gmpc_lqr_backward
        _lib.check(fn(self.ctx, B, _ptr(X), _ptr(U), _ptr(goal), _ptr(out["K"]), _ptr(out["k"]),
                      _ptr(out["grad"]), _ptr(out["adjoints"]), _ptr(out.get("AB")), self._stream()))
        return out

    def ilqr_solve(self, x0, U, goal, kwargs=None):
        B = x0.shape[0]
        n, m, T = self.n, self.m, self.T
        opts = make_opts(kwargs)
        out = dict(X=self.new(B, T + 1, n), U=self.new(B, T, m), obj=self.new(B),
                   grad=self.new(B, T, m), adjoints=self.new(B, T + 1, n),
                   iterations=self.new(B, dtype=torch.int32))
        _lib.check(self.lib.gmpc_ilqr_solve(
            self.ctx, B, _ptr(x0), _ptr(U), _ptr(goal), C.byref(opts), _ptr(out["X"]), _ptr(out["U"]),
            _ptr(out["obj"]), _ptr(out["grad"]), _ptr(out["adjoints"]), _ptr(out["iterations"]),
            self._stream()))
        return out

    def bilevel_grad(self, B, loss_kind, desired=None, critic=None, sign=1.0, grad_sum=None):
        """grad_sum: optional caller-owned [3 + cost_count] view (e.g. of a packed all-reduce buffer)."""
        loss = self.new(B)
        grad_sum = self.new(3 + self.cost_count) if grad_sum is None else grad_sum
        assert grad_sum.numel() == 3 + self.cost_count
        _lib.check(self.lib.gmpc_bilevel_grad(self.ctx, B, int(loss_kind), _ptr(desired), _ptr(critic),
                                              float(sign), _ptr(loss), _ptr(grad_sum), self._stream()))
        return loss, grad_sum

    def upper_loss(self, B, loss_kind, desired=None, critic=None):
        loss = self.new(B)
        _lib.check(self.lib.gmpc_upper_loss(self.ctx, B, int(loss_kind), _ptr(desired), _ptr(critic),
                                            _ptr(loss), self._stream()))
        return loss

    def expert_rollout(self, history, expert_flat, expert_shape):
        """history (B, hist+1, n) -> goal (B, T+1, n), init_U (B, T, m) from the expert sequence model."""
        B, hist = history.shape[0], history.shape[1] - 1
        want = self.lib.gmpc_expert_param_count(self.nx, C.byref(expert_shape))
        assert expert_flat.numel() == want, (expert_flat.numel(), want)
        goal = self.new(B, self.T + 1, self.nx)
        init_U = self.new(B, self.T, self.m)
        _lib.check(self.lib.gmpc_expert_rollout(self.ctx, B, hist, C.byref(expert_shape), _ptr(expert_flat),
                                                _ptr(history), _ptr(goal), _ptr(init_U), self._stream()))
        return goal, init_U

    def dynamics_loss_grad(self, xseq, useq, next_xseq, discount, teacher_forcing, loss_sum=None,
                           grad_sum=None):
        """-> (loss_sum[1], grad_sum[dyn_count]) of the multi-step prediction loss over the batch."""
        B, S = xseq.shape[0], xseq.shape[1]
        loss_sum = self.new(1) if loss_sum is None else loss_sum
        grad_sum = self.new(self.dyn_count) if grad_sum is None else grad_sum
        assert loss_sum.numel() == 1 and grad_sum.numel() == self.dyn_count
        _lib.check(self.lib.gmpc_dynamics_loss_grad(
            self.ctx, B, S, _ptr(xseq), _ptr(useq), _ptr(next_xseq), float(discount),
            int(bool(teacher_forcing)), _ptr(loss_sum), _ptr(grad_sum), self._stream()))
        return loss_sum, grad_sum

    def polyak(self, prev, cur, factor, out=None):
        out = cur if out is None else out
        _lib.check(self.lib.gmpc_polyak(self.ctx, prev.numel(), _ptr(prev), _ptr(cur), float(factor),
                                        _ptr(out), self._stream()))
        return out

    def critic_loss_grad(self, xseq, label, critic, loss_sum=None, grad_sum=None):
        """loss_sum [1], grad_sum [critic_count]: optional caller-owned views (a packed all-reduce buffer)."""
        Bc = xseq.shape[0]
        loss_sum = self.new(1) if loss_sum is None else loss_sum
        grad_sum = self.new(self.critic_count) if grad_sum is None else grad_sum
        assert loss_sum.numel() == 1 and grad_sum.numel() == self.critic_count
        _lib.check(self.lib.gmpc_critic_loss_grad(self.ctx, Bc, _ptr(xseq), _ptr(label), _ptr(critic),
                                                  _ptr(loss_sum), _ptr(grad_sum), self._stream()))
        return loss_sum, grad_sum

    def critic_score_vjp(self, xseq, critic, want_dx=True):
        Bc = xseq.shape[0]
        score = self.new(Bc)
        dx = self.new(Bc, self.T + 1, self.nx) if want_dx else None
        _lib.check(self.lib.gmpc_critic_score_vjp(self.ctx, Bc, _ptr(xseq), _ptr(critic), _ptr(score),
                                                  _ptr(dx), self._stream()))
        return score, dx

    def adam_clip_step(self, params, grad, m, v, step, lr, grad_scale=1.0, max_norm=100.0, b1=0.9,
                       b2=0.999, eps=1e-8):
        _lib.check(self.lib.gmpc_adam_clip_step(
            self.ctx, params.numel(), _ptr(params), _ptr(grad), _ptr(m), _ptr(v), float(grad_scale),
            int(step), float(lr), float(max_norm), float(b1), float(b2), float(eps), self._stream()))

    def set_linearize_event(self, event=None):
        """Record `event` (a torch.cuda.Event that has been recorded once, or None) after the Jacobian chain of
        every backward pass: gmpc_set_linearize_event.  The caller keeps the event alive."""
        import ctypes as C
        h = None if event is None else C.c_void_p(event.cuda_event)
        _lib.check(self.lib.gmpc_set_linearize_event(self.ctx, h))
        self._lin_event = event

    def linesearch_candidates(self):
        """Candidate rollouts evaluated by the line searches of the last ilqr_solve."""
        return int(self.lib.gmpc_linesearch_candidates(self.ctx))

    def linesearch_stats(self):
        """Counters of the last ilqr_solve's line searches: how many accepted alpha_0 / 2^k (list `accepted`,
        k = 0..15), how many ran out of step sizes, candidate rollouts per speculative round."""
        out = (C.c_long * 64)()
        _lib.check(self.lib.gmpc_linesearch_stats(self.ctx, out, 64))
        v = list(out)
        rounds = v[24:64]
        while rounds and rounds[-1] == 0:
            rounds.pop()
        return {"accepted": v[:16], "exhausted": v[16], "deepest_accepted": v[17], "round_items": rounds}

    PROF_SLOTS = ("rollout", "linearize", "terminal", "riccati", "linesearch", "lstm_fwd", "head",
                  "lstm_bwd", "wgrad", "adam")

    def profile_enable(self, on=True):
        _lib.check(self.lib.gmpc_profile_enable(self.ctx, int(bool(on))))

    def profile_read(self):
        """{kernel: (total_ms, launches)} since the last read (HIP events on the launch stream)."""
        out = {}
        for i, name in enumerate(self.PROF_SLOTS):
            ms, cnt = C.c_double(), C.c_int()
            _lib.check(self.lib.gmpc_profile_read(self.ctx, i, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    def profile_kernel_name(self, slot_name):
        """Name of the kernel the slot's last launch ran on ('' when the slot has one kernel only)."""
        name = self.lib.gmpc_profile_kernel_name(self.ctx, self.PROF_SLOTS.index(slot_name))
        return name.decode() if name else ""

    def debug_buffer(self, which, shape):
        """Copy of one of the ctx's internal solution buffers (see gmpc_debug_buffer)."""
        p = self.lib.gmpc_debug_buffer(self.ctx, which)
        if not p:
            raise _lib.GmpcError(f"no debug buffer {which}")
        n = int(np.prod(shape))
        have = self.lib.gmpc_debug_buffer_count(self.ctx, which)
        if n > have:
            raise _lib.GmpcError(f"debug buffer {which} holds {have} floats, {n} requested {tuple(shape)}")
        torch.cuda.synchronize(self.device)
        view = torch.as_tensor(_RawDevBuffer(p, n), device=self.device)
        return view.clone().reshape(shape)


class _RawDevBuffer:
    """__cuda_array_interface__ view of a raw fp32 device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {
            "shape": (count,), "typestr": "<f4", "data": (int(ptr), False), "version": 2}
