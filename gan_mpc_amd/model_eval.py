"""Stand-alone evaluation of the model protocol (reference base.py:4-49) with the reference's own
signatures: cost_model.get_cost(xc, u, t, cost_params, mpc_weights, goal_X), dynamics_model.predict(xc, u,
t, dynamics_params), critic_model.predict(xseq, critic_params).  Inside a solve these evaluations are
fused into the trajectory kernels; called on their own (the way policy/eval.py:64-73 and
gan/js_policy.py:43 call them) each model builds a small engine for its own network, lazily, with
zero-weight placeholders for the networks the call does not involve, and runs the matching C-ABI entry
point (gmpc_get_cost / gmpc_predict / gmpc_critic_score_vjp).  No arithmetic happens on the host."""

import numpy as np
import torch

from gan_mpc_amd import params as P
from gan_mpc_amd.engine import Engine

_ENGINES = {}


def _engine(n, m, T, dyn_dims, cost_dims, F=0, head=None, batch=1, dyn_lstm=0, x_size=0):
    key = (int(n), int(m), int(T), tuple(dyn_dims), tuple(cost_dims), int(F), tuple(head or ()),
           int(dyn_lstm), int(x_size), torch.cuda.current_device() if torch.cuda.is_available() else -1)
    eng = _ENGINES.get(key)
    if eng is None or eng.max_batch < batch:
        if eng is not None:
            eng.close()
        eng = Engine(n, m, T, list(dyn_dims), list(cost_dims), max_batch=max(8, batch), lstm_features=F,
                     head_dims=list(head) if head else None, dyn_lstm=dyn_lstm, x_size=x_size)
        _ENGINES[key] = eng
    return eng


def _count(dims):
    return sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))


def _rows(a, width):
    a = np.asarray(a, np.float32)
    single = a.ndim == 1
    a = a.reshape(-1, width)
    return a, single


def get_cost(horizon, xc, u, t, cost_params, mpc_weights, goal_X):
    """reference cost/cost_model.py:33-42: where(t == horizon, terminal, staging) for one (xc, u) or a
    batch of them (leading axis; goal_X is then (B, T+1, n))."""
    cost_dims = P.mlp_dims(cost_params)
    n = cost_dims[0]
    x, single = _rows(xc, n)
    uu = np.asarray(u, np.float32).reshape(x.shape[0], -1)
    m = uu.shape[1]
    # (the terminal branch reads no goal: goal_X may be None there, like the NULL goal_row of gmpc_get_cost)
    ng = n if goal_X is None else np.shape(goal_X)[-1]
    if ng == n:
        eng = _engine(n, m, 1, [n + m, 1, n], cost_dims, batch=x.shape[0])
        dyn_count = _count([n + m, 1, n])
    else:
        # xc carries an LSTM dynamics' (c, h) behind x: the staging cost sees xc[:ng] (cost_model.py:24-25)
        Fd = (n - ng) // 2
        eng = _engine(n, m, 1, [Fd, ng], cost_dims, batch=x.shape[0], dyn_lstm=Fd, x_size=ng)
        dyn_count = (ng + m + Fd) * 4 * Fd + 4 * Fd + _count([Fd, ng])
    d = eng.to_dev
    eng.set_params(d(np.asarray(mpc_weights, np.float32).reshape(3)),
                   d(np.zeros(dyn_count, np.float32)), d(P.pack_mlp(cost_params)))
    terminal = int(t) == int(horizon)
    goal_row = None
    if not terminal and goal_X is None:
        raise ValueError("get_cost: the staging branch (t < horizon) needs goal_X")
    if not terminal:
        g = np.asarray(goal_X, np.float32)
        goal_row = d(g[int(t)][None] if g.ndim == 2 else g[:, int(t)])
    cost = eng.get_cost(d(x), d(uu), goal_row, terminal)
    return cost[0] if single else cost


def predict(xc, u, dynamics_params):
    """reference dynamics/dynamics_model.py:45-48 (the MLP's carry is empty, dynamics/nn.py:15-17)."""
    dyn_dims, Fd = P.dynamics_meta(dynamics_params)
    nx = dyn_dims[-1]
    if Fd:
        p = dynamics_params["params"]
        m = int(np.asarray(p[P._lstm_scope(p)]["ii"]["kernel"]).shape[0]) - nx
        n = nx + 2 * Fd
    else:
        n, m = nx, dyn_dims[0] - nx
    x, single = _rows(xc, n)
    uu, _ = _rows(u, m)
    eng = _engine(n, m, 1, dyn_dims, [n, 1], batch=x.shape[0], dyn_lstm=Fd, x_size=nx if Fd else 0)
    d = eng.to_dev
    eng.set_params(d(np.zeros(3, np.float32)), d(P.pack_dynamics(dynamics_params)),
                   d(np.zeros(_count([n, 1]), np.float32)))
    nxt = eng.predict(d(x), d(uu))
    return nxt[0] if single else nxt


def critic_predict(xseq, critic_params):
    """reference critic/critic_model.py:15-16 -> critic/nn.py:28-42: one sequence (T+1, n) -> (1,)
    score, a batch (B, T+1, n) -> (B,)."""
    n, F, head = P.critic_dims(critic_params)
    xs = xseq if torch.is_tensor(xseq) else np.asarray(xseq, np.float32)
    single = xs.ndim == 2
    if single:
        xs = xs[None]
    T = xs.shape[1] - 1
    eng = _engine(n, 1, T, [n + 1, 1, n], [n, 1], F=F, head=head, batch=(xs.shape[0] + 1) // 2)
    d = eng.to_dev
    score, _ = eng.critic_score_vjp(d(xs), d(P.pack_critic(critic_params)), want_dx=False)
    return score[:1] if single else score
