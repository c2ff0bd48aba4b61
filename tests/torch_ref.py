"""Independent float64 autograd transcription of the reference formulas.

TEST INFRASTRUCTURE.  A literal, single-trajectory restatement of the
reference's model definitions in torch (CPU, float64), differentiated by
torch autograd instead of by the hand-derived formulas in ``oracle/``:

* dynamics  -- reference dynamics/nn.py:27-34
* cost      -- reference cost/cost_model.py:20-42, cost/nn.py:23-29
* objective -- reference policy/optimizers.py:24-31
* bilevel   -- reference policy/optimizers.py:61-71,78-105 (dense Hessian,
               dense solve, mixed-derivative VJP), exactly as written there.

It shares no code with the oracle, which is the point.
"""

import torch

ALPHA = 1e-2


def t64(a):
    return torch.as_tensor(a, dtype=torch.float64)


def layers64(layers):
    return [(t64(W), t64(b)) for W, b in layers]


def mlp(layers, q):
    for W, b in layers[:-1]:
        q = torch.relu(q @ W + b)
    W, b = layers[-1]
    return q @ W + b


def dynamics(dyn, x, u):
    return mlp(dyn, torch.cat([x, u])) + x


def cost(cmlp, mpc_w, goal, x, u, t, T):
    w = torch.sigmoid(mpc_w)
    if t == T:
        y = mlp(cmlp, x)
        return w[2] * torch.dot(y, y)
    u_cost = torch.sqrt(torch.dot(u, u) + ALPHA**2) - ALPHA
    d = x - goal[t]
    x_cost = torch.sqrt(torch.dot(d, d) + ALPHA**2) - ALPHA
    return w[0] * u_cost + w[1] * x_cost


def rollout(dyn, U, x0):
    xs = [x0]
    for t in range(U.shape[0]):
        xs.append(dynamics(dyn, xs[-1], U[t]))
    return torch.stack(xs)


def objective(dyn, cmlp, mpc_w, goal, U, x0):
    T = U.shape[0]
    X = rollout(dyn, U, x0)
    zero_u = torch.zeros_like(U[0])
    tot = 0.0
    for t in range(T + 1):
        tot = tot + cost(cmlp, mpc_w, goal, X[t], U[t] if t < T else zero_u, t, T)
    return tot


def l2_loss(X, desired):
    return torch.sum(torch.mean((X - desired) ** 2, dim=0))


def lstm_critic(cr, xseq):
    Wx, Wh, b = cr["Wx"], cr["Wh"], cr["b"]
    F = Wh.shape[0]
    c = torch.zeros(F, dtype=xseq.dtype)
    h = torch.zeros(F, dtype=xseq.dtype)
    for t in range(xseq.shape[0]):
        z = xseq[t] @ Wx + h @ Wh + b
        i = torch.sigmoid(z[:F])
        f = torch.sigmoid(z[F:2 * F])
        g = torch.tanh(z[2 * F:3 * F])
        o = torch.sigmoid(z[3 * F:])
        c = f * c + i * g
        h = o * torch.tanh(c)
    return mlp(cr["head"], h)


def critic64(cr):
    return dict(Wx=t64(cr["Wx"]), Wh=t64(cr["Wh"]), b=t64(cr["b"]),
                head=layers64(cr["head"]))


def generator_loss(cr, X):
    p = torch.sigmoid(lstm_critic(cr, X))
    return torch.mean(-torch.log(p) + torch.log(1 - p))


def bilevel_dense(dyn, cmlp, mpc_w, goal, x0, U, loss_fn):
    """policy/optimizers.py:61-71 as written: B, dense A, solve, cost_vjp.

    Returns B (T*m), A (T*m,T*m), H, grads wrt (mpc_w, cost layers)."""
    T, m = U.shape
    U = U.clone().requires_grad_(True)

    def J(Uf, cm, mw):
        return objective(dyn, cm, mw, goal, Uf.reshape(T, m), x0)

    Bv = torch.autograd.grad(loss_fn(rollout(dyn, U, x0)), U)[0].reshape(-1)
    A = torch.autograd.functional.hessian(
        lambda Uf: J(Uf, cmlp, mpc_w), U.detach().reshape(-1)
    )
    H = torch.linalg.solve(A, Bv)
    leaves = [mpc_w.clone().requires_grad_(True)]
    cm = []
    for W, b in cmlp:
        W = W.clone().requires_grad_(True)
        b = b.clone().requires_grad_(True)
        cm.append((W, b))
        leaves += [W, b]
    Uf = U.detach().reshape(-1).clone().requires_grad_(True)
    gU = torch.autograd.grad(J(Uf, cm, leaves[0]), Uf, create_graph=True)[0]
    outer = torch.dot(H.detach(), gU)
    grads = torch.autograd.grad(outer, leaves, allow_unused=True)
    return Bv, A, H, grads
