"""CPU oracle for the GAN-MPC inner loop -- TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the arithmetic on the hot path of
returaj/gan_mpc.  It exists so that the
HIP kernels under ``gan_mpc_amd/csrc`` can be checked against an independent
CPU computation of the same quantities.  It is NOT part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it.  The product path never falls back to it.

PARITY UNPINNED.  The reference ships no tests, golden vectors or fixtures,
and its hot path cannot be imported here (jax / flax / optax / trajax are
absent).  The iLQR arithmetic lives in the
un-vendored third-party module ``trajax`` pinned in the reference's
``requirements.txt:51`` to commit c94a637c5a397b3d4100153f25b4b165507b5b20;
the functions marked [trajax] below restate that module's published algorithm
(``trajax/optimizers.py``: pad, rollout, evaluate, linearize, quadratize,
lqr_step, tvlqr, ddp_rollout, line_search_ddp, adjoint, ilqr_base) as recalled,
anchored on the reference's own call sites (``policy/optimizers.py:19,26-29,
55,80``) and on the keyword set in ``policy/eval.py:10-20``.  What pins this
oracle instead: analytic known-answer tests, finite differences, and an
independent torch-CPU float64 autograd transcription of the reference's
formulas (tests/torch_ref.py).

All functions are batched over a leading axis B (the reference's ``jax.vmap``
axis, ``policy/base.py:122-125``) and are dtype-generic: pass float64 arrays for
the arbiter, float32 arrays for the like-for-like comparison.

Conventions
-----------
* A dense layer is a pair ``(W, b)`` with ``W`` of shape (in, out) -- the flax
  ``Dense`` kernel layout -- and ``y = x @ W + b``.
* ``dyn``  : list of L (W, b); hidden layers use relu; residual output
  (reference ``dynamics/nn.py:27-34``).  The MLP dynamics has an empty carry
  (``dynamics/nn.py:15-17``), so ``xc == x``.
* ``cmlp`` : list of (W, b); relu hidden; output y; cost feature ``dot(y, y)``
  (reference ``cost/nn.py:23-29``).
* ``mpc_w``: raw 3-vector (action, state, terminal); ``sigmoid`` is applied
  inside the cost (reference ``cost/cost_model.py:37``).
* ``goal`` : (B, T+1, n); row t is used at stage t (``cost_model.py:36``).
"""

import numpy as np

ALPHA = 1e-2  # smoothing constant, reference cost/cost_model.py:22

ILQR_KWARGS = {  # reference policy/eval.py:10-20
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


# --------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x)
    return (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)


def _dt(*arrs):
    return np.result_type(*[np.asarray(a).dtype for a in arrs])


def mlp_forward(layers, q):
    """relu MLP.  Returns output and the list of hidden pre-activations."""
    zs = []
    for W, b in layers[:-1]:
        z = q @ W + b
        zs.append(z)
        q = np.maximum(z, 0)
    W, b = layers[-1]
    return q @ W + b, zs


def dynamics_predict(dyn, x, u):
    """a1/a2: next_x = MLP([x, u]) + x   (dynamics/nn.py:27-34).  A dict selects the LSTM variant
    (dynamics/nn.py:37-57, see lstm_dynamics_predict); x is then xc = [x, c, h]."""
    if isinstance(dyn, dict):
        return lstm_dynamics_predict(dyn, x, u)
    q = np.concatenate([x, u], axis=-1)
    out, zs = mlp_forward(dyn, q)
    return out + x, zs


def lstm_dynamics_predict(dl, xc, u, keep=False):
    """dynamics/nn.py:37-57: xc = [x (nx), c (F), h (F)];  q0 = [x, u];  (c', h') = OptimizedLSTMCell((c, h), q0)
    (gates i, f, g, o as in the critic: i, f, o sigmoid, g tanh, c' = f c + i g, h' = o tanh(c'));  the
    relu MLP tail maps h' -> nx;  next_xc = [tail(h') + x, c', h'].
    dl = dict(Wx ((nx+m), 4F), Wh (F, 4F), b (4F), tail [(W, b), ...]).  Returns (next_xc, tail pre-activations)
    or, with keep, also the cell's intermediates for the Jacobian."""
    F = dl["Wh"].shape[0]
    nx = xc.shape[-1] - 2 * F
    x, c, h = xc[..., :nx], xc[..., nx:nx + F], xc[..., nx + F:]
    z = np.concatenate([x, u], -1) @ dl["Wx"] + h @ dl["Wh"] + dl["b"]
    i, f, o = sigmoid(z[..., :F]), sigmoid(z[..., F:2 * F]), sigmoid(z[..., 3 * F:])
    g = np.tanh(z[..., 2 * F:3 * F])
    c2 = f * c + i * g
    tc = np.tanh(c2)
    h2 = o * tc
    out, zs = mlp_forward(dl["tail"], h2)
    nxt = np.concatenate([out + x, c2, h2], -1)
    if keep:
        return nxt, zs, dict(i=i, f=f, g=g, o=o, c=c, tc=tc)
    return nxt, zs


def mlp_input_jacobian(layers, zs):
    """d out / d in for a relu MLP at the point whose pre-activations are zs.

    Reverse chain  J = W_L^T D_{L-1} W_{L-1}^T ... D_1 W_1^T  with D = [z > 0]
    (jax.nn.relu has derivative 0 at z == 0).  Returns (B, out, in).
    """
    W_last = layers[-1][0]
    B = zs[0].shape[0] if zs else 1
    G = np.broadcast_to(W_last.T, (B,) + W_last.T.shape).copy()  # (B,out,h)
    for (W, _), z in zip(reversed(layers[:-1]), reversed(zs)):
        G = G * (z > 0)[:, None, :].astype(G.dtype)
        G = G @ W.T
    return G


def lstm_dynamics_jacobians(dl, xc, u):
    """A = d next_xc / d xc (B,N,N), Bm = d next_xc / d u (B,N,m), N = nx + 2F, by the chain rule through
    the cell: dc'/dz = [g i(1-i), c f(1-f), i (1-g^2), 0], dc'/dc = f, dh'/dc' = o (1 - tanh^2 c'),
    dh'/dz_o = tanh(c') o (1-o), dz/d[x,u] = Wx^T, dz/dh = Wh^T; next_x = tail(h') + x."""
    F = dl["Wh"].shape[0]
    N = xc.shape[-1]
    nx = N - 2 * F
    m = u.shape[-1]
    _, zs, k = lstm_dynamics_predict(dl, xc, u, keep=True)
    i, f, g, o, c, tc = k["i"], k["f"], k["g"], k["o"], k["c"], k["tc"]
    Bsz = xc.shape[0]
    dt = xc.dtype
    # d z / d [x, c, h, u]  (B, 4F, N + m): columns x | c | h | u
    WxT, WhT = dl["Wx"].T, dl["Wh"].T          # (4F, nx+m), (4F, F)
    dz = np.zeros((4 * F, N + m), dt)
    dz[:, :nx] = WxT[:, :nx]
    dz[:, nx + F:N] = WhT
    dz[:, N:] = WxT[:, nx:]
    dzi, dzf, dzg, dzo = dz[:F], dz[F:2 * F], dz[2 * F:3 * F], dz[3 * F:]
    dc2 = ((g * i * (1 - i))[..., None] * dzi + (c * f * (1 - f))[..., None] * dzf
           + (i * (1 - g * g))[..., None] * dzg)                       # (B, F, N+m)
    eyeF = np.eye(F, dtype=dt)
    dc2[:, :, nx:nx + F] += f[..., None] * eyeF
    dh2 = (o * (1 - tc * tc))[..., None] * dc2 + (tc * o * (1 - o))[..., None] * dzo
    Jt = mlp_input_jacobian(dl["tail"], zs)                            # (B, nx, F)
    J = np.zeros((Bsz, N, N + m), dt)
    J[:, :nx] = Jt @ dh2
    J[:, :nx, :nx] += np.eye(nx, dtype=dt)
    J[:, nx:nx + F] = dc2
    J[:, nx + F:] = dh2
    return J[:, :, :N], J[:, :, N:]


def lstm_dynamics_curvature(dl, xc, u, lam_next):
    """Phi = d^2/dz^2 [lam_next . f(z)], z = (xc, u), for the LSTM dynamics: (B, N+m, N+m), columns x | c | h | u.

    The relu tail is piecewise linear, so all curvature sits in the cell.  With w_h = Jt^T lam_x + lam_h (the
    adjoint reaching h'), w_c = lam_c, unit j contributes phi_j(z_i, z_f, z_g, z_o, c) = w_h o tanh(c') + w_c c',
    c' = f c + i g, a function of its four gate pre-activations (affine in (x, u, h)) and of c_j:
    Phi = sum_j D_j^T H_j D_j with D_j the 5 x (N+m) Jacobian of (z_i, z_f, z_g, z_o, c)_j and H_j the 5 x 5
    Hessian of phi_j."""
    F = dl["Wh"].shape[0]
    N = xc.shape[-1]
    nx = N - 2 * F
    m = u.shape[-1]
    Bsz = xc.shape[0]
    dt = xc.dtype
    _, zs, k = lstm_dynamics_predict(dl, xc, u, keep=True)
    i, f, g, o, c, tc = k["i"], k["f"], k["g"], k["o"], k["c"], k["tc"]
    Jt = mlp_input_jacobian(dl["tail"], zs)                          # (B, nx, F)
    wh = np.einsum("bxf,bx->bf", Jt, lam_next[:, :nx]) + lam_next[:, nx + F:]
    wc = lam_next[:, nx:nx + F]
    di, df, dg, do = i * (1 - i), f * (1 - f), 1 - g * g, o * (1 - o)
    ddi, ddf, ddg, ddo = di * (1 - 2 * i), df * (1 - 2 * f), -2 * g * dg, do * (1 - 2 * o)
    alpha = wh * o * (1 - tc * tc) + wc                              # d phi / d c'
    beta = wh * o * (-2 * tc) * (1 - tc * tc)                        # d^2 phi / d c'^2
    gamma = wh * (1 - tc * tc)                                       # d^2 phi / d o d c'
    # gradient and Hessian of c' wrt v = (z_i, z_f, z_g, z_o, c)
    gc = np.stack([g * di, c * df, i * dg, np.zeros_like(c), f], -1)            # (B, F, 5)
    H = beta[..., None, None] * gc[..., :, None] * gc[..., None, :]
    H[..., 0, 0] += alpha * g * ddi
    H[..., 0, 2] += alpha * di * dg
    H[..., 2, 0] += alpha * di * dg
    H[..., 1, 1] += alpha * c * ddf
    H[..., 1, 4] += alpha * df
    H[..., 4, 1] += alpha * df
    H[..., 2, 2] += alpha * i * ddg
    H[..., 3, 3] += wh * tc * ddo
    for q in (0, 1, 2, 4):
        H[..., 3, q] += gamma * do * gc[..., q]
        H[..., q, 3] += gamma * do * gc[..., q]
    # D_j: rows z_i, z_f, z_g, z_o (columns of Wx / Wh for unit j) and c_j (unit vector)
    WxT, WhT = dl["Wx"].T, dl["Wh"].T
    dz = np.zeros((4 * F, N + m), dt)
    dz[:, :nx] = WxT[:, :nx]
    dz[:, nx + F:N] = WhT
    dz[:, N:] = WxT[:, nx:]
    D = np.zeros((F, 5, N + m), dt)
    for q in range(4):
        D[:, q] = dz[q * F:(q + 1) * F]
    D[np.arange(F), 4, nx + np.arange(F)] = 1.0
    return np.einsum("jqa,bjqr,jrc->bac", D, H, D)


def second_order_lqr(dyn, lqr, adjoints, X, U):
    """The LQ model whose Hessian in U is the EXACT d^2 J / dU^2 of the rollout objective (what
    policy/optimizers.py:86-90 differentiates): Q~ = Q + Phi_xx, R~ = R + Phi_uu, M~ = M + Phi_xu with
    Phi_t = d^2/dz^2 [lambda_{t+1} . f(z_t)], lambda the adjoints of J at this U (the Hessian of the Lagrangian
    restricted to the linearised dynamics).  The relu MLP has Phi = 0 almost everywhere: lqr is returned as is."""
    if not isinstance(dyn, dict):
        return lqr
    Q, q, R, r, M, A, Bm = lqr
    Bsz, T, m = U.shape
    N = X.shape[-1]
    Q, R, M = Q.copy(), R.copy(), M.copy()
    for t in range(T):
        Phi = lstm_dynamics_curvature(dyn, X[:, t], U[:, t], adjoints[:, t + 1])
        Q[:, t] += Phi[:, :N, :N]
        R[:, t] += Phi[:, N:, N:]
        M[:, t] += Phi[:, :N, N:]
    return Q, q, R, r, M, A, Bm


def dynamics_jacobians(dyn, x, u):
    """[trajax linearize] A = d f/d x (B,n,n),  Bm = d f/d u (B,n,m)."""
    if isinstance(dyn, dict):
        return lstm_dynamics_jacobians(dyn, x, u)
    n = x.shape[-1]
    _, zs = dynamics_predict(dyn, x, u)
    J = mlp_input_jacobian(dyn, zs)
    A = J[:, :, :n] + np.eye(n, dtype=J.dtype)
    return A, J[:, :, n:]


def stage_cost(x, u, goal_t, w):
    """cost_model.py:20-28 with w = sigmoid(mpc_w)[:2]."""
    a = np.asarray(ALPHA, dtype=x.dtype)
    u_cost = np.sqrt(np.sum(u * u, -1) + a * a) - a
    d = x[..., : goal_t.shape[-1]] - goal_t
    x_cost = np.sqrt(np.sum(d * d, -1) + a * a) - a
    return w[0] * u_cost + w[1] * x_cost


def terminal_cost(cmlp, x, w2):
    """cost_model.py:30-31 and cost/nn.py:23-29."""
    y, _ = mlp_forward(cmlp, x)
    return w2 * np.sum(y * y, -1)


def pad(U):
    """[trajax pad] append one zero control row."""
    return np.concatenate([U, np.zeros_like(U[:, :1])], axis=1)


def rollout(dyn, U, x0):
    """[trajax rollout] X[:,0]=x0, X[:,t+1]=f(X[:,t],U[:,t]).  (B,T+1,n)."""
    B, T, _ = U.shape
    X = np.empty((B, T + 1, x0.shape[-1]), dtype=x0.dtype)
    X[:, 0] = x0
    for t in range(T):
        X[:, t + 1], _ = dynamics_predict(dyn, X[:, t], U[:, t])
    return X


def evaluate(cmlp, mpc_w, goal, X, U):
    """[trajax evaluate] per-step costs (B,T+1) of get_cost (cost_model.py:33-42).

    U has T rows; the padded zero control of the last step never enters because
    ``where(t == H, terminal, stage)`` selects the terminal branch there.
    """
    T = U.shape[1]
    w = sigmoid(np.asarray(mpc_w, dtype=X.dtype))
    c = np.empty(X.shape[:2], dtype=X.dtype)
    c[:, :T] = stage_cost(X[:, :T], U, goal[:, :T], w)
    c[:, T] = terminal_cost(cmlp, X[:, T], w[2])
    return c


def objective(dyn, cmlp, mpc_w, goal, U, x0):
    """a7: policy/optimizers.py:24-31."""
    X = rollout(dyn, U, x0)
    return np.sum(evaluate(cmlp, mpc_w, goal, X, U), axis=1)


# --------------------------------------------------------------------------
# linearise / quadratise  [trajax linearize, quadratize]
# --------------------------------------------------------------------------
def cost_quadratize(cmlp, mpc_w, goal, X, U):
    """Q (B,T+1,n,n), q (B,T+1,n), R (B,T+1,m,m), r (B,T+1,m), M (B,T+1,n,m).

    Stage rows are closed form; the terminal row uses the relu-MLP Jacobian Jc:
    grad = 2 w2 Jc^T y, Hessian = 2 w2 Jc^T Jc exactly (relu'' = 0).  Row T of
    R, r and every row of M are zero (the terminal branch has no u).
    """
    B, T, m = U.shape
    n = X.shape[-1]
    dt = X.dtype
    w = sigmoid(np.asarray(mpc_w, dtype=dt))
    a = np.asarray(ALPHA, dtype=dt)
    Q = np.zeros((B, T + 1, n, n), dt)
    q = np.zeros((B, T + 1, n), dt)
    R = np.zeros((B, T + 1, m, m), dt)
    r = np.zeros((B, T + 1, m), dt)
    M = np.zeros((B, T + 1, n, m), dt)
    ng = goal.shape[-1]          # the staging cost sees xc[:ng] only (cost_model.py:24-25); ng < n with a carry
    d = X[:, :T, :ng] - goal[:, :T]
    s = np.sqrt(np.sum(d * d, -1) + a * a)
    q[:, :T, :ng] = w[1] * d / s[..., None]
    # w1 (I / s - d d^T / s^3), built in place (n = 1024: the temporaries of the one-line form are
    # several GB); the operation sequence per entry is that of the formula
    Qs = Q[:, :T] if ng == n else np.empty((B, T, ng, ng), dt)
    np.multiply(d[..., :, None], d[..., None, :], out=Qs)
    Qs /= (s**3)[..., None, None]
    np.negative(Qs, out=Qs)
    di = np.arange(ng)
    Qs[..., di, di] += (1.0 / s)[..., None].astype(dt)
    Qs *= w[1]
    if ng != n:
        Q[:, :T, :ng, :ng] = Qs
    su = np.sqrt(np.sum(U * U, -1) + a * a)
    r[:, :T] = w[0] * U / su[..., None]
    R[:, :T] = w[0] * (
        np.eye(m, dtype=dt) / su[..., None, None]
        - U[..., :, None] * U[..., None, :] / (su**3)[..., None, None]
    )
    y, zs = mlp_forward(cmlp, X[:, T])
    Jc = mlp_input_jacobian(cmlp, zs)  # (B,f,n)
    q[:, T] = 2 * w[2] * np.einsum("bfn,bf->bn", Jc, y)
    Q[:, T] = 2 * w[2] * np.einsum("bfn,bfk->bnk", Jc, Jc)
    return Q, q, R, r, M


def linearize_dynamics(dyn, X, U):
    """A (B,T+1,n,n), Bm (B,T+1,n,m) at (X[t], pad(U)[t]) for t = 0..T."""
    B, T, m = U.shape
    n = X.shape[-1]
    Up = pad(U)
    A, Bm = dynamics_jacobians(
        dyn, X.reshape(B * (T + 1), n), Up.reshape(B * (T + 1), m)
    )
    return A.reshape(B, T + 1, n, n), Bm.reshape(B, T + 1, n, m)


def get_lqr_params(dyn, cmlp, mpc_w, goal, X, U):
    Q, q, R, r, M = cost_quadratize(cmlp, mpc_w, goal, X, U)
    A, Bm = linearize_dynamics(dyn, X, U)
    return Q, q, R, r, M, A, Bm


# --------------------------------------------------------------------------
# small dense linear algebra with JAX failure semantics
# --------------------------------------------------------------------------
def cholesky_lower(G):
    """Batched lower Cholesky; a non-positive pivot yields NaN (as
    jax.scipy.linalg.cho_factor does), never an exception."""
    G = np.array(G, copy=True)
    m = G.shape[-1]
    L = np.zeros_like(G)
    with np.errstate(invalid="ignore", divide="ignore"):
        for j in range(m):
            s = G[..., j, j] - np.sum(L[..., j, :j] ** 2, -1)
            d = np.sqrt(s)
            L[..., j, j] = d
            for i in range(j + 1, m):
                L[..., i, j] = (
                    G[..., i, j] - np.sum(L[..., i, :j] * L[..., j, :j], -1)
                ) / d
    return L


def cho_solve(L, rhs):
    """Solve (L L^T) x = rhs, rhs (..., m, k)."""
    m = L.shape[-1]
    y = np.zeros_like(rhs)
    with np.errstate(invalid="ignore", divide="ignore"):
        for i in range(m):
            y[..., i, :] = (
                rhs[..., i, :]
                - np.einsum("...j,...jk->...k", L[..., i, :i], y[..., :i, :])
            ) / L[..., i, i][..., None]
        x = np.zeros_like(rhs)
        for i in range(m - 1, -1, -1):
            x[..., i, :] = (
                y[..., i, :]
                - np.einsum("...j,...jk->...k", L[..., i + 1 :, i], x[..., i + 1 :, :])
            ) / L[..., i, i][..., None]
    return x


def _sym(x):
    return (x + np.swapaxes(x, -1, -2)) / 2


# --------------------------------------------------------------------------
# [trajax lqr_step / tvlqr]
# --------------------------------------------------------------------------
def lqr_step(P, p, Q, q, R, r, M, A, Bm, delta=1e-8):
    """One Riccati step (c == 0: the trajectory is dynamically feasible)."""
    At = np.swapaxes(A, -1, -2)
    Bt = np.swapaxes(Bm, -1, -2)
    AtP = At @ P
    AtPA = _sym(AtP @ A)
    BtP = Bt @ P
    BtPA = BtP @ A
    G = _sym(R + BtP @ Bm)
    H = BtPA + np.swapaxes(M, -1, -2)
    h = r + np.einsum("...nm,...n->...m", Bm, p)
    m = G.shape[-1]
    L = cholesky_lower(G + np.asarray(delta, G.dtype) * np.eye(m, dtype=G.dtype))
    Kk = -cho_solve(L, np.concatenate([H, h[..., None]], axis=-1))
    K, k = Kk[..., :-1], Kk[..., -1]
    H_GK = H + G @ K
    Kt = np.swapaxes(K, -1, -2)
    Pn = _sym(Q + AtPA + np.swapaxes(H_GK, -1, -2) @ K + Kt @ H)
    pn = (
        q
        + np.einsum("...ij,...j->...i", At, p)
        + np.einsum("...mn,...m->...n", H_GK, k)
        + np.einsum("...mn,...m->...n", K, h)
    )
    return Pn, pn, K, k


def tvlqr(Q, q, R, r, M, A, Bm):
    """Backward pass.  K (B,T,m,n), k (B,T,m), P (B,T+1,n,n), p (B,T+1,n)."""
    B, T1, n, _ = Q.shape
    T = T1 - 1
    m = R.shape[-1]
    K = np.zeros((B, T, m, n), Q.dtype)
    k = np.zeros((B, T, m), Q.dtype)
    P = np.zeros((B, T + 1, n, n), Q.dtype)
    p = np.zeros((B, T + 1, n), Q.dtype)
    P[:, T], p[:, T] = Q[:, T], q[:, T]
    for t in range(T - 1, -1, -1):
        P[:, t], p[:, t], K[:, t], k[:, t] = lqr_step(
            P[:, t + 1], p[:, t + 1], Q[:, t], q[:, t], R[:, t], r[:, t],
            M[:, t], A[:, t], Bm[:, t],
        )
    return K, k, P, p


def adjoint(A, Bm, q, r):
    """[trajax adjoint] control gradient g (B,T,m) and adjoints lam (B,T+1,n)."""
    B, T1, n = q.shape
    T = T1 - 1
    lam = np.zeros((B, T + 1, n), q.dtype)
    g = np.zeros((B, T, r.shape[-1]), q.dtype)
    lam[:, T] = q[:, T]
    for t in range(T - 1, -1, -1):
        g[:, t] = r[:, t] + np.einsum("bnm,bn->bm", Bm[:, t], lam[:, t + 1])
        lam[:, t] = q[:, t] + np.einsum("bij,bi->bj", A[:, t], lam[:, t + 1])
    return g, lam


# --------------------------------------------------------------------------
# [trajax ddp_rollout / line_search_ddp]
# --------------------------------------------------------------------------
def ddp_rollout(dyn, X, U, K, k, alpha):
    """u = U_t + alpha k_t + K_t (x_new - X_t);  alpha is (B,)."""
    B, T, m = U.shape
    Xn = np.empty_like(X)
    Un = np.empty_like(U)
    Xn[:, 0] = X[:, 0]
    with np.errstate(invalid="ignore", over="ignore"):
        for t in range(T):
            du = alpha[:, None] * k[:, t] + np.einsum(
                "bmn,bn->bm", K[:, t], Xn[:, t] - X[:, t]
            )
            Un[:, t] = U[:, t] + du
            Xn[:, t + 1], _ = dynamics_predict(dyn, Xn[:, t], Un[:, t])
    return Xn, Un


def line_search_ddp(dyn, cmlp, mpc_w, goal, X, U, K, k, obj, alpha_0, alpha_min,
                    active=None):
    """Backtracking line search.  Returns X, U, obj, alpha (per trajectory).

    Per trajectory: alpha starts at alpha_0 and is halved after every trial; the
    loop runs while obj_new >= obj and alpha > alpha_min; a NaN trial objective
    counts as no improvement; a trial is accepted only on a strict decrease.
    The alpha returned is the one AFTER the last halving (so an accepted
    alpha_0 step returns alpha_0 / 2).
    """
    B = X.shape[0]
    dt = X.dtype
    obj = np.where(np.isnan(obj), np.inf, obj).astype(dt)
    Xr, Ur = X.copy(), U.copy()
    objr = obj.copy()
    alpha = np.full((B,), alpha_0, dt)
    run = np.ones((B,), bool) if active is None else active.copy()
    # loop condition at entry: objr >= obj is true, alpha_0 > alpha_min
    run &= alpha > alpha_min
    while run.any():
        Xn, Un = ddp_rollout(dyn, X, U, K, k, alpha)
        with np.errstate(invalid="ignore", over="ignore"):
            on = np.sum(evaluate(cmlp, mpc_w, goal, Xn, Un), axis=1)
        on = np.where(np.isnan(on), obj, on).astype(dt)
        acc = run & (on < obj)
        Xr[acc], Ur[acc] = Xn[acc], Un[acc]
        objr = np.where(run, np.minimum(on, obj), objr).astype(dt)
        alpha = np.where(run, 0.5 * alpha, alpha).astype(dt)
        run = run & (objr >= obj) & (alpha > alpha_min)
    return Xr, Ur, objr, alpha


# --------------------------------------------------------------------------
# [trajax ilqr_base]  (a6)
# --------------------------------------------------------------------------
def ilqr(dyn, cmlp, mpc_w, goal, x0, U, kwargs=None, trace=None):
    """Batched iLQR; per-trajectory semantics identical to jax.vmap of
    trajax.optimizers.ilqr (a stopped trajectory is frozen).

    Returns X, U, obj, gradient, adjoints, lqr, iteration -- the 7-tuple
    unpacked at policy/optimizers.py:55.
    """
    kw = dict(ILQR_KWARGS)
    if kwargs:
        kw.update(kwargs)
    if kw["make_psd"]:
        raise NotImplementedError("make_psd=True is not on the reference path")
    B, T, m = U.shape
    dt = x0.dtype
    U = U.astype(dt).copy()
    X = rollout(dyn, U, x0)
    obj = np.sum(evaluate(cmlp, mpc_w, goal, X, U), axis=1)
    lqr = list(get_lqr_params(dyn, cmlp, mpc_w, goal, X, U))
    grad, adj = adjoint(lqr[5], lqr[6], lqr[1], lqr[3])
    alpha = np.full((B,), kw["alpha_0"], dt)
    it = np.zeros((B,), np.int32)
    obj_step = np.full((B,), np.inf, dt)
    U_step = np.full((B,), np.inf, dt)

    crit = {}

    def cont():
        with np.errstate(invalid="ignore", over="ignore"):
            gn = np.sqrt(np.sum(grad * grad, axis=(1, 2)))
        gn = np.where(np.isnan(gn), np.inf, gn)
        aobj = np.abs(obj) + 1.0
        un = np.sqrt(np.sum(U * U, axis=(1, 2))) + 1.0
        progressing = (obj_step > kw["obj_step_threshold"] * aobj) & (
            U_step > kw["inputs_step_threshold"] * un
        )
        potential = (gn > kw["grad_norm_threshold"]) & (
            gn > kw["relative_grad_norm_threshold"] * aobj
        )
        # (the criterion's quantities, for tests that leave out trajectories decided within rounding of a threshold)
        crit.update(gn=gn.copy(), aobj=aobj.copy(), un=un.copy(), obj_step=obj_step.copy(), U_step=U_step.copy())
        return (it < kw["maxiter"]) & progressing & potential & (alpha > kw["alpha_min"])

    while True:
        act = cont()
        if trace is not None:
            trace.append(dict(active=act.copy(), obj=obj.copy(), alpha=alpha.copy(), crit=dict(crit)))
        if not act.any():
            break
        Q, q, R, r, M, A, Bm = lqr
        with np.errstate(invalid="ignore", over="ignore", divide="ignore"):
            K, k, _, _ = tvlqr(Q, q, R, r, M, A, Bm)
        Xn, Un, objn, alphan = line_search_ddp(
            dyn, cmlp, mpc_w, goal, X, U, K, k, obj,
            kw["alpha_0"], kw["alpha_min"], active=act,
        )
        # frozen trajectories keep everything
        a3 = act[:, None, None]
        U_step = np.where(act, np.sqrt(np.sum((Un - U) ** 2, axis=(1, 2))), U_step).astype(dt)
        obj_step = np.where(act, np.abs(objn - obj), obj_step).astype(dt)
        X = np.where(a3, Xn, X)
        U = np.where(a3, Un, U)
        obj = np.where(act, objn, obj).astype(dt)
        alpha = np.where(act, alphan, alpha).astype(dt)
        new = get_lqr_params(dyn, cmlp, mpc_w, goal, X, U)
        for i in range(7):
            sel = act.reshape((B,) + (1,) * (new[i].ndim - 1))
            lqr[i] = np.where(sel, new[i], lqr[i])
        g2, a2 = adjoint(lqr[5], lqr[6], lqr[1], lqr[3])
        grad = np.where(a3, g2, grad)
        adj = np.where(a3, a2, adj)
        it = it + act.astype(np.int32)
    return X, U, obj, grad, adj, tuple(lqr), it


# --------------------------------------------------------------------------
# upper-level losses (a13, a16) and their state gradients
# --------------------------------------------------------------------------
def l2_loss(X, desired):
    """norm/l2_policy.py:12-18: sum_dims mean_t (x - x*)^2."""
    d = X[..., : desired.shape[-1]] - desired
    return np.sum(np.mean(d * d, axis=1), axis=-1)


def l2_loss_grad_x(X, desired):
    T1 = X.shape[1]
    g = np.zeros_like(X)
    g[..., : desired.shape[-1]] = 2 * (X[..., : desired.shape[-1]] - desired) / T1
    return g


# --------------------------------------------------------------------------
# critic (a14-a16): LSTM(F) over the T+1 rows, final h -> relu head -> score
# --------------------------------------------------------------------------
def critic_forward(cr, xseq, keep=False):
    """cr = dict(Wx (n,4F), Wh (F,4F), b (4F), head [(W,b),...]); gate order
    i, f, g, o (flax OptimizedLSTMCell: i,f,o sigmoid; g tanh; c' = f c + i g;
    h' = o tanh(c'); zero initial carry, critic/nn.py:16-20,28-42)."""
    Wx, Wh, b = cr["Wx"], cr["Wh"], cr["b"]
    B, T1, n = xseq.shape
    F = Wh.shape[0]
    dt = xseq.dtype
    c = np.zeros((B, F), dt)
    h = np.zeros((B, F), dt)
    cache = []
    for t in range(T1):
        z = xseq[:, t] @ Wx + h @ Wh + b
        i = sigmoid(z[:, :F])
        f = sigmoid(z[:, F : 2 * F])
        g = np.tanh(z[:, 2 * F : 3 * F])
        o = sigmoid(z[:, 3 * F :])
        c_new = f * c + i * g
        tc = np.tanh(c_new)
        h_new = o * tc
        if keep:
            cache.append((h, c, i, f, g, o, tc))
        c, h = c_new, h_new
    score, hz = mlp_forward(cr["head"], h)
    if keep:
        return score[:, 0], (cache, h, hz)
    return score[:, 0]


def critic_backward(cr, xseq, dscore, saved):
    """BPTT.  dscore (B,) = d loss / d score.  Returns (grads dict, dxseq)."""
    cache, hT, hz = saved
    Wx, Wh = cr["Wx"], cr["Wh"]
    B, T1, n = xseq.shape
    F = Wh.shape[0]
    head = cr["head"]
    # head backward
    acts = [hT] + [np.maximum(z, 0) for z in hz]
    ghead = [None] * len(head)
    d = dscore[:, None]
    for li in range(len(head) - 1, -1, -1):
        W, _ = head[li]
        ghead[li] = (acts[li].T @ d, d.sum(0))
        d = d @ W.T
        if li > 0:
            d = d * (hz[li - 1] > 0)
    dh = d
    dc = np.zeros_like(dh)
    gWx = np.zeros_like(Wx)
    gWh = np.zeros_like(Wh)
    gb = np.zeros_like(cr["b"])
    dx = np.zeros_like(xseq)
    for t in range(T1 - 1, -1, -1):
        h_prev, c_prev, i, f, g, o, tc = cache[t]
        do = dh * tc
        dc = dc + dh * o * (1 - tc * tc)
        di = dc * g
        df = dc * c_prev
        dg = dc * i
        dz = np.concatenate(
            [di * i * (1 - i), df * f * (1 - f), dg * (1 - g * g), do * o * (1 - o)],
            axis=1,
        )
        gWx += xseq[:, t].T @ dz
        gWh += h_prev.T @ dz
        gb += dz.sum(0)
        dx[:, t] = dz @ Wx.T
        dh = dz @ Wh.T
        dc = dc * f
    return dict(Wx=gWx, Wh=gWh, b=gb, head=ghead), dx


def critic_loss_and_grad(cr, xseq, label):
    """a15: gan/js_policy.py:41-58.  mean_b -log p, p = sigma(s) if label > 0
    else 1 - sigma(s).  Returns (loss, grads) with grads already / B."""
    B = xseq.shape[0]
    s, saved = critic_forward(cr, xseq, keep=True)
    p = sigmoid(s)
    pp = np.where(label > 0, p, 1 - p)
    with np.errstate(divide="ignore"):
        loss = np.mean(-np.log(pp))
    dscore = np.where(label > 0, -(1 - p), p).astype(xseq.dtype) / B
    grads, _ = critic_backward(cr, xseq, dscore, saved)
    return loss, grads


def generator_loss(cr, X):
    """a16: gan/js_policy.py:60-68 evaluated literally; the critic sees the x part of xc only
    (`jnp.split(xcseq, [x_size])`, :64-65)."""
    s = critic_forward(cr, X[..., : cr["Wx"].shape[0]])
    p = sigmoid(s)
    with np.errstate(divide="ignore"):
        return -np.log(p) + np.log(1 - p)


def generator_loss_grad_x(cr, X):
    """d/dX of generator_loss: d/ds = -(1-p) - p = -1 (zero on the carry columns of xc)."""
    nc = cr["Wx"].shape[0]
    Xc = np.ascontiguousarray(X[..., :nc])
    s, saved = critic_forward(cr, Xc, keep=True)
    _, dx = critic_backward(cr, Xc, -np.ones_like(s), saved)
    if nc == X.shape[-1]:
        return dx
    out = np.zeros_like(X)
    out[..., :nc] = dx
    return out


# --------------------------------------------------------------------------
# bilevel gradient (a8-a11), structured form
# --------------------------------------------------------------------------
def loss_grad_wrt_control(A, Bm, lx):
    """a8: B_t = Bm_t^T mu_{t+1}, mu_T = lx_T, mu_t = lx_t + A_t^T mu_{t+1}."""
    B, T1, n = lx.shape
    T = T1 - 1
    out = np.zeros((B, T, Bm.shape[-1]), lx.dtype)
    mu = lx[:, T].copy()
    for t in range(T - 1, -1, -1):
        out[:, t] = np.einsum("bnm,bn->bm", Bm[:, t], mu)
        mu = lx[:, t] + np.einsum("bij,bi->bj", A[:, t], mu)
    return out


def solve_sym_indef(G, rhs):
    """Plain Gaussian elimination with partial pivoting (LU), batched."""
    return np.linalg.solve(G, rhs)


def hessian_solve(lqr, Bvec):
    """a9 + a10's solve: H = A^{-1} B for A = d^2 J / dU^2, without forming A.

    For relu networks the dynamics are piecewise linear, so A is exactly the
    Hessian of the LQ model (Q, R, M, A_t, B_t); A^{-1} B is the minimiser of
    1/2 dU^T A dU - B^T dU, obtained by one Riccati sweep with linear term
    r~_t = -B_t and no regularisation, then a forward tangent roll.  Returns
    (H (B,T,m), dX (B,T+1,n)) with dX the tangent states for dU = H.
    """
    Q, _, R, _, M, A, Bm = lqr
    Bsz, T1, n, _ = Q.shape
    T = T1 - 1
    m = R.shape[-1]
    dt = Q.dtype
    P = Q[:, T].copy()
    p = np.zeros((Bsz, n), dt)
    K = np.zeros((Bsz, T, m, n), dt)
    k = np.zeros((Bsz, T, m), dt)
    for t in range(T - 1, -1, -1):
        At = np.swapaxes(A[:, t], -1, -2)
        Bt = np.swapaxes(Bm[:, t], -1, -2)
        BtP = Bt @ P
        G = _sym(R[:, t] + BtP @ Bm[:, t])
        H = BtP @ A[:, t] + np.swapaxes(M[:, t], -1, -2)
        h = -Bvec[:, t] + np.einsum("bnm,bn->bm", Bm[:, t], p)
        Kk = -solve_sym_indef(G, np.concatenate([H, h[..., None]], -1))
        K[:, t], k[:, t] = Kk[..., :-1], Kk[..., -1]
        Kt = np.swapaxes(K[:, t], -1, -2)
        P = _sym(Q[:, t] + At @ P @ A[:, t] + Kt @ H)
        p = np.einsum("bij,bj->bi", At, p) + np.einsum("bmn,bm->bn", H, k[:, t])
    dX = np.zeros((Bsz, T + 1, n), dt)
    H_out = np.zeros((Bsz, T, m), dt)
    for t in range(T):
        H_out[:, t] = k[:, t] + np.einsum("bmn,bn->bm", K[:, t], dX[:, t])
        dX[:, t + 1] = np.einsum("bij,bj->bi", A[:, t], dX[:, t]) + np.einsum(
            "bnm,bm->bn", Bm[:, t], H_out[:, t]
        )
    return H_out, dX


def hessian_apply(lqr, V):
    """(d^2 J / dU^2) V for V (B,T,m), through the LQ structure: tangent roll dx from dU = V, then
    the adjoint of the quadratic model.  Used to check a solve by its residual A H - B."""
    Q, _, R, _, M, A, Bm = lqr
    Bsz, T1, n, _ = Q.shape
    T = T1 - 1
    dx = np.zeros((Bsz, T + 1, n), Q.dtype)
    for t in range(T):
        dx[:, t + 1] = np.einsum("bij,bj->bi", A[:, t], dx[:, t]) + np.einsum(
            "bnm,bm->bn", Bm[:, t], V[:, t])
    out = np.zeros_like(V)
    lam = np.einsum("bij,bj->bi", Q[:, T], dx[:, T])
    for t in range(T - 1, -1, -1):
        out[:, t] = (np.einsum("bij,bj->bi", R[:, t], V[:, t])
                     + np.einsum("bnm,bn->bm", M[:, t], dx[:, t])
                     + np.einsum("bnm,bn->bm", Bm[:, t], lam))
        lam = (np.einsum("bij,bj->bi", Q[:, t], dx[:, t]) + np.einsum("bnm,bm->bn", M[:, t], V[:, t])
               + np.einsum("bij,bi->bj", A[:, t], lam))
    return out


def cost_vjp(cmlp, mpc_w, goal, X, U, Hc, dX):
    """a11: d/d(cost_params, mpc_weights) of  H . grad_U J(U; theta).

    H . grad_U J is the directional derivative of J along dU = H, i.e.
    sum_t [grad_x c_t . dX_t + grad_u c_t . H_t]; theta enters only through c
    (policy/optimizers.py:93-105 closes the dynamics over the outer params).
    Returns per-trajectory gradients: g_mpc (B,3), list of (gW (B,in,out),
    gb (B,out)) for the cost MLP.
    """
    B, T, m = U.shape
    dt = X.dtype
    raw = np.asarray(mpc_w, dtype=dt)
    w = sigmoid(raw)
    dw = w * (1 - w)
    a = np.asarray(ALPHA, dtype=dt)
    su = np.sqrt(np.sum(U * U, -1) + a * a)
    ng = goal.shape[-1]
    d = X[:, :T, :ng] - goal[:, :T]
    s = np.sqrt(np.sum(d * d, -1) + a * a)
    du_dir = np.sum(np.sum(U * Hc, -1) / su, axis=1)          # sum_t grad cu . H_t
    dx_dir = np.sum(np.sum(d * dX[:, :T, :ng], -1) / s, axis=1)    # sum_t grad cx . dX_t
    # terminal: F = 2 y . ydot, (y, ydot) the JVP of the cost MLP along dX_T
    x = X[:, T]
    xd = dX[:, T]
    acts, dacts, zs = [x], [xd], []
    qa, qd = x, xd
    for W, b in cmlp[:-1]:
        z = qa @ W + b
        zd = qd @ W
        mask = (z > 0).astype(dt)
        qa, qd = z * mask, zd * mask
        zs.append(mask)
        acts.append(qa)
        dacts.append(qd)
    Wl, bl = cmlp[-1]
    y = qa @ Wl + bl
    yd = qd @ Wl
    term_dir = 2 * np.sum(y * yd, -1)
    g_mpc = np.stack([dw[0] * du_dir, dw[1] * dx_dir, dw[2] * term_dir], axis=-1)
    scale = 2 * w[2]
    ybar, ydbar = yd * scale, y * scale  # adjoints of y and ydot
    grads = [None] * len(cmlp)
    grads[-1] = (
        acts[-1][:, :, None] * ybar[:, None, :] + dacts[-1][:, :, None] * ydbar[:, None, :],
        ybar,
    )
    abar, adbar = ybar @ Wl.T, ydbar @ Wl.T
    for li in range(len(cmlp) - 2, -1, -1):
        W, _ = cmlp[li]
        zbar, zdbar = abar * zs[li], adbar * zs[li]
        grads[li] = (
            acts[li][:, :, None] * zbar[:, None, :] + dacts[li][:, :, None] * zdbar[:, None, :],
            zbar,
        )
        abar, adbar = zbar @ W.T, zdbar @ W.T
    return g_mpc, grads


def bilevel_optimization(dyn, cmlp, mpc_w, goal, x0, init_U, loss="l2",
                         desired=None, critic=None, kwargs=None, sign=+1.0):
    """a10: policy/optimizers.py:34-75, per trajectory (no batch mean).

    ``sign=+1`` reproduces the reference as written (the reference
    returns +J_thetaU A^{-1} B; the implicit-function gradient would be -1).
    Returns dict(loss (B,), low_grad (B,T,m), g_mpc (B,3), g_cost [...], itr,
    X, U, H, dX).
    """
    X, U, obj, grad, adj, lqr, it = ilqr(dyn, cmlp, mpc_w, goal, x0, init_U, kwargs)
    if loss == "l2":
        lval = l2_loss(X, desired)
        lx = l2_loss_grad_x(X, desired)
    elif loss == "js":
        lval = generator_loss(critic, X)
        lx = generator_loss_grad_x(critic, X)
    else:
        raise ValueError(loss)
    Bvec = loss_grad_wrt_control(lqr[5], lqr[6], lx)
    # the dense Hessian the reference solves with is that of the LQ model with the dynamics' curvature in it
    Hc, dX = hessian_solve(second_order_lqr(dyn, lqr, adj, X, U), Bvec)
    g_mpc, g_cost = cost_vjp(cmlp, mpc_w, goal, X, U, Hc, dX)
    sg = np.asarray(sign, X.dtype)
    return dict(
        loss=lval, low_grad=grad, g_mpc=sg * g_mpc,
        g_cost=[(sg * gW, sg * gb) for gW, gb in g_cost],
        itr=it, X=X, U=U, H=Hc, dX=dX, Bvec=Bvec, lqr=lqr, obj=obj, adjoints=adj,
    )


def loss_and_grad(dyn, cmlp, mpc_w, goal, x0, init_U, **kw):
    """a12: policy/base.py:87-128 -- batch means of loss and gradients."""
    r = bilevel_optimization(dyn, cmlp, mpc_w, goal, x0, init_U, **kw)
    return (
        np.mean(r["loss"]),
        np.mean(r["g_mpc"], 0),
        [(gW.mean(0), gb.mean(0)) for gW, gb in r["g_cost"]],
        r,
    )


# --------------------------------------------------------------------------
# optimiser (a18) and Polyak (a19)
# --------------------------------------------------------------------------
def adam_clip_step(p, g, m, v, step, lr, max_norm=100.0, b1=0.9, b2=0.999, eps=1e-8):
    """optax.chain(clip_by_global_norm(100), adam(lr)) on one flat trainable
    vector (gan/runner.py:51-63).  ``step`` is the 1-based count after this
    update.  Returns (p, m, v)."""
    dt = p.dtype
    gn = np.sqrt(np.sum(g.astype(dt) ** 2))
    if not gn < max_norm:
        g = g / gn * np.asarray(max_norm, dt)
    m = (b1 * m + (1 - b1) * g).astype(dt)
    v = (b2 * v + (1 - b2) * g * g).astype(dt)
    mh = m / np.asarray(1 - b1**step, dt)
    vh = v / np.asarray(1 - b2**step, dt)
    upd = -np.asarray(lr, dt) * mh / (np.sqrt(vh) + np.asarray(eps, dt))
    return (p + upd).astype(dt), m, v


def polyak(prev, new, factor):
    """norm/cost_trainer.py:88-92."""
    return factor * prev + (1 - factor) * new


# --------------------------------------------------------------------------
# synthetic problem generator shared by tests and bench
# --------------------------------------------------------------------------
# ------------------------------------------------------------------------------------------------
# N2: expert sequence model inference (reference expert/nn.py:10-61, expert/expert_model.py:60-91,
# policy/eval.py:87-107)
# ------------------------------------------------------------------------------------------------
def expert_goal_states_init_actions(ex, history_X, T):
    """Batched get_goal_states_init_actions.  ex = dict(lstm=dict(Wx, Wh, b) | first=(W, b),
    head_x=[(W, b), ...], head_u=[(W, b), ...]).  history_X (B, hist+1, n): rows 0..hist-1 are fed
    teacher-forced (get_history_carry, expert_model.py:67-76), then the carry's last state is
    replaced by the current state history_X[:, -1] and the model runs T steps on its own predictions
    (get_carry_next_state_and_action_seq with teacher_forcing=False).  Returns goal (B, T+1, n)
    with goal[:, 0] = current state, and init_U (B, T, m)."""
    dt = _dt(history_X, ex["head_x"][0][0])
    B, h1, n = history_X.shape
    hist = h1 - 1
    m = ex["head_u"][-1][0].shape[1]
    lstm = ex.get("lstm")
    if lstm is not None:
        F = lstm["Wh"].shape[0]
        c = np.zeros((B, F), dtype=dt)
        h = np.zeros((B, F), dtype=dt)

    def mlp(layers, y):
        for l, (W, b) in enumerate(layers):
            y = y @ W + b
            if l < len(layers) - 1:
                y = np.maximum(y, 0)
        return y

    goal = np.zeros((B, T + 1, n), dtype=dt)
    U = np.zeros((B, T, m), dtype=dt)
    x = history_X[:, 0]
    for st in range(hist + T):
        if st <= hist:
            x = history_X[:, st]
        if lstm is not None:
            z = x @ lstm["Wx"] + h @ lstm["Wh"] + lstm["b"]
            i, f, g, o = (z[:, k * F:(k + 1) * F] for k in range(4))
            c = sigmoid(f) * c + sigmoid(i) * np.tanh(g)
            h = sigmoid(o) * np.tanh(c)
            y = h
        else:
            W0, b0 = ex["first"]
            y = np.maximum(x @ W0 + b0, 0)
        nx = mlp(ex["head_x"], y) + x
        u = np.tanh(mlp(ex["head_u"], y))
        if st >= hist:
            goal[:, st - hist + 1] = nx
            U[:, st - hist] = u
        x = nx
    goal[:, 0] = history_X[:, hist]
    return goal, U


def make_expert(rng, n, m, lstm_features=128, num_layers=3, num_hidden_units=128, dtype=np.float32,
                bias_scale=0.1):
    """LeCun-normal expert model (flax Dense / OptimizedLSTMCell kernel init; orthogonal recurrent
    init is replaced by LeCun: only the shapes matter for synthetic work)."""
    if lstm_features:
        F = lstm_features
        ex = {"lstm": dict(Wx=lecun_normal(rng, n, 4 * F, dtype), Wh=lecun_normal(rng, F, 4 * F, dtype),
                           b=(bias_scale * rng.standard_normal(4 * F)).astype(dtype))}
        y, L = F, num_layers
    else:
        ex = {"first": make_mlp(rng, [n, num_hidden_units], dtype, bias_scale)[0]}
        y, L = num_hidden_units, num_layers - 1
    ex["head_x"] = make_mlp(rng, [y] + [num_hidden_units] * (L - 1) + [n], dtype, bias_scale)
    ex["head_u"] = make_mlp(rng, [y] + [num_hidden_units] * (L - 1) + [m], dtype, bias_scale)
    return ex


# ------------------------------------------------------------------------------------------------
# N3: dynamics-model regression (reference norm/dynamics_trainer.py:14-90, utils.py:230-240)
# ------------------------------------------------------------------------------------------------
def dynamics_fit_loss_and_grad(dyn, xseq, useq, next_xseq, discount_factor, teacher_forcing):
    """Batch mean of predict_loss and its gradient w.r.t. the dynamics MLP.

    predict_loss (dynamics_trainer.py:14-47): x_in_t = teacher_forcing ? xseq[t] : pred_{t-1}
    (x_in_0 = xseq[0]); pred_t = MLP([x_in_t, u_t]) + x_in_t; loss = sum_d sum_t g^t (pred_t -
    next_x_t)^2 with the discount built by repeated multiplication (utils.discounted_sum).
    train_per_update (:74-86) takes the mean over the minibatch.  Returns (loss, [(gW, gb), ...]).
    """
    if isinstance(dyn, dict):
        return lstm_dynamics_fit_loss_and_grad(dyn, xseq, useq, next_xseq, discount_factor, teacher_forcing)
    dt = _dt(xseq, dyn[0][0])
    B, S, n = xseq.shape
    L = len(dyn)
    disc = np.ones(S, dtype=dt)
    g = dt.type(discount_factor)
    for t in range(1, S):
        disc[t] = disc[t - 1] * g
    acts = []          # per step: inputs of every layer (a_0 .. a_{L-1})
    preds = np.zeros((B, S, n), dtype=dt)
    x_in = xseq[:, 0]
    for t in range(S):
        if teacher_forcing:
            x_in = xseq[:, t]
        a = np.concatenate([x_in, useq[:, t]], axis=-1)
        layer_in = []
        for l, (W, b) in enumerate(dyn):
            layer_in.append(a)
            a = a @ W + b
            if l < L - 1:
                a = np.maximum(a, 0)
        pred = a + x_in
        preds[:, t] = pred
        acts.append(layer_in)
        x_in = pred
    diff = preds - next_xseq
    loss = (disc[None, :, None] * diff * diff).sum(axis=(1, 2))
    grads = [(np.zeros_like(W), np.zeros_like(b)) for W, b in dyn]
    lam = np.zeros((B, n), dtype=dt)
    for t in range(S - 1, -1, -1):
        gout = dt.type(2.0) * disc[t] * diff[:, t] + lam
        d = gout
        for l in range(L - 1, -1, -1):
            a_l = acts[t][l]
            grads[l][0][...] += a_l.T @ d
            grads[l][1][...] += d.sum(axis=0)
            d = d @ dyn[l][0].T
            if l > 0:
                d = d * (a_l > 0)
        lam = np.zeros((B, n), dtype=dt) if teacher_forcing else d[:, :n] + gout
    inv = dt.type(1.0) / dt.type(B)
    return loss.mean(), [(gW * inv, gb * inv) for gW, gb in grads]


def lstm_dynamics_fit_loss_and_grad(dl, xseq, useq, next_xseq, discount_factor, teacher_forcing):
    """predict_loss (dynamics_trainer.py:14-47) for the LSTM dynamics variant: the scan carries (x_prev, carry);
    x_in_t = teacher_forcing ? xseq[t] : pred_{t-1}; [pred_t, carry] = f([x_in_t, carry], u_t) from the zero carry
    of the training policy (policy/base.py:31-38) -- the carry is never teacher-forced.  Batch mean of the loss
    and its gradient: dict(Wx, Wh, b, tail=[(gW, gb), ...])."""
    dt = _dt(xseq, dl["Wx"])
    B, S, nx = xseq.shape
    F = dl["Wh"].shape[0]
    tail = dl["tail"]
    L = len(tail)
    disc = np.ones(S, dtype=dt)
    g_ = dt.type(discount_factor)
    for t in range(1, S):
        disc[t] = disc[t - 1] * g_
    c = np.zeros((B, F), dt)
    h = np.zeros((B, F), dt)
    x_in = xseq[:, 0]
    saved = []
    preds = np.zeros((B, S, nx), dt)
    for t in range(S):
        if teacher_forcing:
            x_in = xseq[:, t]
        q0 = np.concatenate([x_in, useq[:, t]], -1)
        z = q0 @ dl["Wx"] + h @ dl["Wh"] + dl["b"]
        i, f, o = sigmoid(z[:, :F]), sigmoid(z[:, F:2 * F]), sigmoid(z[:, 3 * F:])
        gg = np.tanh(z[:, 2 * F:3 * F])
        c2 = f * c + i * gg
        tc = np.tanh(c2)
        h2 = o * tc
        a = h2
        layer_in = []
        for l, (W, b) in enumerate(tail):
            layer_in.append(a)
            a = a @ W + b
            if l < L - 1:
                a = np.maximum(a, 0)
        pred = a + x_in
        preds[:, t] = pred
        saved.append((q0, h, c, i, f, gg, o, tc, layer_in))
        c, h, x_in = c2, h2, pred
    diff = preds - next_xseq
    loss = (disc[None, :, None] * diff * diff).sum(axis=(1, 2))
    gWx, gWh, gb = np.zeros_like(dl["Wx"]), np.zeros_like(dl["Wh"]), np.zeros_like(dl["b"])
    gtail = [(np.zeros_like(W), np.zeros_like(b)) for W, b in tail]
    lam = np.zeros((B, nx), dt)          # d loss / d pred_t through the next step's input
    dc = np.zeros((B, F), dt)
    dh = np.zeros((B, F), dt)
    for t in range(S - 1, -1, -1):
        q0, hp, cp, i, f, gg, o, tc, layer_in = saved[t]
        gout = dt.type(2.0) * disc[t] * diff[:, t] + lam
        d = gout
        for l in range(L - 1, -1, -1):
            a_l = layer_in[l]
            gtail[l][0][...] += a_l.T @ d
            gtail[l][1][...] += d.sum(0)
            d = d @ tail[l][0].T
            if l > 0:
                d = d * (a_l > 0)
        dh2 = d + dh
        dc2 = dc + dh2 * o * (1 - tc * tc)
        dz = np.concatenate([dc2 * gg * i * (1 - i), dc2 * cp * f * (1 - f), dc2 * i * (1 - gg * gg),
                             dh2 * tc * o * (1 - o)], -1)
        gWx += q0.T @ dz
        gWh += hp.T @ dz
        gb += dz.sum(0)
        dq0 = dz @ dl["Wx"].T
        dh = dz @ dl["Wh"].T
        dc = dc2 * f
        lam = np.zeros((B, nx), dt) if teacher_forcing else dq0[:, :nx] + gout
    inv = dt.type(1.0) / dt.type(B)
    return loss.mean(), dict(Wx=gWx * inv, Wh=gWh * inv, b=gb * inv,
                             tail=[(gW * inv, gb_ * inv) for gW, gb_ in gtail])


def lecun_normal(rng, fan_in, fan_out, dtype):
    return (rng.standard_normal((fan_in, fan_out)) / np.sqrt(fan_in)).astype(dtype)


def make_mlp(rng, sizes, dtype, bias_scale=0.0):
    layers = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        layers.append(
            (lecun_normal(rng, a, b, dtype),
             (bias_scale * rng.standard_normal(b)).astype(dtype))
        )
    return layers


def make_lstm_dynamics(rng, nx, m, F, tail_hidden, dtype, bias_scale=0.0):
    """Parameters of the LSTM dynamics variant (dynamics/nn.py:37-57): the cell on q0 = [x, u] and the relu
    MLP tail h' -> nx (num_layers - 1 hidden Dense layers, then Dense(nx))."""
    return dict(
        Wx=lecun_normal(rng, nx + m, 4 * F, dtype), Wh=lecun_normal(rng, F, 4 * F, dtype),
        b=(bias_scale * rng.standard_normal(4 * F)).astype(dtype),
        tail=make_mlp(rng, (F,) + tuple(tail_hidden) + (nx,), dtype, bias_scale))


def make_problem(n, m, T, B, seed=0, dtype=np.float32, dyn_hidden=(200, 200, 200),
                 cost_hidden=(128, 128), cost_fout=10, lstm_features=64,
                 head_hidden=(), bias_scale=0.0, dyn_lstm=0):
    """dyn_lstm = F > 0: the LSTM dynamics variant; `n` stays the x size, the state xc = [x, c, h] has
    n + 2F entries (x0 carries a zero carry, dynamics_model.py:20-22), the cost MLP takes xc, goals /
    true sequences / the critic keep n columns."""
    rng = np.random.default_rng(seed)
    if dyn_lstm:
        pb = _make_problem_lstm(rng, n, m, T, B, dtype, dyn_hidden, cost_hidden, cost_fout, lstm_features,
                                head_hidden, bias_scale, dyn_lstm)
        return pb
    dyn = make_mlp(rng, (n + m,) + tuple(dyn_hidden) + (n,), dtype, bias_scale)
    cmlp = make_mlp(rng, (n,) + tuple(cost_hidden) + (cost_fout,), dtype, bias_scale)
    F = lstm_features
    critic = dict(
        Wx=lecun_normal(rng, n, 4 * F, dtype),
        Wh=lecun_normal(rng, F, 4 * F, dtype),
        b=(bias_scale * rng.standard_normal(4 * F)).astype(dtype),
        head=make_mlp(rng, (F,) + tuple(head_hidden) + (1,), dtype, bias_scale),
    )
    return dict(
        n=n, m=m, T=T, B=B, dyn=dyn, cmlp=cmlp, critic=critic,
        mpc_w=np.array([-2.0, 3.0, -3.0], dtype),
        x0=rng.standard_normal((B, n)).astype(dtype),
        U=np.tanh(rng.standard_normal((B, T, m))).astype(dtype),
        goal=rng.standard_normal((B, T + 1, n)).astype(dtype),
        true_seq=rng.standard_normal((B, T + 1, n)).astype(dtype),
    )


def _make_problem_lstm(rng, nx, m, T, B, dtype, dyn_hidden, cost_hidden, cost_fout, lstm_features,
                       head_hidden, bias_scale, Fd):
    N = nx + 2 * Fd
    dyn = make_lstm_dynamics(rng, nx, m, Fd, dyn_hidden, dtype, bias_scale)
    cmlp = make_mlp(rng, (N,) + tuple(cost_hidden) + (cost_fout,), dtype, bias_scale)
    F = lstm_features
    critic = dict(
        Wx=lecun_normal(rng, nx, 4 * F, dtype), Wh=lecun_normal(rng, F, 4 * F, dtype),
        b=(bias_scale * rng.standard_normal(4 * F)).astype(dtype),
        head=make_mlp(rng, (F,) + tuple(head_hidden) + (1,), dtype, bias_scale))
    x0 = np.concatenate([rng.standard_normal((B, nx)), np.zeros((B, 2 * Fd))], -1).astype(dtype)
    return dict(
        n=N, nx=nx, m=m, T=T, B=B, dyn=dyn, cmlp=cmlp, critic=critic,
        mpc_w=np.array([-2.0, 3.0, -3.0], dtype), x0=x0,
        U=np.tanh(rng.standard_normal((B, T, m))).astype(dtype),
        goal=rng.standard_normal((B, T + 1, nx)).astype(dtype),
        true_seq=rng.standard_normal((B, T + 1, nx)).astype(dtype),
    )


def cast_problem(pb, dtype):
    def c(v):
        if isinstance(v, np.ndarray):
            return v.astype(dtype)
        if isinstance(v, (list, tuple)):
            return type(v)(c(e) for e in v)
        if isinstance(v, dict):
            return {k: c(e) for k, e in v.items()}
        return v
    return {k: c(v) for k, v in pb.items()}
