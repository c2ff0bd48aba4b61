"""Helpers shared by the GPU parity tests (TEST INFRASTRUCTURE)."""

import numpy as np

import gan_mpc_oracle as orc
from gan_mpc_amd import params as P

TOL = 1e-5  # BASELINE.json north_star: 1e-5 relative fp32


def rel_err(a, ref):
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    return float(np.abs(a - ref).max() / (np.abs(ref).max() + 1e-300))


def assert_parity(name, hip, o32, o64, tol=TOL, slack=4.0):
    """HIP fp32 vs the oracle: the fp64 oracle is the arbiter.  Pass if HIP is within `tol`
    (max-norm relative) of fp64, or no worse than `slack` x the oracle's own fp32 error -- a chained
    fp32 computation cannot be asked to beat fp32."""
    e_hip = rel_err(hip, o64)
    e_o32 = rel_err(o32, o64)
    assert np.isfinite(e_hip), f"{name}: non-finite"
    assert e_hip <= max(tol, slack * e_o32), (
        f"{name}: HIP err {e_hip:.3e} vs fp64; oracle-fp32 err {e_o32:.3e}; tol {tol:.1e}")
    return e_hip, e_o32


def problem(n, m, T, B, seed=0, dyn_hidden=(200, 200, 200), cost_hidden=(128, 128), cost_fout=10,
            head_hidden=(), bias_scale=0.1, out_scale=1.0):
    pb = orc.make_problem(n, m, T, B, seed=seed, dtype=np.float32, dyn_hidden=dyn_hidden,
                          cost_hidden=cost_hidden, cost_fout=cost_fout, head_hidden=head_hidden,
                          bias_scale=bias_scale)
    if out_scale != 1.0:  # "trained-like" residual dynamics: next_x = x + small
        W, b = pb["dyn"][-1]
        pb["dyn"][-1] = ((W * out_scale).astype(np.float32), (b * out_scale).astype(np.float32))
    return pb


def engine_for(pb, max_batch=None, critic=True):
    from gan_mpc_amd.engine import Engine
    dyn_dims = [pb["dyn"][0][0].shape[0]] + [W.shape[1] for W, _ in pb["dyn"]]
    cost_dims = [pb["cmlp"][0][0].shape[0]] + [W.shape[1] for W, _ in pb["cmlp"]]
    F = pb["critic"]["Wh"].shape[0]
    head = [F] + [W.shape[1] for W, _ in pb["critic"]["head"]]
    eng = Engine(pb["n"], pb["m"], pb["T"], dyn_dims, cost_dims, max_batch or pb["B"],
                 lstm_features=F if critic else 0, head_dims=head)
    eng.set_params(eng.to_dev(pb["mpc_w"]), eng.to_dev(P.pack_mlp(P.layers_to_tree(pb["dyn"]))),
                   eng.to_dev(P.pack_mlp(P.layers_to_tree(pb["cmlp"]))))
    return eng


def critic_flat(pb):
    return P.pack_critic(P.critic_dict_to_tree(pb["critic"]))


def near_kink(layers, q, thresh=3e-6):
    """(rows,) bool: some hidden pre-activation is within `thresh` (relative to that sample's layer scale)
    of the relu kink, where the derivative is discontinuous (hard part iii)."""
    _, zs = orc.mlp_forward(layers, q)
    bad = np.zeros(q.shape[0], bool)
    for z in zs:
        bad |= (np.abs(z) < thresh * np.abs(z).max(axis=1, keepdims=True)).any(axis=1)
    return bad


def pack_grads_cost(g_mpc, g_cost):
    out = [np.asarray(g_mpc).reshape(-1)]
    for gW, gb in g_cost:
        out += [np.asarray(gW).reshape(-1), np.asarray(gb).reshape(-1)]
    return np.concatenate(out)


def pack_grads_critic(g):
    out = [g["Wx"].reshape(-1), g["Wh"].reshape(-1), g["b"].reshape(-1)]
    for gW, gb in g["head"]:
        out += [np.asarray(gW).reshape(-1), np.asarray(gb).reshape(-1)]
    return np.concatenate(out)
