"""Generates tests/golden/*.npz from the build oracle (oracle/gan_mpc_oracle.py) in float64.

These are outputs of THIS REPOSITORY'S ORACLE ("build oracle"), not of the JAX reference: the
reference ships no fixtures and cannot be run here.  They freeze the oracle so that
an accidental change to it is caught, and give the GPU tests fixed vectors that do not depend on the
oracle's code at test time.   Run:  python tests/golden/make_golden.py
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import gan_mpc_oracle as orc  # noqa: E402


def flat_layers(layers):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in layers])


def main():
    n, m, T, B = 4, 2, 6, 5
    pb = orc.make_problem(n, m, T, B, seed=2024, dtype=np.float64, dyn_hidden=(24, 40),
                          cost_hidden=(20,), cost_fout=6, lstm_features=64, head_hidden=(16,),
                          bias_scale=0.2)
    W, b = pb["dyn"][-1]
    pb["dyn"][-1] = (0.2 * W, 0.2 * b)          # contracting residual dynamics: iLQR converges
    X = orc.rollout(pb["dyn"], pb["U"], pb["x0"])
    costs = orc.evaluate(pb["cmlp"], pb["mpc_w"], pb["goal"], X, pb["U"])
    lqr = orc.get_lqr_params(pb["dyn"], pb["cmlp"], pb["mpc_w"], pb["goal"], X, pb["U"])
    K, k, P, p = orc.tvlqr(*lqr)
    grad, adj = orc.adjoint(lqr[5], lqr[6], lqr[1], lqr[3])
    sol = orc.ilqr(pb["dyn"], pb["cmlp"], pb["mpc_w"], pb["goal"], pb["x0"], pb["U"])
    # the oracle's own fp32 error on the same quantities (chained / ill-conditioned ones cannot be
    # asked to beat fp32): stored so the GPU test can apply the "no worse than 4x" rule offline
    p32 = orc.cast_problem(pb, np.float32)
    X32 = orc.rollout(p32["dyn"], p32["U"], p32["x0"])
    lqr32 = orc.get_lqr_params(p32["dyn"], p32["cmlp"], p32["mpc_w"], p32["goal"], X32, p32["U"])
    K32, k32, _, _ = orc.tvlqr(*lqr32)
    g32, a32 = orc.adjoint(lqr32[5], lqr32[6], lqr32[1], lqr32[3])

    def rel(a, ref):
        return float(np.abs(a - ref).max() / np.abs(ref).max())

    f32_err = dict(
        AB=rel(np.concatenate([lqr32[5][:, :T], lqr32[6][:, :T]], -1),
               np.concatenate([lqr[5][:, :T], lqr[6][:, :T]], -1)),
        K=rel(K32, K), k=rel(k32, k), grad=rel(g32, grad), adjoints=rel(a32, adj))
    label = np.array([1.0, -1.0, 1.0, -1.0, 1.0])
    closs, cgrad = orc.critic_loss_and_grad(pb["critic"], pb["true_seq"], label)
    score = orc.critic_forward(pb["critic"], pb["true_seq"])
    out = dict(
        n=n, m=m, T=T, B=B, dyn_dims=np.array([n + m, 24, 40, n]), cost_dims=np.array([n, 20, 6]),
        head_dims=np.array([64, 16, 1]),
        dyn_flat=flat_layers(pb["dyn"]), cost_flat=flat_layers(pb["cmlp"]), mpc_w=pb["mpc_w"],
        critic_flat=np.concatenate([pb["critic"]["Wx"].ravel(), pb["critic"]["Wh"].ravel(),
                                    pb["critic"]["b"].ravel(), flat_layers(pb["critic"]["head"])]),
        x0=pb["x0"], U=pb["U"], goal=pb["goal"], true_seq=pb["true_seq"], label=label,
        X=X, costs=costs, AB=np.concatenate([lqr[5][:, :T], lqr[6][:, :T]], -1), K=K, k=k,
        grad=grad, adjoints=adj,
        ilqr_X=sol[0], ilqr_U=sol[1], ilqr_obj=sol[2], ilqr_iters=sol[6],
        critic_loss=closs, critic_score=score,
        **{f"f32err_{k_}": v_ for k_, v_ in f32_err.items()},
        critic_grad=np.concatenate([cgrad["Wx"].ravel(), cgrad["Wh"].ravel(), cgrad["b"].ravel()]
                                   + [t.ravel() for Wb in cgrad["head"] for t in Wb]),
    )
    np.savez_compressed(os.path.join(HERE, "build_oracle_small.npz"), **out)
    print("wrote build_oracle_small.npz", {k: np.shape(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
