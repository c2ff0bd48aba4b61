"""GPU parity at the REAL configurations of BASELINE.json (full batch, full horizon): the kernels run
the whole batch -- persistent grids, tile loops, work-list line search -- and a random sample of
trajectories (they are independent) is compared with the fp32 / fp64 oracle; batch sums (critic step,
bilevel gradient) are compared over the whole batch.

  c3-bench    n=17  m=6  T=50  B=1024  bench.py's own inputs (LeCun-normal weights, zero biases)
  c3-trained  same shape, residual head scaled by 0.1 ("trained-like": the state stays O(1))
  c4-shard    n=376 m=17 T=50  B=512   one GPU's shard of C4 (4096 over 8)
  c5-shard    n=1024 m=64 T=100 B=64   C5's shape at a batch the oracle's single sampled trajectory and
                                       the test's time budget allow (the per-GPU shard is 1024)
  c5-full     n=1024 m=64 T=100 B=1024 the full per-GPU shard of C5: rollout + backward pass, critic step (whole-batch
                                       sums), gmpc_ilqr_solve(maxiter=1) with sampled trajectories (round 4)
Every entry point of the path runs at every shape (c5-full: all but the bilevel gradient): rollout + costs, backward
pass, critic step, gmpc_ilqr_solve(maxiter=1), gmpc_bilevel_grad."""

import numpy as np
import pytest
import torch

import gan_mpc_oracle as orc
import gpu_util as gu
from gan_mpc_amd import synthetic

pytestmark = pytest.mark.gpu

HEAD = (256, 256, 256)
# name: (n, m, T, B, samples, builder)
CONFIGS = {
    "c3-bench": (17, 6, 50, 1024, 16, "bench"),
    "c3-trained": (17, 6, 50, 1024, 16, 0.1),
    "c4-shard": (376, 17, 50, 512, 2, 0.3),
    "c5-shard": (1024, 64, 100, 64, 1, 0.3),
    # C5's full per-GPU shard (1024 trajectories) -- the batched GEMMs' block and XCD remaps depend on the batch --:
    # rollout + backward pass and the one-iteration solve with one / two sampled trajectories against the oracle, the
    # critic step on its 2048 sequences as whole-batch sums; the bilevel gradient stays at c5-shard (B = 64)
    "c5-full": (1024, 64, 100, 1024, 1, 0.3),
}
FULL_ONLY = ("c5-full",)          # configurations left out of test_bilevel_grad_full_shape


def _problem(name):
    n, m, T, B, ns, kind = CONFIGS[name]
    if kind == "bench":
        # exactly what bench.py feeds rank 0: trajectories of seed 1000, weights of seed 0
        pb = synthetic.make_problem(n, m, T, B, seed=1000, head_hidden=HEAD)
        w = synthetic.make_problem(n, m, T, 1, seed=0, head_hidden=HEAD)
        for key in ("dyn", "cmlp", "critic", "mpc_w"):
            pb[key] = w[key]
    else:
        pb = gu.problem(n, m, T, B, seed=23, head_hidden=HEAD, out_scale=kind)
    gu.set_config(f"{name} n={n} m={m} T={T} B={B}")
    return pb, ns


def _sub(pb, idx, dtype):
    """The oracle's problem restricted to the sampled trajectories."""
    q = dict(pb)
    for key in ("x0", "U", "goal", "true_seq"):
        q[key] = pb[key][idx]
    q["B"] = len(idx)
    return orc.cast_problem(q, dtype)


def _sample(pb, X, U, count, rng, extra=4):
    """`count` trajectory indices whose relu pre-activations along (X, U) keep clear of the kink."""
    B, T, n, m = pb["B"], pb["T"], pb["n"], pb["m"]
    cand = rng.permutation(B)[:min(B, extra * count + 4)]
    Xc = X[cand].astype(np.float64)
    Uc = U[cand].astype(np.float64)
    pb64 = orc.cast_problem(dict(dyn=pb["dyn"], cmlp=pb["cmlp"]), np.float64)
    q = np.concatenate([Xc[:, :T], Uc], -1).reshape(-1, n + m)
    bad = gu.near_kink(pb64["dyn"], q).reshape(len(cand), T).any(1) | gu.near_kink(pb64["cmlp"], Xc[:, T])
    good = cand[~bad]
    assert len(good) >= count, f"only {len(good)} of {len(cand)} candidates clear of a relu kink"
    return np.sort(good[:count])


@pytest.mark.parametrize("name", list(CONFIGS))
def test_rollout_and_backward_full_shape(name):
    pb, ns = _problem(name)
    eng = gu.engine_for(pb, critic=False)
    d = eng.to_dev
    n, m, T = pb["n"], pb["m"], pb["T"]
    try:
        Xd, costs = eng.rollout_cost(d(pb["x0"]), d(pb["U"]), d(pb["goal"]))
        out = eng.lqr_backward(Xd, d(pb["U"]), d(pb["goal"]), after_rollout=True)
        X = Xd.cpu().numpy()
        idx = _sample(pb, X, pb["U"], ns, np.random.default_rng(1))
        s32, s64 = _sub(pb, idx, np.float32), _sub(pb, idx, np.float64)
        X32 = orc.rollout(s32["dyn"], s32["U"], s32["x0"])
        X64 = orc.rollout(s64["dyn"], s64["U"], s64["x0"])
        gu.assert_parity("rollout X", X[idx], X32, X64)
        gu.assert_parity("rollout costs", costs.cpu().numpy()[idx],
                         orc.evaluate(s32["cmlp"], s32["mpc_w"], s32["goal"], X32, s32["U"]),
                         orc.evaluate(s64["cmlp"], s64["mpc_w"], s64["goal"], X64, s64["U"]))
        # both oracles linearise at the trajectory the kernel saw
        Xs = X[idx]
        lqr32 = orc.get_lqr_params(s32["dyn"], s32["cmlp"], s32["mpc_w"], s32["goal"], Xs, s32["U"])
        lqr64 = orc.get_lqr_params(s64["dyn"], s64["cmlp"], s64["mpc_w"], s64["goal"],
                                   Xs.astype(np.float64), s64["U"])
        ti = torch.as_tensor(idx, device=Xd.device)
        if n <= 64:
            AB = out["AB"][ti].cpu().numpy()
            gu.assert_parity("backward AB", AB, np.concatenate([lqr32[5][:, :T], lqr32[6][:, :T]], -1),
                             np.concatenate([lqr64[5][:, :T], lqr64[6][:, :T]], -1))
        with np.errstate(all="ignore"):
            K32, k32, _, _ = orc.tvlqr(*lqr32)
            K64, k64, _, _ = orc.tvlqr(*lqr64)
        g32, a32 = orc.adjoint(lqr32[5], lqr32[6], lqr32[1], lqr32[3])
        g64, a64 = orc.adjoint(lqr64[5], lqr64[6], lqr64[1], lqr64[3])
        gu.assert_parity("backward grad", out["grad"][ti].cpu().numpy(), g32, g64)
        gu.assert_parity("backward adjoints", out["adjoints"][ti].cpu().numpy(), a32, a64)
        gu.assert_parity("backward K", out["K"][ti].cpu().numpy(), K32, K64, ceiling=gu.GAIN_CEILING)
        gu.assert_parity("backward k", out["k"][ti].cpu().numpy(), k32, k64, ceiling=gu.GAIN_CEILING)
        gu.assert_gain_backward_error(lqr64, out["K"][ti].cpu().numpy(), out["k"][ti].cpu().numpy(), K32, k32)
    finally:
        eng.close()


@pytest.mark.parametrize("name", list(CONFIGS))
def test_critic_step_full_shape(name):
    """The critic half of the metric's step on 2B sequences (B true + the B rolled-out ones), whole batch."""
    pb, _ = _problem(name)
    eng = gu.engine_for(pb, critic=True)
    d = eng.to_dev
    B = pb["B"]
    try:
        Xd, _ = eng.rollout_cost(d(pb["x0"]), d(pb["U"]), d(pb["goal"]))
        X = Xd.cpu().numpy()
        if name == "c3-bench":      # the free-running state reaches ~1e7: keep the comparison finite
            assert np.isfinite(X).all()
        xseq = np.concatenate([pb["true_seq"], X], 0)
        label = np.concatenate([np.ones(B), -np.ones(B)]).astype(np.float32)
        cr64 = orc.cast_problem(dict(c=pb["critic"]), np.float64)["c"]
        # The batch gradient sums over 2B x 768 relu decisions of the head: a sequence whose pre-activation sits
        # within 3e-6 (relative) of a kink flips its mask on a 1e-7 perturbation of h_T and moves the sum by a finite
        # amount (hard part iii).  Such sequences -- a handful of the 2048 -- are left out of
        # the batch, for the kernels and for both oracles alike, exactly as the Jacobian tests leave out samples.
        with np.errstate(over="ignore"):
            _, saved = orc.critic_forward(cr64, xseq.astype(np.float64), keep=True)
        keep = ~gu.near_kink(cr64["head"][:-1], saved[1]) if len(cr64["head"]) > 1 else np.ones(2 * B, bool)
        assert keep.sum() >= 0.98 * 2 * B, f"{(~keep).sum()} of {2 * B} sequences near a relu kink"
        xseq, label = xseq[keep], label[keep]
        Bc = int(keep.sum())
        ls, gs = eng.critic_loss_grad(d(xseq), d(label), d(gu.critic_flat(pb)))
        with np.errstate(over="ignore"):
            l32, g32 = orc.critic_loss_and_grad(pb["critic"], xseq, label)
            l64, g64 = orc.critic_loss_and_grad(cr64, xseq.astype(np.float64), label.astype(np.float64))
        gu.assert_parity("critic loss", ls.cpu().numpy() / Bc, l32, l64)
        gu.assert_parity("critic grad", gs.cpu().numpy() / Bc, gu.pack_grads_critic(g32),
                         gu.pack_grads_critic(g64))
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["c3-trained", "c4-shard", "c5-shard", "c5-full"])
def test_ilqr_one_iteration_full_shape(name):
    """gmpc_ilqr_solve(maxiter=1) over the whole batch; sampled trajectories against the oracle's loop."""
    pb, ns = _problem(name)
    eng = gu.engine_for(pb, critic=False)
    d = eng.to_dev
    kw = {"maxiter": 1}
    try:
        out = eng.ilqr_solve(d(pb["x0"]), d(pb["U"]), d(pb["goal"]), kw)
        it = out["iterations"].cpu().numpy()
        idx = np.sort(np.random.default_rng(2).permutation(pb["B"])[:2 * ns])
        s32, s64 = _sub(pb, idx, np.float32), _sub(pb, idx, np.float64)
        with np.errstate(all="ignore"):
            r32 = orc.ilqr(s32["dyn"], s32["cmlp"], s32["mpc_w"], s32["goal"], s32["x0"], s32["U"], kw)
            r64 = orc.ilqr(s64["dyn"], s64["cmlp"], s64["mpc_w"], s64["goal"], s64["x0"], s64["U"], kw)
        np.testing.assert_array_equal(it[idx], r64[6])
        # a line-search branch decided by the last bit is not a parity failure: compare the trajectories
        # on which the fp32 and fp64 oracles accept the same step
        same = np.isclose(r32[2], r64[2], rtol=1e-3)
        assert same.any()
        ti = torch.as_tensor(idx[same], device=out["U"].device)
        # (c5-full: at n = 1024, T = 100 the NumPy fp32 oracle's own step is 1e-2 from the fp64 one on the sampled
        # trajectories -- the "no worse than 4 x the fp32 oracle" rule stays, its cap is 3e-2 there instead of 1e-2)
        cap = 3e-2 if name == "c5-full" else gu.GAIN_CEILING
        gu.assert_parity("ilqr U", out["U"][ti].cpu().numpy(), r32[1][same], r64[1][same], tol=1e-4, ceiling=cap)
        gu.assert_parity("ilqr X", out["X"][ti].cpu().numpy(), r32[0][same], r64[0][same], tol=1e-4, ceiling=cap)
        gu.assert_parity("ilqr obj", out["obj"][ti].cpu().numpy(), r32[2][same], r64[2][same], tol=1e-4, ceiling=cap)
    finally:
        eng.close()


@pytest.mark.parametrize("name,loss_kind", [("c3-trained", 0), ("c3-trained", 1), ("c4-shard", 0),
                                            ("c5-shard", 0), ("c5-shard", 1)])
def test_bilevel_grad_full_shape(name, loss_kind):
    """a8-a11 over the whole batch at the iterate a short solve reaches.  Per sampled trajectory: loss,
    Bvec, the fp64 residual of the structured Hessian solve, the tangent roll; over the WHOLE batch:
    the summed cost_vjp given the GPU's own (H, dX)."""
    pb, ns = _problem(name)
    eng = gu.engine_for(pb, critic=True)
    d = eng.to_dev
    n, m, T, B = pb["n"], pb["m"], pb["T"], pb["B"]
    try:
        out = eng.ilqr_solve(d(pb["x0"]), d(pb["U"]), d(pb["goal"]), {"maxiter": 2})
        crit = d(gu.critic_flat(pb))
        loss, gsum = eng.bilevel_grad(B, loss_kind, desired=d(pb["true_seq"]), critic=crit, sign=1.0)
        X = out["X"].cpu().numpy()
        U = out["U"].cpu().numpy()
        Hd = eng.debug_buffer(2, (B, T, m)).cpu().numpy()
        dXd = eng.debug_buffer(3, (B, T + 1, n)).cpu().numpy()
        Bvd = eng.debug_buffer(4, (B, T, m)).cpu().numpy()
        idx = _sample(pb, X, U, ns, np.random.default_rng(3))
        res = {}
        for dt in (np.float32, np.float64):
            s = _sub(pb, idx, dt)
            Xa, Ua = X[idx].astype(dt), U[idx].astype(dt)
            lqr = orc.get_lqr_params(s["dyn"], s["cmlp"], s["mpc_w"], s["goal"], Xa, Ua)
            if loss_kind == 0:
                lv, lx = orc.l2_loss(Xa, s["true_seq"]), orc.l2_loss_grad_x(Xa, s["true_seq"])
            else:
                lv, lx = orc.generator_loss(s["critic"], Xa), orc.generator_loss_grad_x(s["critic"], Xa)
            Bv = orc.loss_grad_wrt_control(lqr[5], lqr[6], lx)
            lqr = orc.second_order_lqr(s["dyn"], lqr, orc.adjoint(lqr[5], lqr[6], lqr[1], lqr[3])[1], Xa, Ua)
            with np.errstate(all="ignore"):
                Hc, _ = orc.hessian_solve(lqr, Bv)
            # the batch-summed a11 given the GPU's (H, dX), over ALL trajectories
            f = orc.cast_problem(dict(pb), dt)
            g_mpc, g_cost = orc.cost_vjp(f["cmlp"], f["mpc_w"], f["goal"], X.astype(dt), U.astype(dt),
                                         Hd.astype(dt), dXd.astype(dt))
            g = gu.pack_grads_cost(g_mpc.sum(0), [(a.sum(0), b.sum(0)) for a, b in g_cost])
            res[dt] = dict(lqr=lqr, loss=lv, Bv=Bv, H=Hc, g=g)
        s32, s64 = res[np.float32], res[np.float64]
        gu.assert_parity("bilevel loss", loss.cpu().numpy()[idx], s32["loss"], s64["loss"])
        # a8 stage-wise: the adjoint recursion given the GPU's own d loss / d X (the critic's input gradient is
        # checked by test_critic_score_vjp / test_critic_step_full_shape), then end to end
        lxd = eng.debug_buffer(11, (B, T + 1, n)).cpu().numpy()[idx]
        bv = {dt: orc.loss_grad_wrt_control(res[dt]["lqr"][5], res[dt]["lqr"][6], lxd.astype(dt)) for dt in res}
        gu.assert_parity("bilevel Bvec given d loss / d X", Bvd[idx], bv[np.float32], bv[np.float64])
        # end to end with the critic (JS): the critic's input gradient is a 1e-4-level fp32 quantity at these
        # shapes for BOTH fp32 routes ("critic grad" of test_critic_step_full_shape: 2.7e-4 HIP, 1.6e-4 NumPy at
        # c5-shard), and which of the two lands closer to fp64 on a sample of one or two trajectories changes with
        # the iterate -- the bar for this line is therefore a stated 1e-3, not a multiple of the NumPy error
        gu.assert_parity("bilevel Bvec", Bvd[idx], s32["Bv"], s64["Bv"], tol=1e-5 if loss_kind == 0 else 1e-3,
                         slack=4.0, ceiling=gu.GAIN_CEILING)

        def resid(H):
            r = orc.hessian_apply(s64["lqr"], H.astype(np.float64)) - s64["Bv"]
            return np.sqrt((r ** 2).sum((1, 2)) / (s64["Bv"] ** 2).sum((1, 2)))
        r_hip, r_o32 = resid(Hd[idx]), resid(s32["H"])
        gu._record(dict(stage="bilevel Hessian-solve residual |A H - B| / |B| (fp64)", config=gu.CURRENT_CONFIG[0],
                        e_hip=float(r_hip.max()), e_o32=float(r_o32.max()), tol=1e-4,
                        tol_used=float(max(1e-4, 10 * r_o32.max())), branch="tol" if r_hip.max() <= 1e-4 else "slack",
                        entries=int(r_hip.size), passed=bool((r_hip <= np.maximum(1e-4, 10 * r_o32)).all())))
        assert (r_hip <= np.maximum(1e-4, 10 * r_o32)).all(), (r_hip, r_o32)
        lq = s64["lqr"]
        dx = np.zeros((len(idx), T + 1, n))
        for t in range(T):
            dx[:, t + 1] = np.einsum("bij,bj->bi", lq[5][:, t], dx[:, t]) + np.einsum(
                "bnm,bm->bn", lq[6][:, t], Hd[idx][:, t].astype(np.float64))
        e = gu.rel_err(dXd[idx], dx)
        gu._record(dict(stage="bilevel tangent roll dX given H", config=gu.CURRENT_CONFIG[0], e_hip=e, e_o32=None,
                        tol=1e-4, tol_used=1e-4, branch="tol", entries=int(dx.size), passed=bool(e < 1e-4)))
        assert e < 1e-4
        gu.assert_parity("bilevel cost_vjp sum over the batch (given H, dX)", gsum.cpu().numpy(), s32["g"],
                         s64["g"])
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["c3-bench", "c3-trained"])
def test_batch_order_does_not_matter(name):
    """Size-independent property of the whole path at BASELINE's headline size: the trajectories of a batch are
    independent, so handing them over in another order returns the same bits in that order -- the rollout, the
    backward pass (Jacobian chain tiles mix two samples, the sweep and its helper wave are per trajectory) and five
    iterations of gmpc_ilqr_solve (speculative line-search rounds: the candidates of a trajectory land in other
    work-list slots and other 16-candidate workgroups)."""
    pb, _ = _problem(name)
    eng = gu.engine_for(pb, critic=False)
    d = eng.to_dev
    perm = np.random.default_rng(5).permutation(pb["B"])
    out = {}
    for tag, idx in (("id", np.arange(pb["B"])), ("perm", perm)):
        x0, U, goal = d(pb["x0"][idx]), d(pb["U"][idx]), d(pb["goal"][idx])
        X, costs = eng.rollout_cost(x0, U, goal)
        bw = eng.lqr_backward(X, U, goal, after_rollout=True)
        sol = eng.ilqr_solve(x0, U, goal, {"maxiter": 5})
        torch.cuda.synchronize()
        out[tag] = dict(X=X.cpu().numpy(), costs=costs.cpu().numpy(), K=bw["K"].cpu().numpy(), k=bw["k"].cpu().numpy(),
                        grad=bw["grad"].cpu().numpy(), adjoints=bw["adjoints"].cpu().numpy(),
                        sX=sol["X"].cpu().numpy(), sU=sol["U"].cpu().numpy(), sobj=sol["obj"].cpu().numpy(),
                        its=sol["iterations"].cpu().numpy())
    for key, a in out["id"].items():
        np.testing.assert_array_equal(a[perm], out["perm"][key], err_msg=key)


def test_the_critic_step_is_deterministic():
    """Run-to-run determinism at the headline size: the critic step (LSTM forward, head, BPTT with the weight
    gradients accumulated in the sweep, the head's weight gradients on the side stream, the reductions) twice on
    the same inputs returns the same bits -- every reduction has a fixed order, nothing is accumulated with
    floating-point atomics -- which is what keeps the parameter replicas of a multi-GPU run identical."""
    pb, _ = _problem("c3-bench")
    eng = gu.engine_for(pb, critic=True)
    d = eng.to_dev
    B = pb["B"]
    X, _ = eng.rollout_cost(d(pb["x0"]), d(pb["U"]), d(pb["goal"]))
    xseq = torch.cat([d(pb["true_seq"]), X])
    label = d(np.concatenate([np.ones(B), -np.ones(B)]).astype(np.float32))
    crit = d(gu.critic_flat(pb))
    outs = []
    for _ in range(3):
        loss, grad = eng.critic_loss_grad(xseq, label, crit)
        torch.cuda.synchronize()
        outs.append((loss.clone(), grad.clone()))
    for loss, grad in outs[1:]:
        assert torch.equal(loss, outs[0][0]) and torch.equal(grad, outs[0][1])
    assert torch.isfinite(outs[0][1]).all()
