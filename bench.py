#!/usr/bin/env python3
"""bench.py -- trajectories/sec of the GAN-MPC inner-loop step on MI355X.

One "step"  on a batch of B synthetic trajectories per GPU:
  1. rollout X from (x0, U) through the dynamics MLP + per-step costs          (gmpc_rollout_cost)
  2. backward: linearise + quadratise + Riccati gains + adjoint gradient       (gmpc_lqr_backward_after_rollout)
  3. critic step on 2B sequences (B "true" + the B rolled-out ones): BCE loss, BPTT,
     [all-reduce of loss+grads over ranks], clip-by-global-norm(100) + Adam     (gmpc_critic_loss_grad, gmpc_adam_clip_step)
All inputs are resident in HBM before the timed region.  fp32 throughout (the reference's dtype).

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--backend nccl|gloo]
N > 1: one rank per GPU over RCCL.  Started under torch.distributed.run (WORLD_SIZE set) this process
is one rank; started bare, it launches `python -m torch.distributed.run --nproc-per-node N ... bench.py`
as a child BEFORE touching the GPU, relays rank 0's JSON line and exits with the child's code.
weak scaling: 1024 trajectories per GPU; strong: the 1024-trajectory batch split N ways.  The only
exchange is one all-reduce of the packed critic [loss | grads | count] buffer through
gan_mpc_amd/parallel.py -- the same two calls the trainers use.  Two streams (small-state workloads): the critic
step runs on a second stream gated behind the Jacobian chain, i.e. beside the Riccati sweep; its all-reduce is
started on that stream as soon as the last gradient kernel is enqueued and finished before the optimiser step,
so what hides it is whatever the Riccati sweep still has to run when the critic's kernels are through (both
chains take about 0.35 ms side by side: on one GPU nothing is exchanged; the RCCL path with N > 1 has not run on
hardware yet).
Rank 0 prints ONE JSON line; `ms_per_step` / `value` are the contract's single timed window of K steps,
`windows` gives median / min / max over that window and four more of the same length.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# headline workload: BASELINE.json configs[2] shapes (HalfCheetah-sized, H=50, batch 1024, critic
# LSTM(64) + 3x256 head); the metric names the critic step, so the critic config is the one timed.
WORKLOAD = dict(name="C3 synthetic HalfCheetah-shape n=17 m=6 H=50 B=1024/GPU, dynamics MLP 4x200, "
                     "cost MLP 3x128->10, critic LSTM(64)+3x256 head",
                n=17, m=6, T=50, B=1024, dyn_hidden=(200, 200, 200), cost_hidden=(128, 128),
                cost_fout=10, F=64, head_hidden=(256, 256, 256))

# the other BASELINE.json configs, selectable for measurement (not the headline line): per-GPU shards
# of C4 (Humanoid, batch 4096 over 8 GPUs) and C5 (synthetic n=1024, batch 8192 over 8 GPUs)
WORKLOADS = {
    "c3": WORKLOAD,
    "c4": dict(name="C4 synthetic Humanoid-shape n=376 m=17 H=50 B=512/GPU (4096 over 8), dynamics MLP "
                    "4x200, cost MLP 3x128->10, critic LSTM(64)+3x256 head",
               n=376, m=17, T=50, B=512, dyn_hidden=(200, 200, 200), cost_hidden=(128, 128),
               cost_fout=10, F=64, head_hidden=(256, 256, 256)),
    "c5": dict(name="C5 synthetic n=1024 m=64 H=100 B=1024/GPU (8192 over 8), dynamics MLP 4x200, "
                    "cost MLP 3x128->10, critic LSTM(64)+3x256 head",
               n=1024, m=64, T=100, B=1024, dyn_hidden=(200, 200, 200), cost_hidden=(128, 128),
               cost_fout=10, F=64, head_hidden=(256, 256, 256)),
}

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_HBM_GBS = 8000.0


def linearize_flops_per_sample(n, m, dyn_dims):
    """Algorithmic flops of one dynamics Jacobian (reverse chain, n rows): 2*n*(sum of hidden-hidden
    products + first-layer product); SURVEY.md 8d 'Jacobian chain'."""
    hh = sum(a * b for a, b in zip(dyn_dims[1:-2], dyn_dims[2:-1]))
    return 2.0 * n * (hh + dyn_dims[1] * (n + m))


def riccati_flops_per_sample(n, m):
    """Dense products of one lqr_step (trajax tvlqr): P A, P B, A^T(PA), B^T(PA), B^T(PB) and the
    cross terms K^T W; the m x m factorisation and the solves are lower order."""
    return 2.0 * (2 * n ** 3 + 2 * n * n * m + n * m * m + n * n * m)


def lowrank_flops_per_sample(n, m, dyn_dims):
    """Flops the large-state pass EXECUTES per trajectory-step when it runs on the low-rank form of the
    Jacobians (last hidden width h < n / 2, gmpc_large.hip): the factor V^T, Y = W_L P, S = W_L Y^T,
    [Z | S Vu^T / 2] = S V^T / 2 + [Y | 0], [H|G] = Vu (2 [Z|.] - [Y|0]), T1 = P + Vx Z + Z^T Vx^T + K^T W on its
    upper blocks (128-wide: 9/16 of the square at n = 1024)."""
    h = dyn_dims[-2]
    nm = n + m
    hh = sum(a * b for a, b in zip(dyn_dims[1:-2], dyn_dims[2:-1]))
    factors = 2.0 * h * (hh - (dyn_dims[-3] * h if len(dyn_dims) > 3 else 0)) + 2.0 * h * dyn_dims[1] * nm
    nb = -(-n // 128)
    upper = (nb * (nb + 1) / 2) / (nb * nb)
    return (factors + 2.0 * (h * n * n + h * h * n + h * h * nm + m * h * nm + m * h * n) +
            2.0 * upper * n * n * (2 * h + 2 * m))


def step_bytes_per_traj(n, m, T):
    """SURVEY.md 8d algorithmic HBM bytes per trajectory-step of the fused rollout+backward."""
    rd = n + T * m + (T + 1) * n
    wr = (T + 1) * n + (T + 1) + T * (m * n + m) + T * m + (T + 1) * n
    return 4.0 * (rd + wr)


def usable_cpus():
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:                                                                        # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(args, w):
    """The CPU restatement of the same step timed on this box's host cores ('port': the JAX reference cannot
    run here).  Preferred: oracle/gan_mpc_step.c, plain C with OpenMP over the trajectories, on the FULL batch
    (>= 3 warm-ups, median of >= 10 steps); without that shared object, the NumPy oracle
    on a bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import gan_mpc_oracle as orc
    n, m, T, F = w["n"], w["m"], w["T"], w["F"]
    try:
        # one OpenMP thread per CPU this process may USE: the affinity mask, capped by the cgroup CPU quota
        # (a GPU box hands a job a share of its host cores; more threads than that only oversubscribe)
        os.environ.setdefault("OMP_NUM_THREADS", str(usable_cpus()))
        import gan_mpc_step_c as oc
        oc.load()
    except Exception:
        oc = None
    Bs = (args.cpu_batch or w["B"]) if oc is not None else args.cpu_sample
    pb = orc.make_problem(n, m, T, Bs, seed=0, dyn_hidden=w["dyn_hidden"], cost_hidden=w["cost_hidden"],
                          cost_fout=w["cost_fout"], lstm_features=F, head_hidden=w["head_hidden"])
    label = np.concatenate([np.ones(Bs), -np.ones(Bs)]).astype(np.float32)
    head_dims = [F, *w["head_hidden"], 1]
    if oc is not None:
        flat = oc.critic_flat(pb["critic"])
        state = dict(p=flat.copy(), m=np.zeros_like(flat), v=np.zeros_like(flat), k=0)

        def one_step():
            out = oc.trajectories(pb["dyn"], pb["cmlp"], pb["mpc_w"], pb["goal"], pb["x0"], pb["U"])
            xseq = np.concatenate([pb["true_seq"], out["X"]], 0)
            _, g = oc.critic_loss_grad(state["p"], n, F, head_dims, xseq, label)
            state["k"] += 1
            oc.adam_clip(state["p"], g, state["m"], state["v"], 1.0 / (2 * Bs), state["k"], 1e-5)

        cores = oc.threads()
        what = (f"oracle/gan_mpc_step.c (plain C, OpenMP over trajectories, {cores} threads) on the full batch of "
                f"{Bs} trajectories")
    else:
        try:
            from threadpoolctl import threadpool_limits
        except Exception:  # pragma: no cover
            threadpool_limits = None
        cores = os.cpu_count() or 1
        cnt = sum(v.size for v in (pb["critic"]["Wx"], pb["critic"]["Wh"], pb["critic"]["b"]))
        cnt += sum(W.size + b.size for W, b in pb["critic"]["head"])
        st = [np.zeros(cnt, np.float32), np.zeros(cnt, np.float32), np.zeros(cnt, np.float32), 0]

        def one_step():
            X = orc.rollout(pb["dyn"], pb["U"], pb["x0"])
            orc.evaluate(pb["cmlp"], pb["mpc_w"], pb["goal"], X, pb["U"])
            lqr = orc.get_lqr_params(pb["dyn"], pb["cmlp"], pb["mpc_w"], pb["goal"], X, pb["U"])
            with np.errstate(all="ignore"):
                orc.tvlqr(*lqr)
            orc.adjoint(lqr[5], lqr[6], lqr[1], lqr[3])
            xseq = np.concatenate([pb["true_seq"], X], 0)
            _, g = orc.critic_loss_and_grad(pb["critic"], xseq, label)
            flat = np.concatenate([g["Wx"].ravel(), g["Wh"].ravel(), g["b"].ravel()]
                                  + [t.ravel() for Wb in g["head"] for t in Wb])
            st[3] += 1
            st[0], st[1], st[2] = orc.adam_clip_step(st[0], flat, st[1], st[2], st[3], 1e-5)

        what = (f"the NumPy oracle on {Bs} trajectories (oracle/libgan_mpc_step.so not built; BLAS threads <= "
                f"{cores}, the per-trajectory matrices are small, so most cores idle)")
    times = []
    with np.errstate(all="ignore"):
        for _ in range(3):
            one_step()
        t_all = time.perf_counter()
        while len(times) < 10 or (time.perf_counter() - t_all < args.cpu_seconds and len(times) < 200):
            t0 = time.perf_counter()
            one_step()
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": Bs / med, "unit": "trajectories/sec", "cores": cores, "kind": "port",
            "sample": f"median of {len(times)} steps (after 3 warm-ups) of {what}; same shapes, same step"}


def backward_roofline(n, m, T, B, dyn_dims, ms):
    """Roofline entry of one backward pass (`ms` per launch) of a large-state workload: the algorithmic count is
    SURVEY 8d's dense one (Jacobian chain + the n^3 Riccati products); when the pass runs on the low-rank form of
    the Jacobians it EXECUTES fewer flops, and `frac` is then the executed rate over the peak -- the dense-count
    figure (which can exceed 1) stays under its own name."""
    flops = (linearize_flops_per_sample(n, m, dyn_dims) + riccati_flops_per_sample(n, m)) * B * T
    ach = flops / (ms * 1e-3) / 1e12
    out = {"bound": "mfma", "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "avg_launch_ms": round(ms, 4),
           "algorithmic_gflop_per_launch": round(flops / 1e9, 2), "executed": None}
    if 2 * dyn_dims[-2] < n and not os.environ.get("GMPC_BIG_DENSE"):
        ex = lowrank_flops_per_sample(n, m, dyn_dims) * B * T
        ex_t = ex / (ms * 1e-3) / 1e12
        out["executed"] = {"gflop_per_launch": round(ex / 1e9, 2), "TFLOPs": round(ex_t, 3),
                           "frac_of_peak": round(ex_t / PEAK_FP32_TFLOPS, 4)}
        out["achieved"] = round(ex_t, 3)
        out["frac"] = round(ex_t / PEAK_FP32_TFLOPS, 4)
        out["frac_basis"] = "flops executed by the low-rank form"
        out["dense_count_TFLOPs"] = round(ach, 3)
        out["dense_count_frac"] = round(ach / PEAK_FP32_TFLOPS, 4)
    else:
        out["achieved"] = round(ach, 3)
        out["frac"] = round(ach / PEAK_FP32_TFLOPS, 4)
        out["frac_basis"] = "dense algorithmic count (the pass executes it)"
    return out


def large_state_entry(key, steps, warmup, dev_index):
    """A short pass of one of the large-state configurations (per-GPU shard of BASELINE.json configs 4 / 5) for the
    `large_state` key of the headline line: the same step (rollout + costs, critic step on 2B sequences, backward
    pass, clip+Adam), one stream, inputs resident in HBM, `steps` timed steps after `warmup`."""
    import numpy as np
    import torch
    import ctypes as C
    from gan_mpc_amd import _lib
    from gan_mpc_amd import params as P
    from gan_mpc_amd import synthetic
    from gan_mpc_amd.engine import Engine
    w = WORKLOADS[key]
    n, m, T, F, B = w["n"], w["m"], w["T"], w["F"], w["B"]
    mk = dict(dyn_hidden=w["dyn_hidden"], cost_hidden=w["cost_hidden"], cost_fout=w["cost_fout"],
              lstm_features=F, head_hidden=w["head_hidden"])
    pb = synthetic.make_problem(n, m, T, B, seed=1000, **mk)
    wts = synthetic.make_problem(n, m, T, 1, seed=0, **mk)
    dyn_dims = [n + m, *w["dyn_hidden"], n]
    cost_dims = [n, *w["cost_hidden"], w["cost_fout"]]
    head_dims = [F, *w["head_hidden"], 1]
    eng = Engine(n, m, T, dyn_dims, cost_dims, max_batch=B, lstm_features=F, head_dims=head_dims, device=dev_index)
    try:
        d = eng.to_dev
        eng.set_params(d(wts["mpc_w"]), d(P.pack_mlp(P.layers_to_tree(wts["dyn"]))),
                       d(P.pack_mlp(P.layers_to_tree(wts["cmlp"]))))
        critic = d(P.pack_critic(P.critic_dict_to_tree(wts["critic"])))
        adam_m, adam_v = torch.zeros_like(critic), torch.zeros_like(critic)
        x0, U, goal = d(pb["x0"]), d(pb["U"]), d(pb["goal"])
        xseq = eng.new(2 * B, T + 1, n)
        xseq[:B].copy_(d(pb["true_seq"]))
        del pb
        label = d(np.concatenate([np.ones(B), -np.ones(B)]).astype(np.float32))
        X = xseq[B:]
        costs = eng.new(B, T + 1)
        bw = dict(K=eng.new(B, T, m, n), k=eng.new(B, T, m), grad=eng.new(B, T, m), adjoints=eng.new(B, T + 1, n),
                  AB=None)
        packed = torch.zeros(1 + eng.critic_count, dtype=torch.float32, device=eng.device)

        def step(k):
            eng.rollout_cost(x0, U, goal, X=X, costs=costs)
            _lib.check(eng.lib.gmpc_critic_loss_grad(
                eng.ctx, 2 * B, C.c_void_p(xseq.data_ptr()), C.c_void_p(label.data_ptr()),
                C.c_void_p(critic.data_ptr()), C.c_void_p(packed[:1].data_ptr()),
                C.c_void_p(packed[1:].data_ptr()), eng._stream()))
            eng.lqr_backward(X, U, goal, after_rollout=True, out=bw)
            eng.adam_clip_step(critic, packed[1:], adam_m, adam_v, k + 1, lr=1e-5, grad_scale=1.0 / (2 * B))

        for k in range(warmup):
            step(k)
        torch.cuda.synchronize()
        eng.profile_enable(True)
        eng.profile_read()
        t0 = time.perf_counter()
        for k in range(steps):
            step(warmup + k)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof = eng.profile_read()
        eng.profile_enable(False)
        ms_bw = prof["riccati"][0] / max(1, prof["riccati"][1])
        return {"workload": w["name"], "steps": steps, "warmup": warmup, "batch_per_gpu": B,
                "ms_per_step": round(dt / steps * 1e3, 3), "trajectories_per_sec": round(B * steps / dt, 1),
                "kernel_ms_per_step": {kk: round(v[0] / steps, 4) for kk, v in prof.items() if v[1]},
                "roofline": dict(kernel="large-state backward pass (Jacobian chain + batched GEMMs + k_big_step), "
                                        "HIP events around the whole pass",
                                 **backward_roofline(n, m, T, B, dyn_dims, ms_bw))}
    finally:
        eng.close()
        torch.cuda.empty_cache()


def launch_ranks(args):
    """`python bench.py --gpus N` from a bare shell: start N ranks with torch.distributed.run as a CHILD
    process (this parent never initialises the GPU), relay rank 0's JSON line, exit with the child's code."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith('{"metric"')]
    if proc.returncode != 0 or not lines:
        sys.stderr.write(proc.stdout)
        raise SystemExit(proc.returncode or 1)
    print(lines[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed steps before the timed window (default = the driver's contract command, --steps 20 "
                         "--warmup 5; the first ~5 steps of a process run 2-3 %% slow, so the contract's window is the "
                         "slowest of the five reported under `windows`)")
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS),
                    help="c3 = the headline configuration; c4 / c5 = per-GPU shards of the large-state configs")
    ap.add_argument("--batch", type=int, default=0, help="trajectories per GPU (0: the workload's)")
    ap.add_argument("--cpu-sample", type=int, default=32, help="trajectories of the NumPy fallback baseline")
    ap.add_argument("--cpu-batch", type=int, default=0, help="trajectories of the C baseline (0: the workload's batch)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-step", default="",
                    help="rank 0 saves the last step's all-reduced [mean loss | mean critic gradient] and the critic "
                         "parameters to this .npz (tests/test_gpu_multirank.py: N ranks against one)")
    ap.add_argument("--windows", type=int, default=5,
                    help="timed windows of --steps steps each (the first one is the contract's: value / ms_per_step)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="everything on one stream (default: the critic step runs on a second stream beside the "
                         "Riccati sweep, gated behind the Jacobian chain by gmpc_set_linearize_event)")
    ap.add_argument("--overlap", action="store_true", help=argparse.SUPPRESS)     # (the default since round 2)
    ap.add_argument("--no-pipeline", action="store_true",
                    help="the optimiser step of step k is joined back into the main stream before step k + 1 starts "
                         "(the round-3 ordering).  Default: the critic chain, its all-reduce and clip+Adam stay on the "
                         "second stream -- step k + 1's rollout starts behind step k's Riccati sweep, the critic "
                         "parameters are next read by step k + 1's critic chain on that same stream (two rollout "
                         "buffers, so that step k + 1's rollout does not overwrite sequences step k's critic still reads)")
    ap.add_argument("--secondary-maxiter", type=int, default=10,
                    help="maxiter of the secondary full-bilevel measurement (0: skip)")
    ap.add_argument("--solve-maxiter", type=int, default=100,
                    help="maxiter of the complete-solve measurement inside `secondary` (0: skip)")
    ap.add_argument("--dump-windows", action="store_true", help="list every window's ms_per_step under `windows.all`")
    ap.add_argument("--no-large-state", action="store_true",
                    help="skip the short C4 (5 steps) and C5 (1 step) passes attached to the headline line as `large_state`")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend: nccl (= RCCL over xGMI, the real thing) or gloo "
                         "(rehearsal of the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: the workload's batch per GPU; strong: that batch split over the ranks")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    import numpy as np
    import torch
    import torch.distributed as dist

    from gan_mpc_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: gan_mpc_amd has no CPU fallback")
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world_env} ranks")
    # device + process group before the first kernel launch (the same call the runners make)
    rank, world, dev_index = parallel.init_from_env(args.backend)

    from gan_mpc_amd import params as P
    from gan_mpc_amd import synthetic
    from gan_mpc_amd.engine import Engine

    w = dict(WORKLOADS[args.workload])
    Bw = args.batch or w["B"]
    n, m, T, F = w["n"], w["m"], w["T"], w["F"]
    mk = dict(dyn_hidden=w["dyn_hidden"], cost_hidden=w["cost_hidden"], cost_fout=w["cost_fout"],
              lstm_features=F, head_hidden=w["head_hidden"])
    if args.scaling == "strong":
        # ONE global batch (the single-GPU problem), this rank owns a contiguous shard of it
        lo, hi = parallel.shard_range(Bw, rank, world)
        pb = synthetic.make_problem(n, m, T, Bw, seed=1000, **mk)
        for key in ("x0", "U", "goal", "true_seq"):
            pb[key] = pb[key][lo:hi]
        B, global_B = hi - lo, Bw
        if B < 1:
            raise SystemExit(f"strong scaling: batch {Bw} leaves rank {rank} of {world} without work")
    else:
        pb = synthetic.make_problem(n, m, T, Bw, seed=1000 + rank, **mk)
        B, global_B = Bw, Bw * world
    wts = synthetic.make_problem(n, m, T, 1, seed=0, **mk)
    dyn_dims = [n + m, *w["dyn_hidden"], n]
    cost_dims = [n, *w["cost_hidden"], w["cost_fout"]]
    head_dims = [F, *w["head_hidden"], 1]
    eng = Engine(n, m, T, dyn_dims, cost_dims, max_batch=B, lstm_features=F, head_dims=head_dims,
                 device=dev_index)
    d = eng.to_dev
    # identical (seed 0) parameters on every rank, rank-specific trajectories
    eng.set_params(d(wts["mpc_w"]), d(P.pack_mlp(P.layers_to_tree(wts["dyn"]))),
                   d(P.pack_mlp(P.layers_to_tree(wts["cmlp"]))))
    critic = d(P.pack_critic(P.critic_dict_to_tree(wts["critic"])))
    adam_m, adam_v = torch.zeros_like(critic), torch.zeros_like(critic)
    x0, U, goal = d(pb["x0"]), d(pb["U"]), d(pb["goal"])
    xseq = eng.new(2 * B, T + 1, n)
    xseq[:B].copy_(d(pb["true_seq"]))
    label = d(np.concatenate([np.ones(B), -np.ones(B)]).astype(np.float32))
    X = xseq[B:]                       # the rollout writes straight into the critic batch
    costs = eng.new(B, T + 1)
    bw = dict(K=eng.new(B, T, m, n), k=eng.new(B, T, m), grad=eng.new(B, T, m),
              adjoints=eng.new(B, T + 1, n), AB=None)
    # [loss_sum | grad_sum]: the critic kernels write straight into the exchange buffer; the batch is
    # static, so the global sample count is a constant folded into the Adam step's gradient scale
    packed = parallel.new_packed(1 + eng.critic_count, eng.device)
    loss_view, grad_view = packed[:1], packed[1:]
    inv_count = 1.0 / (2 * global_B)
    import ctypes as C
    from gan_mpc_amd import _lib

    # Two streams for the small-state workloads: the Riccati sweep there is two wavefronts per trajectory
    # (k_riccati_w2: half of the wave slots, a quarter of the registers), the critic's kernels fill the rest; the Jacobian chain
    # (matrix-core-bound, all registers) always runs alone.  The large-state pipeline keeps one stream.
    side = torch.cuda.Stream() if (not args.no_overlap and n <= 64) else None
    lin_ev = None
    if side is not None:
        lin_ev = torch.cuda.Event()
        lin_ev.record()                      # creates the HIP event behind the handle
        eng.set_linearize_event(lin_ev)

    # Deferred optimiser step (default with two streams): nothing on the main stream reads the critic parameters --
    # the next reader is the NEXT step's critic chain, on the second stream -- so the critic chain, the exchange and
    # clip+Adam of step k are never joined back: step k + 1's rollout is enqueued right behind step k's Riccati sweep
    # and the exchange has the whole next rollout to hide behind.  The rollout of step k + 2 reuses step k's sequence
    # buffer and waits for the event recorded behind step k's optimiser step.
    pipe = side is not None and not args.no_pipeline
    xbufs = [xseq]
    if pipe:
        xseq_b = eng.new(2 * B, T + 1, n)
        xseq_b.copy_(xseq)
        xbufs.append(xseq_b)
    crit_done = [None] * len(xbufs)

    def critic_grads(xs):
        _lib.check(eng.lib.gmpc_critic_loss_grad(
            eng.ctx, 2 * B, C.c_void_p(xs.data_ptr()), C.c_void_p(label.data_ptr()),
            C.c_void_p(critic.data_ptr()), C.c_void_p(loss_view.data_ptr()),
            C.c_void_p(grad_view.data_ptr()), eng._stream()))
        return parallel.allreduce_start(packed)

    def step(k):
        # rollout -> backward pass (terminal, Jacobian chain, Riccati sweep) on the main stream; the critic
        # gradients, their all-reduce and the optimiser step on a second stream that waits for the Jacobian chain
        # and runs beside the Riccati sweep (--no-overlap: one stream, critic before the backward pass)
        j = k % len(xbufs)
        xs = xbufs[j]
        Xk = xs[B:]
        if crit_done[j] is not None:
            torch.cuda.current_stream().wait_event(crit_done[j])
        eng.rollout_cost(x0, U, goal, X=Xk, costs=costs)
        if side is None:
            work = critic_grads(xs)
            eng.lqr_backward(Xk, U, goal, after_rollout=True, out=bw)
        else:
            eng.lqr_backward(Xk, U, goal, after_rollout=True, out=bw)     # records lin_ev after the chain
            side.wait_event(lin_ev)
            with torch.cuda.stream(side):
                work = critic_grads(xs)
                if pipe:
                    parallel.allreduce_finish(packed, work, counted=False)
                    eng.adam_clip_step(critic, grad_view, adam_m, adam_v, k + 1, lr=1e-5, grad_scale=inv_count)
                    if crit_done[j] is None:
                        crit_done[j] = torch.cuda.Event()
                    crit_done[j].record(side)
            if pipe:
                return
            torch.cuda.current_stream().wait_stream(side)
        parallel.allreduce_finish(packed, work, counted=False)
        eng.adam_clip_step(critic, grad_view, adam_m, adam_v, k + 1, lr=1e-5, grad_scale=inv_count)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # diagnostic only ("the first steps of a process"): GMPC_BENCH_PREHEAT_MS=<ms> keeps the chip busy
    # with fp32 matrix products for that long before the warm-up steps, GMPC_BENCH_PRESTEPS=<n> runs n extra steps.
    # Neither is part of the contract's measurement and neither is set by default.
    if os.environ.get("GMPC_BENCH_PREHEAT_MS"):
        pa = torch.randn(4096, 4096, device="cuda")
        t_end = time.perf_counter() + float(os.environ["GMPC_BENCH_PREHEAT_MS"]) * 1e-3
        while time.perf_counter() < t_end:
            pa = torch.mm(pa, pa) * 1e-4
            torch.cuda.synchronize()
        del pa
    for k in range(int(os.environ.get("GMPC_BENCH_PRESTEPS", "0"))):
        step(k)
    if os.environ.get("GMPC_BENCH_HOSTTIMES"):       # diagnostic: host enqueue time and completion time of the first steps
        torch.cuda.synchronize()
        t_h, evs = [], []
        tb = time.perf_counter()
        for k in range(int(os.environ["GMPC_BENCH_HOSTTIMES"])):
            th0 = time.perf_counter()
            step(k)
            t_h.append((time.perf_counter() - th0) * 1e3)
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            evs.append(e)
        torch.cuda.synchronize()
        sys.stderr.write("host ms per step: " + " ".join(f"{x:.2f}" for x in t_h) + "\n")
        sys.stderr.write("gpu ms between step ends (stream 1): " +
                         " ".join(f"{evs[i - 1].elapsed_time(evs[i]):.3f}" for i in range(1, len(evs))) + "\n")
    for k in range(args.warmup):
        step(k)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    value = global_B * args.steps / dt
    # spread: four more windows of the same length (every rank runs them: they contain the collective)
    win = [ms_per_step]
    for wi in range(args.windows - 1):
        sync()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(args.warmup + args.steps * (wi + 1) + k)
        sync()
        dw = time.perf_counter() - t1
        if world > 1:
            tw = torch.tensor([dw], device="cuda")
            dist.all_reduce(tw, op=dist.ReduceOp.MAX)
            dw = float(tw.item())
        win.append(dw / args.steps * 1e3)

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream
    roof = None
    prof = {}
    nprof = min(args.steps, 10)
    if rank == 0:
        eng.profile_enable(True)
    for k in range(nprof):       # every rank runs these steps: they contain the collective
        step(args.warmup + args.steps + k)
    torch.cuda.synchronize()
    if rank == 0:
        prof = eng.profile_read()
        eng.profile_enable(False)
        dom = max(prof, key=lambda kk: prof[kk][0])
        if n <= 64:
            kname = eng.profile_kernel_name("linearize") or "k_linearize"
            ms, cnt = prof["linearize"]
            flops = linearize_flops_per_sample(n, m, dyn_dims) * B * T
        else:
            # step-major large-state pass: Jacobian chain + batched GEMMs + gain kernels, timed as one
            kname = "large-state backward (k_linearize_mfma + k_bgemm_tn_lds + k_big_step)"
            ms, cnt = prof["riccati"]
            flops = (linearize_flops_per_sample(n, m, dyn_dims) + riccati_flops_per_sample(n, m)) * B * T
        ach = flops / (ms / cnt * 1e-3) / 1e12
        if n <= 64:
            roof = {"kernel": kname, "bound": "mfma", "achieved": round(ach, 3),
                    "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
                    "traffic": None,
                    "avg_launch_ms": round(ms / cnt, 4),
                    "algorithmic_gflop_per_launch": round(flops / 1e9, 2),
                    "executed": None}
        else:
            # `frac` = the rate of the flops the pass EXECUTES over the peak (the low-rank form issues fewer than
            # the dense algorithmic count of SURVEY 8d; that figure is kept as dense_count_frac)
            roof = {"kernel": kname, "traffic": None, **backward_roofline(n, m, T, B, dyn_dims, ms / cnt)}
        roof.update({
                "dominant_by_time": dom,
                "hbm_view": {"algorithmic_bytes_per_step": step_bytes_per_traj(n, m, T) * B,
                             "achieved_GBs_whole_step": round(
                                 step_bytes_per_traj(n, m, T) * B / (ms_per_step * 1e-3) / 1e9, 2),
                             "peak_GBs": PEAK_HBM_GBS},
                "kernel_ms_per_step": {kk: round(v[0] / nprof, 4) for kk, v in prof.items() if v[1]},
                "streams": 1 if side is None else 2,
                "kernel_ms_note": None if side is None else
                "terminal / riccati (stream 1) and the critic kernels (stream 2) run side by side: their times "
                "overlap and include the sharing; rollout and linearize run alone"})
        tr = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tr) and args.workload == "c3":
            try:
                tj = json.load(open(tr))
                roof["traffic"] = tj.get("k_linearize_hbm_bytes_per_launch")
                # PMC counters need their own rocprofv3 passes: this figure is NOT measured in this run
                roof["traffic_source"] = ("profiles/traffic_latest.json: separate rocprofv3 --pmc passes of "
                                          "this command (" + str(tj.get("source", "see file")) + "), not this run")
            except Exception:
                pass

    # ---- secondary: full bilevel_optimization = iLQR solve at fixed maxiter + bilevel
    # gradient (L2 upper loss), with the iteration histogram; and the product's real hot loop, the
    # complete solve at the reference's maxiter = 100, with the roofline of its dominant kernel (the
    # line-search rollouts).  Not part of `value`.
    secondary = None
    if rank == 0 and world == 1 and args.secondary_maxiter > 0 and n <= 64:
        kw = {"maxiter": args.secondary_maxiter}
        desired = xseq[:B]
        sol = eng.ilqr_solve(x0, U, goal, kw)           # warm-up
        eng.bilevel_grad(B, 0, desired=desired)
        torch.cuda.synchronize()
        reps = 3
        t1 = time.perf_counter()
        for _ in range(reps):
            sol = eng.ilqr_solve(x0, U, goal, kw)
            eng.bilevel_grad(B, 0, desired=desired)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t1) / reps
        its = sol["iterations"].cpu().numpy()
        hist = np.bincount(its, minlength=args.secondary_maxiter + 1)
        secondary = {"what": f"gmpc_ilqr_solve(maxiter={args.secondary_maxiter}) + gmpc_bilevel_grad(L2) "
                             f"on {B} trajectories",
                     "trajectories_per_sec": round(B / dt2, 1), "ms": round(dt2 * 1e3, 3),
                     "iteration_histogram": {str(i): int(c) for i, c in enumerate(hist) if c}}
        # a8-a11 on their own (the structured Hessian solve + loss adjoint + cost_vjp + batch sums), at the solution the
        # ctx holds; kernel_ms = the Hessian-solve kernel alone (HIP events around its launch)
        eng.profile_enable(True)
        eng.profile_read()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.bilevel_grad(B, 0, desired=desired)
        torch.cuda.synchronize()
        dtb = (time.perf_counter() - t1) / 5
        profb = eng.profile_read()
        eng.profile_enable(False)
        secondary["bilevel"] = {"what": "gmpc_bilevel_grad(L2) alone: loss adjoint + structured Hessian solve + cost_vjp "
                                        "+ weight-gradient sums",
                                "ms": round(dtb * 1e3, 4),
                                "kernel_ms": round(profb["riccati"][0] / max(1, profb["riccati"][1]), 4),
                                "kernel": "Hessian solve (Riccati sweep, mode 1)"}
        if args.solve_maxiter > 0:
            kw = {"maxiter": args.solve_maxiter}
            eng.ilqr_solve(x0, U, goal, kw)             # warm-up
            eng.profile_enable(True)
            eng.profile_read()
            t1 = time.perf_counter()
            sol = eng.ilqr_solve(x0, U, goal, kw)       # synchronises before returning
            dt3 = time.perf_counter() - t1
            prof3 = eng.profile_read()
            eng.profile_enable(False)
            its = sol["iterations"].cpu().numpy()
            hist = np.bincount(its, minlength=1)
            ncand = eng.linesearch_candidates()
            ls_ms, ls_cnt = prof3["linesearch"]
            # one candidate = one rollout through the dynamics MLP (T * 2 * sum d_i d_{i+1}) plus the
            # gain products K_t (x - xbar) of the DDP forward pass
            roll_flops = T * (2.0 * sum(a * b for a, b in zip(dyn_dims[:-1], dyn_dims[1:])) + 2.0 * n * m)
            ach = ncand * roll_flops / (ls_ms * 1e-3) / 1e12 if ls_ms > 0 else 0.0
            secondary["solve"] = {
                "what": f"gmpc_ilqr_solve(maxiter={args.solve_maxiter}) on {B} trajectories (trajax defaults)",
                "ms": round(dt3 * 1e3, 2), "trajectories_per_sec": round(B / dt3, 1),
                "iteration_histogram": {str(i): int(c) for i, c in enumerate(hist) if c},
                "iterations_run": int(its.max()),
                "ms_per_iteration": {kk: round(v[0] / max(1, int(its.max())), 4) for kk, v in prof3.items() if v[1]},
                "linesearch_candidates": ncand,
                "linesearch_stats": eng.linesearch_stats(),
                "roofline": {"kernel": "k_ls32 (work lists above 4096 candidates) + k_ls16 (1537 and more) + k_traj_rw<true> (shorter ones); "
                                       "place/decide kernels included in the time",
                             "bound": "mfma", "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
                             "algorithmic_mflop_per_candidate": round(roll_flops / 1e6, 3),
                             "linesearch_ms_total": round(ls_ms, 2), "linesearch_calls": ls_cnt}}

    # the same solve on "trained-like" dynamics (residual head x 0.1: next_x = x + small, the state stays O(1)):
    # trajectories stop at different iterations, so the masked iterations of the loop are timed as well
    if secondary is not None and args.solve_maxiter > 0:
        tdyn = [(W.copy(), b.copy()) for W, b in wts["dyn"]]
        tdyn[-1] = ((tdyn[-1][0] * 0.1).astype(np.float32), tdyn[-1][1])
        eng.set_params(d(wts["mpc_w"]), d(P.pack_mlp(P.layers_to_tree(tdyn))),
                       d(P.pack_mlp(P.layers_to_tree(wts["cmlp"]))))
        kw = {"maxiter": args.solve_maxiter}
        eng.ilqr_solve(x0, U, goal, kw)                 # warm-up
        eng.profile_enable(True)
        eng.profile_read()
        t1 = time.perf_counter()
        sol = eng.ilqr_solve(x0, U, goal, kw)
        dt4 = time.perf_counter() - t1
        prof4 = eng.profile_read()
        eng.profile_enable(False)
        its = sol["iterations"].cpu().numpy()
        edges = [0, 1, 2, 5, 10, 20, 50, 100, 1 << 30]
        secondary["solve_trained"] = {
            "what": f"gmpc_ilqr_solve(maxiter={args.solve_maxiter}) on {B} trajectories, dynamics output layer x 0.1 "
                    "(trained-like); trajectories stop at different iterations",
            "ms": round(dt4 * 1e3, 2), "trajectories_per_sec": round(B / dt4, 1),
            "iterations": {"min": int(its.min()), "median": float(np.median(its)), "mean": round(float(its.mean()), 2),
                           "max": int(its.max())},
            "iteration_histogram_bins": {f"{lo}-{min(hi, args.solve_maxiter + 1) - 1}": int(((its >= lo) & (its < hi)).sum())
                                         for lo, hi in zip(edges[:-1], edges[1:]) if ((its >= lo) & (its < hi)).any()},
            "ms_total": {kk: round(v[0], 3) for kk, v in prof4.items() if v[1]},
            "linesearch_candidates": eng.linesearch_candidates(),
            "linesearch_stats": eng.linesearch_stats()}

    # ---- the large-state configurations, briefly, so that the driver's record carries them (BASELINE.json
    # configs 4 and 5 at their per-GPU shards): not part of `value`
    large = None
    if rank == 0 and world == 1 and args.workload == "c3" and not args.no_large_state:
        eng.close()
        del eng
        torch.cuda.empty_cache()
        large = {}
        for key, st, wu in (("c4", 5, 2), ("c5", 1, 1)):
            try:
                large[key] = large_state_entry(key, st, wu, dev_index)
            except Exception as exc:       # the headline line must not be lost to a failure here
                large[key] = {"error": repr(exc)[:300]}

    # replicas: every rank applied the same all-reduced gradient with the same fused clip+Adam -- the parameter
    # vectors must be bitwise identical (checked on every multi-rank run, it costs one small all-gather)
    if world > 1:
        mine = critic.detach().clone()
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        if not all(torch.equal(allp[0], q) for q in allp[1:]):
            raise SystemExit("bench.py: the critic parameter replicas diverged across ranks")
    if args.dump_step and rank == 0:
        torch.cuda.synchronize()
        np.savez(args.dump_step, mean=(packed * inv_count).cpu().numpy(), critic=critic.cpu().numpy(),
                 world=world, global_batch=global_B)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args, w)

    if rank == 0:
        out = {
            "metric": "trajectories/sec (rollout+backward+critic step) at batch=1024, H=50"
                      if args.workload == "c3" else
                      f"trajectories/sec (rollout+backward+critic step), workload {args.workload}",
            "value": round(value, 1), "unit": "trajectories/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["name"], "batch_per_gpu": B, "global_batch": global_B,
                       "horizon": T, "state_dim": n, "act_dim": m,
                       "parallelism": f"trajectory-sharded x{world} ({args.scaling} scaling, backend "
                                      f"{args.backend if world > 1 else 'none'}), 1 all-reduce of critic grads/step"},
            "windows": {"n": len(win), "steps_each": args.steps, "ms_per_step_median": round(float(np.median(win)), 4),
                        "ms_per_step_min": round(min(win), 4), "ms_per_step_max": round(max(win), 4),
                        **({"all": [round(x, 4) for x in win]} if args.dump_windows else {})},
            "step_ordering": ("one stream" if side is None else
                              "two streams, optimiser step deferred: critic chain + all-reduce + clip/Adam of step k stay "
                              "on stream 2 beside the Riccati sweep of step k and the rollout of step k + 1"
                              if pipe else "two streams, optimiser step joined into stream 1 at the end of every step"),
            "roofline": roof, "cpu_baseline": cpu, "secondary": secondary, "large_state": large,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
