"""Synthetic workloads (SURVEY 8d "Synthetic inputs"): seeded parameters and trajectories of a given
shape for benchmarks and smoke runs -- no dataset or checkpoint ships with either repository.

Weights: LeCun-normal kernels N(0, 1/fan_in) and zero biases (the flax Dense defaults) for the dynamics
MLP, the cost MLP and the critic; mpc_weights (-2, 3, -3) as in the reference's yaml; x0 ~ N(0, 1);
initial controls tanh(N(0, 1)) (expert actions are tanh-bounded); goals and "true" state sequences
~ N(0, 1) (standard-normalised states)."""

import numpy as np

from gan_mpc_amd.nn_init import lecun_normal


def _mlp(rng, sizes):
    return [(lecun_normal(rng, a, b), np.zeros(b, np.float32)) for a, b in zip(sizes[:-1], sizes[1:])]


def make_problem(n, m, T, B, seed=0, dyn_hidden=(200, 200, 200), cost_hidden=(128, 128), cost_fout=10,
                 lstm_features=64, head_hidden=()):
    rng = np.random.default_rng(seed)
    F = lstm_features
    f32 = np.float32
    return dict(
        n=n, m=m, T=T, B=B,
        dyn=_mlp(rng, (n + m, *dyn_hidden, n)),
        cmlp=_mlp(rng, (n, *cost_hidden, cost_fout)),
        critic=dict(Wx=lecun_normal(rng, n, 4 * F), Wh=lecun_normal(rng, F, 4 * F),
                    b=np.zeros(4 * F, f32), head=_mlp(rng, (F, *head_hidden, 1))),
        mpc_w=np.array([-2.0, 3.0, -3.0], f32),
        x0=rng.standard_normal((B, n)).astype(f32),
        U=np.tanh(rng.standard_normal((B, T, m))).astype(f32),
        goal=rng.standard_normal((B, T + 1, n)).astype(f32),
        true_seq=rng.standard_normal((B, T + 1, n)).astype(f32),
    )
