"""N > 1 host logic on CPU (gloo, world_size 2): trajectory sharding and the single all-reduce of
the packed [loss_sum | grad_sum] buffer reproduce the single-process batch mean."""

import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gan_mpc_oracle as orc
from gan_mpc_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out):
    for p in (ROOT, os.path.join(ROOT, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import gan_mpc_oracle as orc_  # noqa: F401
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        pb = orc.make_problem(4, 2, 5, B, seed=9, lstm_features=8, dyn_hidden=(12,),
                              cost_hidden=(10,), cost_fout=3)
        lo, hi = parallel.shard_range(B)
        label = np.where(np.arange(B) % 2 == 0, 1.0, -1.0).astype(np.float32)
        # this rank's partial SUMS (what gmpc_critic_loss_grad returns), from the oracle
        n_loc = hi - lo
        l, g = orc.critic_loss_and_grad(pb["critic"], pb["true_seq"][lo:hi], label[lo:hi])
        flat = np.concatenate([g["Wx"].ravel(), g["Wh"].ravel(), g["b"].ravel()]
                              + [t.ravel() for Wb in g["head"] for t in Wb])
        packed = parallel.new_packed(1 + flat.size, "cpu", n_loc)
        packed[:-1] = torch.from_numpy(np.concatenate([[l * n_loc], flat * n_loc]).astype(np.float32))
        work = parallel.allreduce_start(packed)          # the bench's two-call form
        means = parallel.allreduce_finish(packed, work)
        out[rank] = (lo, hi, means.numpy().copy())
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_everything():
    for count in (1, 7, 8, 1024, 1025):
        for ws in (1, 2, 3, 8):
            spans = [parallel.shard_range(count, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_mean():
    packed = torch.tensor([10.0, 4.0, -6.0, 4.0])        # [sums | sample count]
    means = parallel.allreduce_mean_from_sums(packed)
    np.testing.assert_allclose(means.numpy(), [2.5, 1.0, -1.5])


def _empty_shard_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = parallel.shard_range(1)                 # one sample, two ranks: rank 1 owns nothing
        packed = parallel.new_packed(2, "cpu", hi - lo)
        if hi > lo:
            packed[:2] = torch.tensor([3.0, -8.0])
        out[rank] = (hi - lo, parallel.allreduce_mean_from_sums(packed).numpy().copy())
    finally:
        dist.destroy_process_group()


def test_empty_shard_joins_the_exchange_with_count_zero():
    """batch smaller than the world size: the rank without samples contributes zeros and count 0,
    both ranks end with the mean over the one real sample."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_empty_shard_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert (out[0][0], out[1][0]) == (1, 0)
    for r in (0, 1):
        np.testing.assert_allclose(out[r][1], [3.0, -8.0])


def test_two_rank_allreduce_equals_single_process_batch_mean():
    B, world = 7, 2          # ragged shards: 4 + 3
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, B, out), nprocs=world, join=True)
    pb = orc.make_problem(4, 2, 5, B, seed=9, lstm_features=8, dyn_hidden=(12,),
                          cost_hidden=(10,), cost_fout=3)
    label = np.where(np.arange(B) % 2 == 0, 1.0, -1.0).astype(np.float32)
    l, g = orc.critic_loss_and_grad(pb["critic"], pb["true_seq"], label)
    flat = np.concatenate([g["Wx"].ravel(), g["Wh"].ravel(), g["b"].ravel()]
                          + [t.ravel() for Wb in g["head"] for t in Wb])
    want = np.concatenate([[l], flat])
    assert (out[0][0], out[0][1], out[1][0], out[1][1]) == (0, 4, 4, 7)
    np.testing.assert_array_equal(out[0][2], out[1][2])       # replicas stay identical
    np.testing.assert_allclose(out[0][2], want, rtol=2e-5, atol=1e-7)   # N-GPU vs 1-GPU mean


class _StubEngine:
    pass


class _StubPolicy:
    """What sgd_pass touches of a policy: `_engine` (None until a loss call with samples builds one)."""

    def __init__(self):
        self._engine = None
        self.built = 0

    def to_device_params(self, params):
        return params

    def engine_for(self, batch, dparams=None):
        self.built += 1
        self._engine = _StubEngine()
        return self._engine


class _StubOpt:
    def __init__(self):
        self.engines = []

    def update(self, engine, params, grads, opt_state):
        self.engines.append(engine)
        return params - 0.1 * grads, opt_state


def _sgd_pass_worker(rank, world, port, out):
    for p in (ROOT,):
        if p not in sys.path:
            sys.path.insert(0, p)
    from gan_mpc_amd import trainer_common as tc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        policy, opt = _StubPolicy(), _StubOpt()
        data = torch.tensor([2.0, -4.0, 6.0])

        def loss_and_grad(idx):
            # like BaseMPC.loss_and_grad: an engine only when this rank has samples; the exchange always
            packed = parallel.new_packed(2, "cpu", len(idx))
            if len(idx) > 0:
                policy.engine_for(len(idx))
                packed[0] = data[idx].sum()
                packed[1] = (2 * data[idx]).sum()
            means = parallel.allreduce_mean_from_sums(packed)
            return means[0], means[1:]

        schedule = np.array([[0], [2], [1]])            # minibatches of ONE sample: rank 1's shard is always empty
        params, _, mean_loss = tc.sgd_pass(policy, opt, {}, torch.zeros(1), schedule, loss_and_grad)
        out[rank] = (policy.built, all(e is not None for e in opt.engines), len(opt.engines),
                     float(params[0]), float(mean_loss))
    finally:
        dist.destroy_process_group()


def test_sgd_pass_with_a_minibatch_smaller_than_the_world():
    """batch_size < world_size: the rank that never receives a sample still takes every optimiser step (with an
    engine) and stays a replica of the other one -- it used to hit `None.adam_clip_step` while rank 0 went on to
    the next collective."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_sgd_pass_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for r in (0, 1):
        built, all_engines, steps, p, loss = out[r]
        assert all_engines and steps == 3 and built >= 1
        np.testing.assert_allclose(p, -0.1 * 2 * (2.0 + 6.0 - 4.0), rtol=1e-6)
        np.testing.assert_allclose(loss, (2.0 + 6.0 - 4.0) / 3, rtol=1e-6)
    assert out[0][3] == out[1][3]
