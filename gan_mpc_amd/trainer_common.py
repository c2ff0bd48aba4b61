"""What the three trainers (cost, critic, dynamics) have in common: the minibatch schedule of the
reference (`jax.random.choice(key, datasize, shape=(steps_per_update, batch_size))` once per update,
reference norm/cost_trainer.py:76-78, gan/critic_trainer.py:88-90, norm/dynamics_trainer.py:107-110)
and the loop  minibatch -> this rank's shard -> loss and gradient -> clip + Adam.

`key` is a NumPy seed or Generator (JAX's threefry stream is not reproduced).  Under torch.distributed
every rank draws the same indices and processes its contiguous shard of each minibatch; the all-reduce
inside the loss makes the replicas take identical steps."""

import numpy as np

from gan_mpc_amd import parallel


def as_rng(key):
    return key if isinstance(key, np.random.Generator) else np.random.default_rng(key)


def minibatch_schedule(rng, datasize, batch_size):
    """Index matrix (datasize // batch_size, batch_size), sampled with replacement like the reference."""
    return rng.choice(datasize, size=(datasize // batch_size, batch_size))


def sgd_pass(policy, opt, opt_state, params, schedule, loss_and_grad):
    """One pass over `schedule`: loss_and_grad(local_indices) -> (loss, flat gradient), both already
    averaged over the global minibatch.  Returns (params, opt_state, mean loss of the pass)."""
    total, steps = 0.0, 0
    for batch in schedule:
        lo, hi = parallel.shard_range(len(batch))
        loss, grads = loss_and_grad(batch[lo:hi])
        # the optimiser step runs on EVERY rank, also on one whose shard of this minibatch was empty and that
        # therefore built no engine inside the loss (minibatch smaller than the world): make sure there is one
        eng = policy._engine
        if eng is None:
            eng = policy.engine_for(1, policy.to_device_params(params))
        params, opt_state = opt.update(eng, params, grads, opt_state)
        total += float(loss)
        steps += 1
    # datasize < batch_size leaves no minibatch: the reference's mean over an empty scan is NaN
    return params, opt_state, (total / steps if steps else float("nan"))


def select_expert_rows(policy, idx):
    """Table-driven experts (tests, synthetic workloads) need to know which samples are coming."""
    select = getattr(policy.expert_model, "select", None)
    if select is not None:
        select(idx)
