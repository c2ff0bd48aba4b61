"""Cost-parameter training (reference norm/cost_trainer.py:12-93): minibatches of
loss_and_grad + clip/Adam, test loss, Polyak blend.  `key` is a numpy seed/Generator (JAX's threefry
stream is not reproduced).  With torch.distributed initialised every rank draws the same minibatch
indices and processes its shard of each minibatch; the all-reduce inside loss_and_grad makes the
replicas take identical steps."""

import numpy as np

from gan_mpc_amd import parallel, utils


def _rng(key):
    return key if isinstance(key, np.random.Generator) else np.random.default_rng(key)


def _expert_select(policy, idx):
    sel = getattr(policy.expert_model, "select", None)
    if sel is not None:
        sel(idx)


def calculate_loss(policy, params, dataset):
    batch_x, batch_y = dataset
    lo, hi = parallel.shard_range(len(batch_x))
    _expert_select(policy, np.arange(lo, hi) + getattr(policy, "_test_offset", 0))
    return policy.batch_loss(params, batch_x[lo:hi], batch_y[lo:hi])


def train_cost_parameters(train_args, opt_state, params, perm, dataset):
    policy, opt = train_args
    X, Y = dataset
    losses = []
    for p in perm:
        lo, hi = parallel.shard_range(len(p))
        ps = p[lo:hi]
        _expert_select(policy, ps)
        loss, grads = policy.loss_and_grad(X[ps], params, (Y[ps],))
        params, opt_state = opt.update(policy._engine, params, grads, opt_state)
        losses.append(loss)
    return params, opt_state, sum(float(l) for l in losses) / len(losses)


@utils.timeit
def train(train_args, opt_state, params, dataset, num_updates, batch_size, polyak_factor, key, id):
    del id
    policy, opt = train_args
    rng = _rng(key)
    train_data, test_data = dataset
    params = policy.to_device_params(params)
    prev_params = params.clone()
    datasize = train_data[0].shape[0]
    steps_per_update = datasize // batch_size
    policy._test_offset = datasize
    train_losses, test_losses = [], []
    for _ in range(1, num_updates + 1):
        perm = rng.choice(datasize, size=(steps_per_update, batch_size))
        params, opt_state, train_loss = train_cost_parameters(
            train_args=(policy, opt), opt_state=opt_state, params=params, perm=perm,
            dataset=train_data)
        test_loss = calculate_loss(policy=policy, params=params, dataset=test_data)
        train_losses.append(float(train_loss))
        test_losses.append(float(test_loss))
    # Polyak blend over every leaf (cost_trainer.py:88-92)
    policy._engine.polyak(prev_params.flat, params.flat, polyak_factor, out=params.flat)
    return params, opt_state, train_losses, test_losses
