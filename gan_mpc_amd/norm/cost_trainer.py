"""Training of the cost parameters and the MPC weights (reference norm/cost_trainer.py:12-93): per
update a pass of bilevel loss_and_grad + clip/Adam over freshly sampled minibatches, then the test
loss; at the end every parameter is blended with its value at entry (Polyak, :88-92)."""

import numpy as np

from gan_mpc_amd import parallel, trainer_common as tc, utils

_rng = tc.as_rng                      # names kept for the other trainers
_expert_select = tc.select_expert_rows


def calculate_loss(policy, params, dataset):
    """Mean upper-level loss of iLQR(x) over `dataset` (this rank's shard, then all-reduced)."""
    batch_x, batch_y = dataset
    lo, hi = parallel.shard_range(len(batch_x))
    tc.select_expert_rows(policy, np.arange(lo, hi) + getattr(policy, "_test_offset", 0))
    return policy.batch_loss(params, batch_x[lo:hi], batch_y[lo:hi])


def train_cost_parameters(train_args, opt_state, params, perm, dataset):
    policy, opt = train_args
    X, Y = dataset

    def step(idx):
        tc.select_expert_rows(policy, idx)
        return policy.loss_and_grad(X[idx], params, (Y[idx],))

    return tc.sgd_pass(policy, opt, opt_state, params, perm, step)


@utils.timeit
def train(train_args, opt_state, params, dataset, num_updates, batch_size, polyak_factor, key, id):
    del id
    policy, opt = train_args
    rng = tc.as_rng(key)
    train_data, test_data = dataset
    params = policy.to_device_params(params)
    entry_params = params.clone()
    datasize = train_data[0].shape[0]
    policy._test_offset = datasize          # rows of a table expert: test samples follow the train ones
    train_losses, test_losses = [], []
    for _ in range(num_updates):
        schedule = tc.minibatch_schedule(rng, datasize, batch_size)
        params, opt_state, loss = train_cost_parameters((policy, opt), opt_state, params, schedule,
                                                        train_data)
        train_losses.append(float(loss))
        test_losses.append(float(calculate_loss(policy, params, test_data)))
    policy._engine.polyak(entry_params.flat, params.flat, polyak_factor, out=params.flat)
    return params, opt_state, train_losses, test_losses
