"""Critic training (reference gan/critic_trainer.py:12-104): dataset = true sequences (+1) and the
iLQR-predicted ones (-1); minibatches of critic_loss_and_grad + clip/Adam; test loss."""

import numpy as np
import torch

from gan_mpc_amd import parallel, utils
from gan_mpc_amd.norm.cost_trainer import _expert_select, _rng


def get_dataset(policy, params, true_dataset, key):
    rng = _rng(key)

    def func(X, true_Y, offset):
        datasize = true_Y.shape[0]
        _expert_select(policy, np.arange(datasize) + offset)
        xc, *_ = policy.get_optimal_values(params, X)          # (N, T+1, n) device
        xsize = X.shape[-1]
        pred_Y = xc[..., :xsize]
        true_t = torch.as_tensor(np.asarray(true_Y, np.float32), device=pred_Y.device)
        label = torch.cat([torch.ones(datasize), -torch.ones(datasize)]).to(pred_Y.device)
        return torch.cat([true_t, pred_Y], dim=0).contiguous(), label

    true_train_data, true_test_data = true_dataset
    train_X, train_label = func(*true_train_data, 0)
    test_X, test_label = func(*true_test_data, true_train_data[0].shape[0])
    perm = torch.as_tensor(rng.permutation(train_X.shape[0]), device=train_X.device)
    return (train_X[perm], train_label[perm]), (test_X, test_label)


def calculate_loss(policy, params, dataset):
    X, Y = dataset
    lo, hi = parallel.shard_range(X.shape[0])
    dparams = policy.to_device_params(params)
    ls, _ = policy._critic_sums(X[lo:hi].contiguous(), Y[lo:hi].contiguous(), dparams)
    return parallel.allreduce_mean_from_sums(ls.reshape(1).clone(), hi - lo)[0]


def train_critic_parameters(train_args, opt_state, params, perm, dataset):
    policy, opt = train_args
    X, Y = dataset
    losses = []
    for p in perm:
        lo, hi = parallel.shard_range(len(p))
        idx = torch.as_tensor(p[lo:hi], device=X.device)
        loss, grads = policy.critic_loss_and_grad(X[idx].contiguous(), Y[idx].contiguous(), params)
        params, opt_state = opt.update(policy._engine, params, grads, opt_state)
        losses.append(loss)
    return params, opt_state, sum(float(l) for l in losses) / len(losses)


@utils.timeit
def train(train_args, opt_state, params, true_dataset, num_updates, batch_size, key, id):
    del id
    policy, opt = train_args
    rng = _rng(key)
    params = policy.to_device_params(params)
    train_data, test_data = get_dataset(policy, params, true_dataset, rng)
    datasize = train_data[0].shape[0]
    steps_per_update = datasize // batch_size
    train_losses, test_losses = [], []
    for _ in range(1, num_updates + 1):
        perm = rng.choice(datasize, size=(steps_per_update, batch_size))
        params, opt_state, train_loss = train_critic_parameters(
            train_args=(policy, opt), opt_state=opt_state, params=params, perm=perm,
            dataset=train_data)
        test_loss = calculate_loss(policy=policy, params=params, dataset=test_data)
        train_losses.append(float(train_loss))
        test_losses.append(float(test_loss))
    return params, opt_state, train_losses, test_losses
