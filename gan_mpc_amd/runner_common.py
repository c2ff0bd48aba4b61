"""Pieces shared by gan/runner.py and norm/runner.py (the reference repeats them in both files:
gan/runner.py:37-82 == norm/runner.py:32-77)."""

import numpy as np

from gan_mpc_amd import data_normalizer, optim


def get_params(policy, config, x_size, u_size, with_critic):
    """reference gan/runner.py:37-48.  The MLP dynamics carry is empty, so xc_size == x_size."""
    seed = config.seed
    carry = policy.get_dynamics_carry(np.zeros((1, x_size), np.float32))
    xc_size = x_size + carry.shape[-1]
    mpc_weights = tuple(config.mpc.model.cost.weights.to_dict().values())
    args = [mpc_weights, (seed, xc_size), (seed, u_size), (True,)]
    if with_critic:
        args.append((seed, x_size))
    return policy.init(*args)


def get_optimizer(params, masked_vars, lr, policy=None):
    """reference gan/runner.py:51-63 -> (opt, opt_state).  The state lives on the device next to
    the flat parameter vector, so it needs the policy (for the device) when `params` is a tree."""
    opt = optim.get_optimizer(list(params.keys()) if isinstance(params, dict) else list(params.KEYS),
                              masked_vars, lr)
    dparams = policy.to_device_params(params) if policy is not None else params
    return opt, opt.init(dparams)


def get_normalizer(norm_config):
    """reference gan/runner.py:66-81"""
    if norm_config.state == "standard_norm":
        state_normalizer = data_normalizer.StandardNormalizer()
    else:
        state_normalizer = data_normalizer.IdentityNormalizer()
    if norm_config.action == "identity":
        action_normalizer = data_normalizer.IdentityNormalizer()
    else:
        raise Exception(f"Please set appropriate action normalizer. Given: {norm_config.action}")
    return data_normalizer.JointNormalizer(state_normalizer=state_normalizer,
                                           action_normalizer=action_normalizer)


def split_keys(key, count):
    """`count` independent child generators + the advanced parent (stands in for jax.random.split)."""
    rng = key if isinstance(key, np.random.Generator) else np.random.default_rng(key)
    children = rng.spawn(count)
    return rng, children
