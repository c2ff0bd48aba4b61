"""Dynamics network descriptor (reference dynamics/nn.py:10-34): residual relu MLP, empty carry.
The LSTM variant (dynamics/nn.py:37-57) is not built."""

import numpy as np

from gan_mpc_amd import base, nn_init


class MLP(base.BaseDynamicsNN):
    def __init__(self, num_layers, num_hidden_units, x_out):
        self.num_layers = int(num_layers)
        self.num_hidden_units = int(num_hidden_units)
        self.x_out = int(x_out)

    def dims(self, u_size):
        return ([self.x_out + int(u_size)] + [self.num_hidden_units] * (self.num_layers - 1)
                + [self.x_out])

    def get_carry(self, x):
        return np.empty((*np.shape(x)[:-1], 0), np.float32)

    def get_init_params(self, seed, u_size):
        return (int(seed), int(u_size))

    def init(self, seed, u_size):
        return nn_init.dense_tree(np.random.default_rng(seed), self.dims(u_size))


class LSTM(MLP):
    def __init__(self, *a, **k):
        raise NotImplementedError(
            "the LSTM dynamics variant (reference dynamics/nn.py:37-57) is not on the built path")
