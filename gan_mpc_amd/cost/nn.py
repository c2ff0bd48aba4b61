"""Cost network descriptor (reference cost/nn.py:10-29): relu MLP -> dot(y, y)."""

import numpy as np

from gan_mpc_amd import base, nn_init


class MLP(base.BaseCostNN):
    def __init__(self, num_layers, num_hidden_units, fout):
        self.num_layers = int(num_layers)
        self.num_hidden_units = int(num_hidden_units)
        self.fout = int(fout)

    def dims(self, xc_size):
        return [int(xc_size)] + [self.num_hidden_units] * (self.num_layers - 1) + [self.fout]

    def get_init_params(self, seed, xc_size):
        return (int(seed), int(xc_size))

    def init(self, seed, xc_size):
        return nn_init.dense_tree(np.random.default_rng(seed), self.dims(xc_size))
