"""Critic network descriptor (reference critic/nn.py:10-42): LSTM(features) over the sequence,
final h -> (num_layers-1) relu Dense(num_hidden_units) -> Dense(fout=1)."""

import numpy as np

from gan_mpc_amd import base, nn_init


class LSTM(base.BaseNN):
    def __init__(self, lstm_features, num_layers, num_hidden_units, fout=1):
        self.lstm_features = int(lstm_features)
        self.num_layers = int(num_layers)
        self.num_hidden_units = int(num_hidden_units)
        self.fout = int(fout)
        if self.fout != 1:
            raise ValueError("the critic head ends in one score (reference critic/nn.py:14)")

    def head_dims(self):
        return [self.lstm_features] + [self.num_hidden_units] * (self.num_layers - 1) + [self.fout]

    def get_init_params(self, seed, xsize):
        return (int(seed), int(xsize))

    def init(self, seed, xsize):
        return nn_init.lstm_critic_tree(np.random.default_rng(seed), int(xsize), self.lstm_features,
                                        self.head_dims())
