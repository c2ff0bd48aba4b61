"""Factories and small helpers with the reference's names (reference utils.py:26-35,159-213)."""

import time

from gan_mpc_amd.config import load_config
from gan_mpc_amd.cost import cost_model
from gan_mpc_amd.cost import nn as cost_nn
from gan_mpc_amd.critic import critic_model
from gan_mpc_amd.critic import nn as critic_nn
from gan_mpc_amd.dynamics import dynamics_model
from gan_mpc_amd.dynamics import nn as dynamics_nn
from gan_mpc_amd.optim import get_masked_labels  # noqa: F401


def timeit(fn):
    def wrapper_fn(*args, **kwargs):
        start_time = time.time()
        ret = fn(*args, **kwargs)
        exe_time = (time.time() - start_time) / 60
        if isinstance(ret, tuple):
            return *ret, exe_time
        return ret, exe_time

    return wrapper_fn


def get_config(config_path):
    return load_config.Config.from_yaml(config_path)


def get_cost_model(config):
    model_config = config.mpc.model.cost
    mlp_config = model_config.mlp
    nn_model = cost_nn.MLP(num_layers=mlp_config.num_layers,
                           num_hidden_units=mlp_config.num_hidden_units, fout=mlp_config.fout)
    return cost_model.MujocoBasedModel(config, nn_model), model_config


def get_dynamics_model(config, x_size):
    model_config = config.mpc.model.dynamics
    if model_config.use == "lstm":
        lstm_config = model_config.lstm
        nn_model = dynamics_nn.LSTM(lstm_features=lstm_config.lstm_features,
                                    num_layers=lstm_config.num_layers,
                                    num_hidden_units=lstm_config.num_hidden_units, x_out=x_size)
    elif model_config.use == "mlp":
        mlp_config = model_config.mlp
        nn_model = dynamics_nn.MLP(num_layers=mlp_config.num_layers,
                                   num_hidden_units=mlp_config.num_hidden_units, x_out=x_size)
    else:
        raise ValueError("Choose either mlp or lstm model.")
    return dynamics_model.DynamicsModel(config, nn_model), model_config


def get_critic_model(config):
    model_config = config.mpc.model.critic
    if model_config.use == "lstm":
        lstm_config = model_config.lstm
        nn_model = critic_nn.LSTM(lstm_features=lstm_config.lstm_features,
                                  num_layers=lstm_config.num_layers,
                                  num_hidden_units=lstm_config.num_hidden_units)
    else:
        raise ValueError("Choose lstm model.")
    return critic_model.CriticModel(config, nn_model), model_config
