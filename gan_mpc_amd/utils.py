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


# ---- on-disk artefacts (reference utils.py:119-156) ---------------------------------------------
import json  # noqa: E402
import os  # noqa: E402

import numpy as np  # noqa: E402

_MAIN_DIR_PATH = os.path.dirname(__file__)


def _abs(path):
    return path if os.path.isabs(path) else os.path.join(_MAIN_DIR_PATH, path)


def check_or_create_dir(path):
    if not os.path.exists(path):
        os.makedirs(path, exist_ok=True)


def save_json(data, dir_path, basename):
    dir_path = _abs(dir_path)
    check_or_create_dir(dir_path)
    with open(os.path.join(dir_path, basename), "w") as fp:
        json.dump(data, fp, indent=4, sort_keys=True)


def load_json(path):
    with open(_abs(path), "r") as fp:
        return json.load(fp)


def flatten_tree(tree, prefix=""):
    """{'a': {'b': array}} -> {'a/b': array}; None leaves are dropped."""
    out = {}
    for k, v in tree.items():
        name = f"{prefix}/{k}" if prefix else str(k)
        if isinstance(v, dict):
            out.update(flatten_tree(v, name))
        elif v is not None:
            out[name] = np.asarray(v)
    return out


def unflatten_tree(flat):
    tree = {}
    for name, v in flat.items():
        node = tree
        parts = name.split("/")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = np.asarray(v)
    return tree


def save_all_args(dir_path, params, model_config, *other_json_args):
    """reference utils.py:135-147: next numbered sub-directory with config.json, the parameters and
    the loss curves.  Parameters are written as `params.npz` (one array per leaf, keys are the tree
    paths) instead of a pickled dict, so loading them never executes code."""
    abs_dir_path = _abs(dir_path)
    check_or_create_dir(abs_dir_path)
    runs = sorted((d for d in os.listdir(abs_dir_path) if d.isdigit()), key=lambda x: -int(x))
    key = "0" if not runs else f"{int(runs[0]) + 1}"
    full_path = os.path.join(abs_dir_path, key)
    save_json(model_config, full_path, "config.json")
    tree = params.to_tree() if hasattr(params, "to_tree") else params
    np.savez(os.path.join(full_path, "params.npz"), **flatten_tree(tree))
    for json_data, name in other_json_args:
        save_json(json_data, full_path, name)
    return full_path


def load_params(params_path, from_np=True, allow_pickle=False):
    """`params.npz` written by save_all_args, or the reference's pickled `params.npy`
    (utils.py:150-156) when the caller opts in with allow_pickle=True (pickle runs code: only for
    files you trust)."""
    if not from_np:
        raise NotImplementedError("params must be saved using numpy.")
    path = _abs(params_path)
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            return unflatten_tree({k: z[k] for k in z.files})
    if not allow_pickle:
        raise ValueError(f"{path}: a pickled .npy parameter file needs allow_pickle=True")
    return np.load(path, allow_pickle=True).item()


def get_expert_model(config, x_size, u_size):
    """reference utils.py:216-227: the pretrained behaviour-cloning sequence model saved under
    trained_models/expert/<env type>/<env name>/<load_id>/ (config.json names its architecture).  No
    such model ships with either repository; when none is found the goal is "hold the
    current state" with zero initial controls."""
    from gan_mpc_amd.expert import expert_model
    env_type, env_name = config.env.type, config.env.expert.name
    env_id = config.mpc.model.expert.load_id
    saved_config_path = f"trained_models/expert/{env_type}/{env_name}/{env_id}/config.json"
    if not os.path.exists(_abs(saved_config_path)):
        return expert_model.HoldExpert(config.mpc.horizon, u_size)
    saved_config = load_json(saved_config_path)
    model_config = load_config.Config.from_dict(saved_config["model"])
    nn_model = expert_model.ExpertModel.get_model(model_config=model_config, x_size=x_size,
                                                  u_size=u_size)
    return expert_model.ExpertModel(config, nn_model)
