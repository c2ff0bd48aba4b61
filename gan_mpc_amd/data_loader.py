"""Expert-trajectory loader and the training-set windowing (reference data_loader.py:12-129).

Same class, method names and selection rules; differences: `key` is a NumPy seed or Generator (the
JAX threefry permutation is not reproduced), windows are cut with strided views, and the trajectory
file may be given explicitly (`init(path=...)`) because no dataset ships with either repository."""

import json
import os

import numpy as np

from gan_mpc_amd.data_buffers import sliding_windows

_MAIN_DIR_PATH = os.path.dirname(__file__)
_REWARD_THRESHOLD = 500      # reference data_loader.py:24-28 ("ensure expert trajectories are proper")


def _rng(key):
    return key if isinstance(key, np.random.Generator) else np.random.default_rng(key)


class DataLoader:
    def __init__(self, config, normalizer):
        self.config = config
        self.normalizer = normalizer
        self.expert_trajectories = None

    def get_expert_trajectories(self, path, num_trajectories, trajectory_len):
        """Best `num_trajectories` by summed reward among those above the threshold, first
        `trajectory_len` steps of each (reference :18-33)."""
        with open(path, "r") as fp:
            data = json.load(fp)
        total = np.sum(data["rewards"], axis=1)
        order = np.argsort(-total)
        idx = [i for i in order if total[i] > _REWARD_THRESHOLD][:num_trajectories]
        return {k: np.array(data[k])[idx, :trajectory_len]
                for k in ("states", "actions", "rewards") if k in data}

    def init(self, path=None):
        config = self.config
        if path is None:
            env_type, env_name = config.env.type, config.env.expert.name
            path = os.path.join(_MAIN_DIR_PATH,
                                f"expert_trajectories/{env_type}/{env_name}/trajectories.json")
        self.expert_trajectories = self.get_expert_trajectories(
            path=path, num_trajectories=config.mpc.train.num_trajectories,
            trajectory_len=config.mpc.train.trajectory_len)
        self.normalizer.update(state_dataset=self.expert_trajectories["states"],
                               action_dataset=self.expert_trajectories["actions"])
        rewards = np.sum(self.expert_trajectories["rewards"], axis=1)
        print(f"Expert trajectories reward mean: {np.mean(rewards):.3f} "
              f"and reward std: {np.std(rewards):.3f}")
        return self

    def shuffle_and_split_dataset(self, dataset, key, train_split=0.8):
        size = dataset[0].shape[0]
        cut = int(size * train_split)
        perm = _rng(key).permutation(size)
        return (tuple(d[perm[:cut]] for d in dataset), tuple(d[perm[cut:]] for d in dataset))

    def _require_init(self):
        if self.expert_trajectories is None:
            raise Exception("Please call init before calling get_cost_dataset.")

    def get_cost_dataset(self, key):
        """X = history+1 states ending at step i (zero-padded before the start), Y = the next
        horizon+1 states from i, for i in [history, len - horizon) (reference :68-91)."""
        self._require_init()
        s_trajs = self.normalizer.normalize_state(self.expert_trajectories["states"])
        horizon, history = self.config.mpc.horizon, self.config.mpc.history
        X, Y = [], []
        for s_traj in s_trajs:
            traj_len, xsize = s_traj.shape
            count = traj_len - horizon - history
            padded = np.concatenate([np.zeros((history, xsize)), s_traj], axis=0)
            X.append(sliding_windows(padded, history + 1, count, start=0))
            Y.append(sliding_windows(padded, horizon + 1, count, start=history))
        return self.shuffle_and_split_dataset((np.concatenate(X, 0), np.concatenate(Y, 0)), key)

    def get_dynamics_dataset(self, key):
        train_dataset, _ = self.get_expert_dataset(key, seqlen=self.config.mpc.horizon)
        return train_dataset

    def get_expert_dataset(self, key, seqlen=None):
        self._require_init()
        s_trajs, a_trajs = self.normalizer.normalize(
            state_dataset=self.expert_trajectories["states"],
            action_dataset=self.expert_trajectories["actions"])
        seqlen = seqlen or self.config.expert_prediction.train.seqlen
        X, U, Y = [], [], []
        for s_traj, a_traj in zip(s_trajs, a_trajs):
            count = s_traj.shape[0] - seqlen
            X.append(sliding_windows(s_traj, seqlen, count))
            U.append(sliding_windows(a_traj, seqlen, count))
            Y.append(sliding_windows(s_traj, seqlen, count, start=1))
        return self.shuffle_and_split_dataset(
            (np.concatenate(X, 0), np.concatenate(U, 0), np.concatenate(Y, 0)), key)
