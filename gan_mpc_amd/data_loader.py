"""Expert trajectories -> training sets (the behaviour of reference data_loader.py:12-129 behind the
same class and method names).

trajectories.json holds {"states": [N][L][n], "actions": [N][L][m], "rewards": [N][L]}.  The loader
keeps the best trajectories, fits the normaliser on them and cuts three kinds of windows:
  cost set      X = the history+1 states up to step i (zeros before the start), Y = the horizon+1
                states from step i on
  expert set    (states, actions, next states) windows of `seqlen` steps
  dynamics set  the training split of the expert set with seqlen = horizon
Differences from the reference: `key` is a NumPy seed or Generator (the JAX threefry permutation is
not reproduced), windows are strided views instead of Python loops, and the file may be named
explicitly (`init(path=...)`) because no dataset ships with either repository."""

import json
import os

import numpy as np

from gan_mpc_amd.data_buffers import sliding_windows

_MAIN_DIR_PATH = os.path.dirname(__file__)
_MIN_RETURN = 500      # reference data_loader.py:24-28: trajectories at or below this return are dropped
_FIELDS = ("states", "actions", "rewards")


def _generator(key):
    return key if isinstance(key, np.random.Generator) else np.random.default_rng(key)


def _stack(per_trajectory):
    """[(a0, b0, ...), (a1, b1, ...)] -> (concat a, concat b, ...)"""
    return tuple(np.concatenate(parts, axis=0) for parts in zip(*per_trajectory))


class DataLoader:
    def __init__(self, config, normalizer):
        self.config = config
        self.normalizer = normalizer
        self.expert_trajectories = None

    # ---- loading ---------------------------------------------------------------------------------
    def get_expert_trajectories(self, path, num_trajectories, trajectory_len):
        """The `num_trajectories` highest-return trajectories above the threshold, cut to
        `trajectory_len` steps (reference :18-33)."""
        with open(path, "r") as fp:
            raw = json.load(fp)
        returns = np.sum(raw["rewards"], axis=1)
        ranked = [i for i in np.argsort(-returns) if returns[i] > _MIN_RETURN]
        keep = ranked[:num_trajectories]
        return {name: np.array(raw[name])[keep, :trajectory_len] for name in _FIELDS if name in raw}

    def init(self, path=None):
        train_cfg = self.config.mpc.train
        if path is None:
            env = self.config.env
            path = os.path.join(_MAIN_DIR_PATH, "expert_trajectories", env.type, env.expert.name,
                                "trajectories.json")
        trajs = self.get_expert_trajectories(path=path, num_trajectories=train_cfg.num_trajectories,
                                             trajectory_len=train_cfg.trajectory_len)
        self.expert_trajectories = trajs
        self.normalizer.update(state_dataset=trajs["states"], action_dataset=trajs["actions"])
        returns = np.sum(trajs["rewards"], axis=1)
        print(f"Expert trajectories reward mean: {np.mean(returns):.3f} "
              f"and reward std: {np.std(returns):.3f}")
        return self

    def _trajectories(self):
        if self.expert_trajectories is None:
            raise Exception("Please call init before calling get_cost_dataset.")
        return self.expert_trajectories

    # ---- splitting -------------------------------------------------------------------------------
    def shuffle_and_split_dataset(self, dataset, key, train_split=0.8):
        count = dataset[0].shape[0]
        order = _generator(key).permutation(count)
        head, tail = order[:int(count * train_split)], order[int(count * train_split):]
        return tuple(d[head] for d in dataset), tuple(d[tail] for d in dataset)

    # ---- windows ---------------------------------------------------------------------------------
    def get_cost_dataset(self, key):
        """reference :68-91: for i in [history, len - horizon) of the zero-padded trajectory,
        X = padded[i-history : i+1], Y = padded[i : i+horizon+1]."""
        trajs = self._trajectories()
        states = self.normalizer.normalize_state(trajs["states"])
        horizon, history = self.config.mpc.horizon, self.config.mpc.history

        def cut(traj):
            padded = np.concatenate([np.zeros((history, traj.shape[1])), traj], axis=0)
            count = traj.shape[0] - horizon - history
            return (sliding_windows(padded, history + 1, count, start=0),
                    sliding_windows(padded, horizon + 1, count, start=history))

        return self.shuffle_and_split_dataset(_stack([cut(t) for t in states]), key)

    def get_expert_dataset(self, key, seqlen=None):
        trajs = self._trajectories()
        states, actions = self.normalizer.normalize(state_dataset=trajs["states"],
                                                    action_dataset=trajs["actions"])
        seqlen = seqlen or self.config.expert_prediction.train.seqlen

        def cut(s_traj, a_traj):
            count = s_traj.shape[0] - seqlen
            return (sliding_windows(s_traj, seqlen, count), sliding_windows(a_traj, seqlen, count),
                    sliding_windows(s_traj, seqlen, count, start=1))

        return self.shuffle_and_split_dataset(_stack([cut(s, a) for s, a in zip(states, actions)]), key)

    def get_dynamics_dataset(self, key):
        train_split, _ = self.get_expert_dataset(key, seqlen=self.config.mpc.horizon)
        return train_split
