"""State / action normalisers (reference data_normalizer.py:6-70, same class and method names).

Host-side NumPy: these run once per dataset / once per environment step, never in the hot path."""

import numpy as np


class BaseNormalizer:
    def update(self, dataset, *args, **kwargs):
        raise NotImplementedError

    def normalize(self, dataset, *args, **kwargs):
        raise NotImplementedError


class IdentityNormalizer(BaseNormalizer):
    """reference data_normalizer.py:14-20"""

    def update(self, dataset, *args, **kwargs):
        return None

    def normalize(self, dataset):
        return np.array(dataset)


class StandardNormalizer(BaseNormalizer):
    """Per-feature mean / population std over every leading axis (reference :23-44)."""

    def __init__(self, mean=None, std=None, verbose=True):
        self.mean = mean
        self.std = std
        self.verbose = verbose

    def update(self, dataset):
        data = np.array(dataset)
        lead = tuple(range(data.ndim - 1))
        self.mean = data.mean(axis=lead)
        self.std = data.std(axis=lead)
        if self.verbose:
            print(f"mean: {self.mean}")
            print(f"std: {self.std}")

    def normalize(self, dataset):
        return (np.array(dataset) - self.mean) / self.std

    # on-disk form used by utils.save_all_args (plain arrays, no pickle)
    def state_dict(self):
        return {"mean": np.asarray(self.mean), "std": np.asarray(self.std)}

    def load_state_dict(self, d):
        self.mean, self.std = np.asarray(d["mean"]), np.asarray(d["std"])
        return self


class JointNormalizer(BaseNormalizer):
    """reference data_normalizer.py:47-70"""

    def __init__(self, state_normalizer: BaseNormalizer, action_normalizer: BaseNormalizer):
        self.state_normalizer = state_normalizer
        self.action_normalizer = action_normalizer

    def update(self, state_dataset, action_dataset):
        self.state_normalizer.update(state_dataset)
        self.action_normalizer.update(action_dataset)

    def normalize_state(self, state_dataset):
        return self.state_normalizer.normalize(state_dataset)

    def normalize_action(self, action_dataset):
        return self.action_normalizer.normalize(action_dataset)

    def normalize(self, state_dataset, action_dataset):
        return self.normalize_state(state_dataset), self.normalize_action(action_dataset)
