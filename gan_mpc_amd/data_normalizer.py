"""State / action normalisers with the reference's class and method names (reference
data_normalizer.py:6-70).  Host-side NumPy: they run once per dataset and once per environment step,
never in the hot path.

All of them are the affine map  x -> (x - shift) / scale  applied over the last axis; the identity
keeps shift = 0, scale = 1, the standard normaliser fits both to a dataset."""

import numpy as np


class BaseNormalizer:
    """Interface: update(dataset, ...) fits the statistics, normalize(dataset, ...) applies them."""

    def update(self, dataset, *args, **kwargs):
        raise NotImplementedError

    def normalize(self, dataset, *args, **kwargs):
        raise NotImplementedError


class _Affine(BaseNormalizer):
    shift = None      # None: nothing to subtract / divide by
    scale = None

    def normalize(self, dataset):
        data = np.array(dataset)
        if self.shift is None:
            return data
        return (data - self.shift) / self.scale


class IdentityNormalizer(_Affine):
    """Leaves the data alone (reference :14-20)."""

    def update(self, dataset, *args, **kwargs):
        return None


class StandardNormalizer(_Affine):
    """Per-feature mean and population standard deviation over every leading axis (reference :23-44);
    `mean` / `std` are the reference's attribute names."""

    def __init__(self, mean=None, std=None, verbose=True):
        self.shift, self.scale = mean, std
        self.verbose = verbose

    mean = property(lambda self: self.shift, lambda self, v: setattr(self, "shift", v))
    std = property(lambda self: self.scale, lambda self, v: setattr(self, "scale", v))

    def update(self, dataset):
        data = np.array(dataset)
        samples = data.reshape(-1, data.shape[-1])         # all leading axes are samples
        self.shift, self.scale = samples.mean(axis=0), samples.std(axis=0)
        if self.verbose:
            print(f"mean: {self.shift}")
            print(f"std: {self.scale}")

    # plain arrays for utils.save_all_args (no pickle)
    def state_dict(self):
        return {"mean": np.asarray(self.shift), "std": np.asarray(self.scale)}

    def load_state_dict(self, state):
        self.shift, self.scale = np.asarray(state["mean"]), np.asarray(state["std"])
        return self


class JointNormalizer(BaseNormalizer):
    """A state normaliser and an action normaliser behind one object (reference :47-70)."""

    def __init__(self, state_normalizer: BaseNormalizer, action_normalizer: BaseNormalizer):
        self.state_normalizer = state_normalizer
        self.action_normalizer = action_normalizer

    def update(self, state_dataset, action_dataset):
        for normalizer, data in ((self.state_normalizer, state_dataset),
                                 (self.action_normalizer, action_dataset)):
            normalizer.update(data)

    def normalize_state(self, state_dataset):
        return self.state_normalizer.normalize(state_dataset)

    def normalize_action(self, action_dataset):
        return self.action_normalizer.normalize(action_dataset)

    def normalize(self, state_dataset, action_dataset):
        return self.normalize_state(state_dataset), self.normalize_action(action_dataset)
