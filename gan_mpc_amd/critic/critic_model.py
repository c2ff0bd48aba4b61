"""Critic model wrapper (reference critic/critic_model.py:6-16)."""

from gan_mpc_amd import base


class CriticModel(base.BaseCriticModel):
    def __init__(self, config, model):
        self.config = config
        self.model = model

    def init(self, *args):
        model_args = self.model.get_init_params(*args)
        return self.model.init(*model_args)

    def predict(self, xseq, params, policy=None):
        """score (1,) of one sequence (T+1, n), or (B,) scores of a batch, by the LSTM kernels."""
        if policy is None:
            raise ValueError("predict needs the policy that owns the HIP engine (policy=...)")
        return policy.critic_scores(xseq, params)
