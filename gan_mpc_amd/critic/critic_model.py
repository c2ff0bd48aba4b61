"""The discriminator as the policies see it (reference critic/critic_model.py:6-16): parameter
initialisation is delegated to the network description, scoring runs on the GPU through the policy
that owns the engine (LSTM forward + head, gmpc_critic_score_vjp)."""

from gan_mpc_amd import base


class CriticModel(base.BaseCriticModel):
    def __init__(self, config, model):
        super().__init__(config)
        self.model = model

    def init(self, *args):
        return self.model.init(*self.model.get_init_params(*args))

    def predict(self, xseq, params, policy=None):
        """One sequence (T+1, n) -> score of shape (1,); a batch (B, T+1, n) -> (B,) scores."""
        if policy is None:
            raise ValueError("predict needs the policy that owns the HIP engine (policy=...)")
        return policy.critic_scores(xseq, params)
