"""Cost model (reference cost/cost_model.py:11-42).  The arithmetic of get_cost -- sigmoid weights,
smooth-L2 stage cost, terminal |MLP(x)|^2, where(t == H, terminal, stage) -- runs inside the HIP
kernels (gmpc_rollout_cost and the backward pass); this object carries the configuration and the
parameter initialiser, and exposes get_cost for single samples through the same kernels."""

import numpy as np

from gan_mpc_amd import base


class MujocoBasedModel(base.BaseCostModel):
    def __init__(self, config, model):
        super().__init__(config)
        self.model = model

    def init(self, *args):
        model_args = self.model.get_init_params(*args)
        return self.model.init(*model_args)

    def get_cost(self, xc, u, t, params, weights, goal_X, policy=None):
        """Per-step cost of one (xc, u, t) as the reference computes it.  Evaluated by the rollout
        kernel on a one-trajectory batch whose step t is pinned to (xc, u); `policy` supplies the
        engine (dynamics parameters do not matter for a stage cost; for t == H the kernel's terminal
        branch needs x_H == xc, which holds when xc is the state the policy's rollout reaches)."""
        if policy is None:
            raise ValueError("get_cost needs the policy that owns the HIP engine (policy=...)")
        return policy.single_cost(xc, u, int(t), params, weights, np.asarray(goal_X))
