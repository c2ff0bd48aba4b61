"""L2 learnable MPC policy (reference norm/l2_policy.py:12-18): loss = sum_dims mean_t (x - x*)^2,
evaluated by the k_l2loss kernel (loss_kind 0)."""

from gan_mpc_amd.policy import base


class L2MPC(base.BaseMPC):
    LOSS_KIND = 0

    def loss(self, xcseq, useq, params, desired_xseq):
        del useq, params
        eng = self._engine
        if eng is None:
            raise RuntimeError("loss() needs a bound engine: call get_optimal_values/loss_and_grad")
        import torch
        d = (xcseq[..., : desired_xseq.shape[-1]] - torch.as_tensor(desired_xseq, device=xcseq.device)) ** 2
        return d.mean(dim=-2).sum(dim=-1)
