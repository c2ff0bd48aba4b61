"""optax.multi_transform({tx: chain(clip_by_global_norm(100), adam(lr)), zero: set_to_zero}) as the
reference builds it (gan/runner.py:51-63, utils.get_masked_labels utils.py:159-166), on the flat
device parameter vector: one fused clip+Adam kernel over the contiguous trainable range."""

import torch


def get_masked_labels(all_vars, masked_vars, tx_key, zero_key):
    return {v: (zero_key if v in masked_vars else tx_key) for v in all_vars}


class MaskedAdam:
    def __init__(self, trainable_keys, lr, max_norm=100.0, b1=0.9, b2=0.999, eps=1e-8):
        self.trainable_keys = tuple(trainable_keys)
        self.lr, self.max_norm, self.b1, self.b2, self.eps = lr, max_norm, b1, b2, eps

    def init(self, dparams, keys=None):
        lo, cnt = dparams.range_of(keys or self.trainable_keys)
        z = torch.zeros(cnt, dtype=torch.float32, device=dparams.flat.device)
        return {"m": z, "v": z.clone(), "count": 0, "range": (lo, cnt)}

    def update(self, engine, dparams, grad, opt_state):
        """grad: flat device vector over the state's range (already the batch mean).  In place."""
        lo, cnt = opt_state["range"]
        assert grad.numel() == cnt, (grad.numel(), cnt)
        opt_state["count"] += 1
        engine.adam_clip_step(dparams.flat[lo:lo + cnt], grad.contiguous(), opt_state["m"],
                              opt_state["v"], opt_state["count"], self.lr, 1.0, self.max_norm,
                              self.b1, self.b2, self.eps)
        return dparams, opt_state


def get_optimizer(params_keys, masked_vars, lr):
    """reference gan/runner.py:51-63: returns the optimiser over every key not in masked_vars."""
    labels = get_masked_labels(params_keys, masked_vars, "tx", "zero")
    return MaskedAdam([k for k, v in labels.items() if v == "tx"], lr)
