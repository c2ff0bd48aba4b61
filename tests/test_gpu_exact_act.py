"""The LSTM activations of the critic kernels (k_lstm_fwd2 / k_lstm_bwd2) are v_exp_f32 + v_rcp_f32 (1 ulp each)
where the reference (flax OptimizedLSTMCell: jax.nn.sigmoid / tanh) evaluates them exactly.  The library is built a
second time with expf and an IEEE division (-DGMPC_LSTM2_EXACT_ACT, gan_mpc_amd/csrc/Makefile: libgan_mpc_amd_exact.so);
this test runs the critic step on the same seeded batch through both builds -- the exact one in a child process that
loads it through GMPC_LIB -- and requires (i) the two builds within 2e-6 of each other (loss, scores) / 1e-5 (summed
gradient), i.e. the approximation is far inside the parity bar, and (ii) the exact build within the suite's bars of
the oracle, like the fast one in test_gpu_parity.py::test_critic_loss_grad."""

import os
import subprocess
import sys

import numpy as np
import pytest

import gan_mpc_oracle as orc
import gpu_util as gu

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXACT = os.path.join(ROOT, "gan_mpc_amd", "libgan_mpc_amd_exact.so")

CHILD = r"""
import sys, numpy as np
sys.path[:0] = [%(root)r, %(root)r + "/oracle", %(root)r + "/tests"]
import gpu_util as gu
pb = gu.problem(17, 6, 50, 96, seed=31, head_hidden=(256, 256, 256))
eng = gu.engine_for(pb)
d = eng.to_dev
B = pb["B"]
xseq = np.concatenate([pb["true_seq"], pb["goal"]], 0)
label = np.concatenate([np.ones(B), -np.ones(B)]).astype(np.float32)
crit = d(gu.critic_flat(pb))
loss, grad = eng.critic_loss_grad(d(xseq), d(label), crit)
score, dx = eng.critic_score_vjp(d(xseq), crit)
np.savez(sys.argv[1], loss=loss.cpu().numpy(), grad=grad.cpu().numpy(), score=score.cpu().numpy(), dx=dx.cpu().numpy())
"""


def _run(tmp, lib):
    out = os.path.join(tmp, "exact.npz" if lib else "fast.npz")
    env = dict(os.environ)
    env.pop("GMPC_LIB", None)
    if lib:
        env["GMPC_LIB"] = lib
    res = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, out], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    return np.load(out)


@pytest.mark.skipif(not os.path.exists(EXACT), reason="libgan_mpc_amd_exact.so not built (make -C gan_mpc_amd/csrc)")
def test_fast_activations_against_the_exact_build(tmp_path):
    fast, exact = _run(str(tmp_path), None), _run(str(tmp_path), EXACT)
    gu.set_config("critic step, fast vs exact LSTM activations n=17 F=64 T=50 Bc=192")
    errs = {}
    for key, bar in (("loss", 2e-6), ("score", 2e-6), ("grad", 1e-5), ("dx", 1e-5)):
        e = gu.rel_err(fast[key], exact[key].astype(np.float64))
        errs[key] = e
        gu._record(dict(stage=f"critic {key}: v_exp / v_rcp build against the expf / IEEE-division build",
                        config=gu.CURRENT_CONFIG[0], e_hip=e, e_o32=None, tol=bar, tol_used=bar, branch="tol",
                        entries=int(np.asarray(fast[key]).size), passed=bool(e <= bar)))
    assert all(errs[k] <= b for k, b in (("loss", 2e-6), ("score", 2e-6), ("grad", 1e-5), ("dx", 1e-5))), errs
    assert any(errs[k] > 0 for k in errs), "the two builds returned identical bits: is the exact build the exact one?"
    # the exact build against the oracle (whole-batch sums: the suite's bars)
    pb = gu.problem(17, 6, 50, 96, seed=31, head_hidden=(256, 256, 256))
    pb64 = orc.cast_problem(pb, np.float64)
    B = pb["B"]
    xseq = np.concatenate([pb["true_seq"], pb["goal"]], 0)
    label = np.concatenate([np.ones(B), -np.ones(B)]).astype(np.float32)
    l32, g32 = orc.critic_loss_and_grad(pb["critic"], xseq, label)
    l64, g64 = orc.critic_loss_and_grad(pb64["critic"], xseq.astype(np.float64), label.astype(np.float64))
    n = 2 * B
    gu.assert_parity("exact-activation build: critic loss", exact["loss"] / n, l32, l64)
    gu.assert_parity("exact-activation build: critic grad", exact["grad"] / n, gu.pack_grads_critic(g32),
                     gu.pack_grads_critic(g64))
