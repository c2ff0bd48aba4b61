"""The two CPU restatements pin each other: oracle/gan_mpc_step.c (plain C + OpenMP, fp32 -- what bench.py's
cpu_baseline leg times) against oracle/gan_mpc_oracle.py (NumPy, fp64 as arbiter) on the same seeded inputs."""

import os

import numpy as np
import pytest

import gan_mpc_oracle as orc
import gan_mpc_step_c as oc

pytestmark = pytest.mark.skipif(not os.path.exists(oc.LIB_PATH), reason="oracle/libgan_mpc_step.so not built")


def _rel(a, ref):
    return float(np.abs(np.asarray(a, np.float64) - ref).max() / (np.abs(ref).max() + 1e-300))


@pytest.mark.parametrize("shape", [(5, 2, 8, 7, (33, 47), (24,), 6, 1.0), (17, 6, 20, 6, (200, 200, 200), (128, 128), 10, 0.1),
                                   (9, 3, 6, 4, (40,), (16,), 4, 0.5)])
def test_c_step_matches_numpy_oracle(shape):
    n, m, T, B, dh, ch, f, scale = shape
    pb = orc.make_problem(n, m, T, B, seed=4, dyn_hidden=dh, cost_hidden=ch, cost_fout=f, bias_scale=0.1)
    W, b = pb["dyn"][-1]
    pb["dyn"][-1] = ((W * scale).astype(np.float32), (b * scale).astype(np.float32))
    p64 = orc.cast_problem(pb, np.float64)
    out = oc.trajectories(pb["dyn"], pb["cmlp"], pb["mpc_w"], pb["goal"], pb["x0"], pb["U"])
    X64 = orc.rollout(p64["dyn"], p64["U"], p64["x0"])
    assert _rel(out["X"], X64) < 5e-6
    assert _rel(out["costs"], orc.evaluate(p64["cmlp"], p64["mpc_w"], p64["goal"], X64, p64["U"])) < 5e-6
    # backward pass at the trajectory the C code saw
    Xc = out["X"].astype(np.float64)
    lqr = orc.get_lqr_params(p64["dyn"], p64["cmlp"], p64["mpc_w"], p64["goal"], Xc, p64["U"])
    lqr32 = orc.get_lqr_params(pb["dyn"], pb["cmlp"], pb["mpc_w"], pb["goal"], out["X"], pb["U"])
    K64, k64, _, _ = orc.tvlqr(*lqr)
    K32, k32, _, _ = orc.tvlqr(*lqr32)
    g64, a64 = orc.adjoint(lqr[5], lqr[6], lqr[1], lqr[3])
    assert _rel(out["grad"], g64) < 1e-5 and _rel(out["adjoints"], a64) < 1e-5
    # the gains carry cond(G) ~ 1e4 (tests/gpu_util.py GAIN_CEILING): no worse than 4x the NumPy fp32 oracle
    assert _rel(out["K"], K64) <= max(1e-5, 4 * _rel(K32, K64))
    assert _rel(out["k"], k64) <= max(1e-5, 4 * _rel(k32, k64))


@pytest.mark.parametrize("head", [(), (12,), (32, 16)])
def test_c_critic_matches_numpy_oracle(head):
    pb = orc.make_problem(5, 2, 8, 9, seed=6, lstm_features=16, head_hidden=head, bias_scale=0.2)
    p64 = orc.cast_problem(pb, np.float64)
    xseq = np.concatenate([pb["true_seq"], pb["goal"]], 0)
    label = np.where(np.arange(len(xseq)) % 3 == 0, -1.0, 1.0).astype(np.float32)
    flat = oc.critic_flat(pb["critic"])
    hd = [16] + list(head) + [1]
    ls, gs = oc.critic_loss_grad(flat, 5, 16, hd, xseq, label)
    l64, g64 = orc.critic_loss_and_grad(p64["critic"], xseq.astype(np.float64), label.astype(np.float64))
    g64 = np.concatenate([g64["Wx"].ravel(), g64["Wh"].ravel(), g64["b"].ravel()]
                         + [t.ravel() for Wb in g64["head"] for t in Wb])
    assert abs(ls / len(xseq) - l64) < 1e-6 * abs(l64)
    assert _rel(gs / len(xseq), g64) < 1e-5


def test_c_adam_clip_matches_numpy_oracle():
    rng = np.random.default_rng(1)
    p = rng.standard_normal(500).astype(np.float32)
    m, v = np.zeros(500, np.float32), np.zeros(500, np.float32)
    p64, m64, v64 = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in (1, 2, 3):
        g = (rng.standard_normal(500) * (1e3 if step == 2 else 1.0)).astype(np.float32)
        oc.adam_clip(p, g, m, v, 0.5, step, 1e-3)
        p64, m64, v64 = orc.adam_clip_step(p64, 0.5 * g.astype(np.float64), m64, v64, step, 1e-3)
        assert _rel(p, p64) < 1e-6 and _rel(m, m64) < 1e-6 and _rel(v, v64) < 1e-6


def test_c_step_under_address_and_ub_sanitizers(tmp_path):
    """The C restatement built with -fsanitize=address,undefined and driven through the tests above in a child
    process (the sanitizer runtime has to be preloaded before Python): no out-of-bounds access, no undefined
    behaviour on the seeded shapes.  (GPU sanitizers are not available on the pool; this is the CPU side.)"""
    import shutil
    import subprocess
    import sys
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not asan or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = str(tmp_path / "libgan_mpc_step_asan.so")
    subprocess.check_call([gcc, "-O1", "-g", "-fPIC", "-fopenmp", "-mavx2", "-mfma", "-std=c11", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-shared", "-o", lib,
                           os.path.join(root, "oracle", "gan_mpc_step.c"), "-lm"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", OMP_NUM_THREADS="2",
               GMPC_STEP_LIB=lib)
    code = ("import sys, gan_mpc_step_c as oc; assert oc.LIB_PATH.endswith('_asan.so'), oc.LIB_PATH; "
            "import pytest; sys.exit(pytest.main(['-x', '-q', '-p', 'no:cacheprovider', %r, '-k', 'not sanitizers']))"
            % os.path.abspath(__file__))
    r = subprocess.run([sys.executable, "-c", code], env=dict(env, PYTHONPATH=os.path.join(root, "oracle")),
                       capture_output=True, text=True, cwd=root, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "passed" in r.stdout
