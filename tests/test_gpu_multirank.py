"""The multi-rank path of bench.py's own step() on the GPU box: two ranks (gloo -- the box has one GPU, both
ranks share it) on the strong-scaling split of the 1024-trajectory batch against one rank on the whole batch.
Same launcher, same sharding, same exchange (gan_mpc_amd/parallel.py) and same stream ordering as the driver's
`python bench.py --gpus N` run over RCCL; bench.py itself checks that the parameter replicas stay bitwise identical
on every multi-rank run."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(tmp, tag, gpus):
    out = os.path.join(tmp, f"{tag}.npz")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
           "--windows", "1", "--no-cpu-baseline", "--secondary-maxiter", "0", "--scaling", "strong",
           "--dump-step", out]
    if gpus > 1:
        cmd += ["--backend", "gloo"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [ln for ln in res.stdout.splitlines() if ln.startswith('{"metric"')][-1]
    return json.loads(line), np.load(out)


def test_two_ranks_take_the_step_of_one(tmp_path):
    j1, d1 = _run(str(tmp_path), "one", 1)
    j2, d2 = _run(str(tmp_path), "two", 2)
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["scaling"] == "strong"
    assert int(d1["global_batch"]) == int(d2["global_batch"]) == 1024
    # N-rank mean against the 1-rank mean (1e-6; only the order of the sums differs)
    m1, m2 = d1["mean"].astype(np.float64), d2["mean"].astype(np.float64)
    assert abs(m1[0] - m2[0]) <= 1e-6 * abs(m1[0])                                   # loss
    assert np.abs(m1[1:] - m2[1:]).max() <= 1e-6 * np.abs(m1[1:]).max()              # gradient
    # and the parameters after the same four optimiser steps
    assert np.abs(d1["critic"] - d2["critic"]).max() <= 1e-6 * np.abs(d1["critic"]).max()
