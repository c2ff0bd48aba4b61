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


def _run(tmp, tag, gpus, extra=()):
    out = os.path.join(tmp, f"{tag}.npz")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "1",
           "--windows", "1", "--no-cpu-baseline", "--secondary-maxiter", "0", "--no-large-state", "--scaling", "strong",
           "--dump-step", out, *extra]
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


def test_deferred_optimiser_step_gives_the_parameters_of_the_immediate_one(tmp_path):
    """bench.py's default ordering leaves the critic chain, its all-reduce and clip+Adam of step k on the second stream
    (the main stream goes on to step k + 1's rollout; the parameters are next read by step k + 1's critic chain, on
    that same stream; two sequence buffers) -- against --no-pipeline, which joins the optimiser step into the main
    stream at the end of every step: the same kernels on the same data in the same order per stream, so the critic
    parameters after the same optimiser steps must be BITWISE equal, with one rank and with two."""
    for gpus in (1, 2):
        ja, da = _run(str(tmp_path), f"defer{gpus}", gpus)
        jb, db = _run(str(tmp_path), f"join{gpus}", gpus, ("--no-pipeline",))
        assert "deferred" in ja["step_ordering"] and "joined" in jb["step_ordering"]
        np.testing.assert_array_equal(da["critic"], db["critic"])
        np.testing.assert_array_equal(da["mean"], db["mean"])


CHILD = r"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
rank, world, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
from gan_mpc_amd import _lib
from gan_mpc_amd.engine import Engine
torch.cuda.set_device(rank)
eng = Engine(5, 2, 8, [7, 16, 5], [5, 8, 4], max_batch=4, device=rank)
lib = eng.lib
uid = C.create_string_buffer(128)
if rank == 0:
    _lib.check(lib.gmpc_comm_unique_id(uid))
    open(idfile + ".tmp", "wb").write(uid.raw)
    os.replace(idfile + ".tmp", idfile)
else:
    for _ in range(600):
        if os.path.exists(idfile):
            break
        time.sleep(0.1)
    uid = C.create_string_buffer(open(idfile, "rb").read(), 128)
_lib.check(lib.gmpc_comm_init(eng.ctx, world, rank, uid))
buf = torch.arange(1, 20111, dtype=torch.float32, device=eng.device) * (rank + 1)
_lib.check(lib.gmpc_allreduce_grads(eng.ctx, C.c_void_p(buf.data_ptr()), buf.numel(), eng._stream()))
torch.cuda.synchronize()
np.save(out, buf.cpu().numpy())
"""


def test_c_abi_exchange_between_two_ranks(tmp_path):
    """gmpc_comm_unique_id / gmpc_comm_init / gmpc_allreduce_grads through the C ABI with a world of two: two
    processes, one device each, the id passed through a file.  RCCL refuses a communicator with two ranks on ONE device
    (ncclInvalidUsage, "duplicate GPU"), so the test needs two visible GPUs: on a one-GPU box it is skipped -- with
    that reason -- and the N > 1 arithmetic of the exchange stays covered by tests/test_parallel_gloo.py and by the
    two-rank gloo runs above; the driver's 8-GPU node runs it."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("RCCL needs one device per rank (two ranks on one GPU: ncclInvalidUsage / duplicate GPU); this box "
                    f"has {torch.cuda.device_count()}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    idfile = str(tmp_path / "nccl_id.bin")
    procs = [subprocess.Popen([sys.executable, "-c", CHILD % {"root": ROOT}, str(r), "2", idfile,
                               str(tmp_path / f"r{r}.npy")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-1500:] for o in outs)
    want = np.arange(1, 20111, dtype=np.float32) * 3.0
    for r in range(2):
        np.testing.assert_array_equal(np.load(str(tmp_path / f"r{r}.npy")), want)
