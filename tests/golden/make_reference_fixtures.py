#!/usr/bin/env python3
"""Golden vectors from the REFERENCE's own host modules (run in the build container only).

The three reference modules that need nothing but numpy / yaml -- data_normalizer.py,
data_buffers.py, config/load_config.py -- are loaded by file path from /root/reference and driven
with seeded synthetic inputs; inputs and outputs are stored in reference_host_fixtures.npz /
reference_config_fixture.json.  The reference never travels: tests read only these data files.

    python tests/golden/make_reference_fixtures.py
"""
import contextlib
import importlib.util
import io
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    dn = load("ref_data_normalizer", "data_normalizer.py")
    db = load("ref_data_buffers", "data_buffers.py")
    lc = load("ref_load_config", "config/load_config.py")
    rng = np.random.default_rng(20261004)
    out = {}
    # ---- normalisers
    states = rng.normal(2.0, 3.0, (4, 30, 5))
    actions = np.tanh(rng.normal(size=(4, 30, 2)))
    out["states"], out["actions"] = states, actions
    with contextlib.redirect_stdout(io.StringIO()):
        std = dn.StandardNormalizer()
        std.update(states)
        joint = dn.JointNormalizer(dn.StandardNormalizer(), dn.IdentityNormalizer())
        joint.update(state_dataset=states, action_dataset=actions)
    out["std_mean"], out["std_std"] = std.mean, std.std
    out["std_normalized"] = std.normalize(states)
    ns, na = joint.normalize(states, actions)
    out["joint_states"], out["joint_actions"] = ns, na
    out["identity"] = dn.IdentityNormalizer().normalize(states.tolist())
    # ---- history buffer: more appends than it holds
    buf = db.Buffer(maxlen=6, normalizer=joint)
    xs, us = rng.normal(size=(11, 5)), rng.normal(size=(10, 2))
    for i in range(10):
        buf.append_state(xs[i])
        buf.append_action(us[i])
    buf.append_state(xs[10])
    out["buf_x_in"], out["buf_u_in"] = xs, us
    out["buf_states"], out["buf_actions"] = buf.get_state_data(), buf.get_action_data()
    # ---- replay buffer: windows + FIFO truncation
    rb = db.ReplayBuffer(horizon=7, q_maxlen=30, normalizer=joint)
    trajs_s = [rng.normal(size=(L, 5)) for L in (20, 9, 31)]
    trajs_a = [rng.normal(size=(L, 2)) for L in (20, 9, 31)]
    for k, (s, a) in enumerate(zip(trajs_s, trajs_a)):
        out[f"rb_s{k}"], out[f"rb_a{k}"] = s, a
        rb.add(s, a)
    d = rb.get_dataset()
    out["rb_states"], out["rb_actions"], out["rb_next"] = d
    w = rb.from_traj_to_seq(trajs_s[0], trajs_a[0])
    out["rb_win_states"], out["rb_win_actions"], out["rb_win_next"] = w
    np.savez_compressed(os.path.join(HERE, "reference_host_fixtures.npz"), **out)
    # ---- config: the reference's Config on THIS repository's yaml
    cfg = lc.Config.from_yaml(os.path.join(HERE, "mirror_config.yaml"))
    with open(os.path.join(HERE, "reference_config_fixture.json"), "w") as fp:
        json.dump({"to_dict": cfg.to_dict(),
                   "probe": {"mpc.horizon": cfg.mpc.horizon,
                             "mpc.model.cost.mlp.num_hidden_units": cfg.mpc.model.cost.mlp.num_hidden_units}},
                  fp, indent=1, sort_keys=True)
    print("wrote reference_host_fixtures.npz, reference_config_fixture.json")


if __name__ == "__main__":
    main()
