"""Property tests of the host-side pieces (hypothesis): window cutting, the replay FIFO, trajectory
sharding over ranks, parameter packing, Config round trips."""

import numpy as np
from hypothesis import given, settings, strategies as st

from gan_mpc_amd import data_buffers, data_normalizer, parallel, params as P
from gan_mpc_amd.config import load_config


@settings(max_examples=60, deadline=None)
@given(L=st.integers(1, 40), width=st.integers(1, 3), length=st.integers(1, 12), start=st.integers(0, 5),
       count=st.integers(-2, 30))
def test_sliding_windows_equal_python_slices(L, width, length, start, count):
    traj = np.arange(L * width, dtype=np.float64).reshape(L, width)
    count = min(count, L - length - start + 1)
    got = data_buffers.sliding_windows(traj, length, count, start=start)
    want = [traj[start + i:start + i + length] for i in range(max(count, 0))]
    assert got.shape == (max(count, 0), length, width)
    for g, w in zip(got, want):
        np.testing.assert_array_equal(g, w)
    if len(got):
        got[0, 0, 0] = -1.0                      # a copy: the trajectory is untouched
        assert traj[start, 0] != -1.0


@settings(max_examples=40, deadline=None)
@given(horizon=st.integers(1, 6), maxlen=st.integers(1, 25),
       lens=st.lists(st.integers(0, 15), min_size=1, max_size=6))
def test_replay_buffer_is_a_fifo_of_windows(horizon, maxlen, lens):
    ident = data_normalizer.JointNormalizer(data_normalizer.IdentityNormalizer(),
                                            data_normalizer.IdentityNormalizer())
    rb = data_buffers.ReplayBuffer(horizon, maxlen, ident)
    all_s, all_n = [], []
    offset = 0.0
    for L in lens:
        s = (offset + np.arange(L, dtype=np.float64))[:, None] * np.ones((1, 2))
        a = -s[:, :1]
        offset += 100.0
        rb.add(s, a)
        for i in range(L - horizon):
            all_s.append(s[i:i + horizon])
            all_n.append(s[i + 1:i + 1 + horizon])
    S, A, N = rb.get_dataset()
    keep_s, keep_n = all_s[-maxlen:], all_n[-maxlen:]
    assert len(rb) == len(keep_s)
    if keep_s:
        np.testing.assert_array_equal(S, np.array(keep_s))
        np.testing.assert_array_equal(N, np.array(keep_n))
        np.testing.assert_array_equal(A, -S[..., :1])
    else:
        assert S.shape == (0,)


@settings(max_examples=80, deadline=None)
@given(count=st.integers(0, 5000), world=st.integers(1, 16))
def test_shards_tile_the_batch(count, world):
    ranges = [parallel.shard_range(count, r, world) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == count
    for (lo, hi), (lo2, _) in zip(ranges, ranges[1:]):
        assert hi == lo2 and hi >= lo
    sizes = [hi - lo for lo, hi in ranges]
    assert max(sizes) - min(sizes) <= 1


@settings(max_examples=30, deadline=None)
@given(dims=st.lists(st.integers(1, 9), min_size=2, max_size=5), seed=st.integers(0, 10_000))
def test_mlp_packing_round_trip(dims, seed):
    rng = np.random.default_rng(seed)
    layers = [(rng.standard_normal((a, b)).astype(np.float32), rng.standard_normal(b).astype(np.float32))
              for a, b in zip(dims[:-1], dims[1:])]
    tree = P.layers_to_tree(layers)
    flat = P.pack_mlp(tree)
    assert flat.size == sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))
    assert P.mlp_dims(tree) == dims
    back = P.tree_to_layers(P.unpack_mlp(flat, dims))
    for (W, b), (W2, b2) in zip(layers, back):
        np.testing.assert_array_equal(W, W2)
        np.testing.assert_array_equal(b, b2)


_leaf = st.one_of(st.integers(-5, 5), st.floats(-1, 1, allow_nan=False), st.text(max_size=4),
                  st.lists(st.integers(0, 3), max_size=3))
_key = st.text(alphabet="abcdefgh_", min_size=1, max_size=5)


@settings(max_examples=50, deadline=None)
@given(tree=st.recursive(st.dictionaries(_key, _leaf, max_size=4),
                         lambda kids: st.dictionaries(_key, st.one_of(_leaf, kids), max_size=4), max_leaves=12))
def test_config_round_trip(tree):
    cfg = load_config.Config.from_dict(tree)
    assert cfg.to_dict() == tree
    for k, v in tree.items():
        got = getattr(cfg, k)
        assert (got.to_dict() if isinstance(v, dict) else got) == v
