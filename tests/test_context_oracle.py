"""Row f2 on the CPU: the context-builder oracle against the fixture generated from the
reference's own functions (oracle/make_golden_context.py), and the product's host-side polyline
resampling against the same fixture."""
import os

import numpy as np

from oracle import context_oracle as O


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "g7_context.npz"))


def test_oracle_matches_reference_fixture(golden_dir):
    g = _g(golden_dir)
    cloud = g["cloud"]
    for i in range(4):
        radius, decay, n, seed = g[f"cfg{i}"]
        dense, line = O.arc_resample(g[f"raw{i}"], 200), O.arc_resample(g[f"raw{i}"], 32)
        assert np.array_equal(dense, g[f"dense{i}"]) and np.array_equal(line, g[f"line{i}"])
        mask = O.crop_mask(cloud, dense, radius)
        assert np.array_equal(mask, g[f"mask{i}"])
        assert np.allclose(O.nearest_distance(cloud[:, :3], dense), g[f"dist{i}"], rtol=0, atol=1e-12)
        np.random.seed(int(seed))
        ctx, k = O.build_context(cloud, dense, line, radius, decay, int(n))
        assert k == int(mask.sum())
        assert np.array_equal(ctx, g[f"ctx{i}"])          # the reference's draw under the same seed
        if k > n:
            assert np.array_equal(O.sampling_weights(cloud[mask], line, decay), g[f"weights{i}"])


def test_oracle_regimes():
    rng = np.random.default_rng(0)
    cloud = np.column_stack([rng.uniform(-1, 1, (50, 3)), rng.uniform(0, 9, 50)]).astype(np.float32)
    line = np.stack([np.linspace(-1, 1, 32), np.zeros(32), np.zeros(32)], 1)
    far = line + np.array([0.0, 100.0, 0.0])
    ctx, k = O.build_context(cloud, far, far, 0.5, 2.0, 8)
    assert k == 0 and np.array_equal(ctx[:, :3], np.zeros((8, 3)) - far.mean(0)) and not ctx[:, 3].any()
    ctx, k = O.build_context(cloud, line, line, 10.0, 2.0, 64)            # K = 50 <= N: with replacement
    assert k == 50 and ctx.shape == (64, 4)
    ctx, k = O.build_context(cloud, line, line, 10.0, 2.0, 10)            # without replacement
    assert len({tuple(r) for r in ctx.round(6)}) == 10
    flat = cloud.copy(); flat[:, 3] = 7.0
    w = O.sampling_weights(flat, line, 2.0)
    assert np.allclose(w, np.exp(-O.nearest_distance(flat[:, :3], line) / 2.0))     # 0.5 + 0.5


def test_product_resampling_matches_reference(golden_dir):
    from pointnet_refine_amd.context import resample_polyline
    g = _g(golden_dir)
    for i in range(4):
        assert np.array_equal(resample_polyline(g[f"raw{i}"], 200), g[f"dense{i}"])
        assert np.array_equal(resample_polyline(g[f"raw{i}"], 32), g[f"line{i}"])
    assert np.array_equal(resample_polyline(np.zeros((1, 3)), 5), np.zeros((5, 3)))
