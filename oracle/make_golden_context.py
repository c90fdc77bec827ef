"""Pins oracle/context_oracle.py against the reference's own functions and writes
tests/golden/g7_context.npz.  Runs in the build container only (needs /root/reference and
scipy); the fixture holds data, not code.
  python oracle/make_golden_context.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
from scipy.spatial import KDTree                                    # noqa: E402
from src.dataset import resample_polyline, weighted_sampling       # noqa: E402  (the reference)

from oracle import context_oracle as O                              # noqa: E402


def scene(seed, npts=6000, nlines=4):
    rng = np.random.default_rng(seed)
    lines = []
    for i in range(nlines):
        x = np.sort(rng.uniform(-25, 25, 7))
        y = 3.0 * i - 4.0 + 0.3 * np.sin(x / 7.0) + rng.normal(0, 0.05, 7)
        z = rng.normal(0, 0.03, 7)
        lines.append(np.stack([x, y, z], 1))
    xyz = np.stack([rng.uniform(-27, 27, npts), rng.uniform(-7, 9, npts), rng.normal(0, 0.1, npts)], 1)
    inten = np.clip(np.round(rng.exponential(12.0, npts)), 0, 255)
    cloud = np.column_stack([xyz, inten]).astype(np.float32)
    return cloud, lines


def main():
    cloud, lines = scene(7)
    out = {"cloud": cloud}
    for i, raw in enumerate(lines):
        radius, decay, n = (0.5, 2.0, 64) if i < 3 else (0.05, 2.0, 64)     # last line: K <= N regime
        dense_ref, line_ref = resample_polyline(raw, 200), resample_polyline(raw, 32)
        dense, line = O.arc_resample(raw, 200), O.arc_resample(raw, 32)
        assert np.array_equal(dense, dense_ref) and np.array_equal(line, line_ref)
        d_ref, _ = KDTree(dense_ref).query(cloud[:, :3])
        mask_ref = d_ref < radius
        mask = O.crop_mask(cloud, dense, radius)
        assert np.array_equal(mask, mask_ref), "crop mask differs from the reference"
        cands = cloud[mask_ref]
        # weights: the reference exposes only the drawn sample; equal draws under one seed pin them
        np.random.seed(100 + i)
        samp_ref = weighted_sampling(cands, line_ref, n, decay)
        np.random.seed(100 + i)
        ctx, k = O.build_context(cloud, dense, line, radius, decay, n)
        centre = np.mean(line_ref, axis=0)
        assert k == len(cands)
        assert np.array_equal(ctx, np.hstack([samp_ref[:, :3] - centre, samp_ref[:, 3:4]])), "sample differs"
        w = O.sampling_weights(cands, line, decay) if k > n else np.zeros(0)
        out.update({f"raw{i}": raw, f"dense{i}": dense, f"line{i}": line, f"mask{i}": mask_ref,
                    f"dist{i}": d_ref, f"weights{i}": w, f"ctx{i}": ctx,
                    f"cfg{i}": np.array([radius, decay, n, 100 + i], dtype=np.float64)})
        print(f"line {i}: K={k} N={n} radius={radius}: oracle == reference")
    # empty crop: zeros minus the centre (src/dataset.py:87-88,231-232)
    far = lines[0] + np.array([0.0, 500.0, 0.0])
    ctx, k = O.build_context(cloud, O.arc_resample(far, 200), O.arc_resample(far, 32), 0.5, 2.0, 16)
    ref = weighted_sampling(np.zeros((0, 4)), resample_polyline(far, 32), 16, 2.0)
    assert k == 0 and np.array_equal(ctx[:, :3], ref[:, :3] - np.mean(resample_polyline(far, 32), axis=0))
    path = os.path.join(ROOT, "tests", "golden", "g7_context.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
