"""Row f4 (host code): the PCD reader against the outputs of the reference's loader on the same
byte streams (fixture tests/golden/g8_pcd.npz from oracle/make_golden_io.py) and the scene-JSON
reader against the reference's schema (src/dataset.py:172-187)."""
import json
import os

import numpy as np

from pointnet_refine_amd.io import load_pcd_data, load_scene_items


def test_pcd_formats_match_reference_outputs(golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "g8_pcd.npz"))
    for kind in ("ascii", "binary16", "binary14", "binary_unknown"):
        path = tmp_path / f"{kind}.pcd"
        path.write_bytes(g[kind + "_bytes"].tobytes())
        pts = load_pcd_data(str(path))
        want = g[kind + "_points"]
        assert pts.dtype == np.float32 and pts.shape == want.shape and np.array_equal(pts, want), kind
    assert g["binary_unknown_points"].shape == (0, 4)                   # 12-byte records: rejected, not raised
    assert load_pcd_data(str(tmp_path / "missing.pcd")).shape == (0, 4)  # the reference never raises here
    bad = tmp_path / "bad.pcd"
    bad.write_bytes(b"not a pcd file\n")
    assert load_pcd_data(str(bad)).shape == (0, 4)


def test_scene_json_schema(tmp_path):
    p = lambda a: [{"x": float(x), "y": float(y), "z": float(z)} for x, y, z in a]
    gt = np.array([[0, 0, 0], [1, 0.5, 0], [2, 1, 0.1]])
    scene = {"items": [{"position": p(gt), "noisy_candidates": [p(gt + 0.1), p(gt - 0.2)],
                        "context_lines": [p(gt + 3), []]},
                       {"noisy_candidates": [p(gt)]}]}
    path = tmp_path / "scene.json"
    path.write_text(json.dumps(scene))
    items = load_scene_items(str(path))
    assert len(items) == 2
    assert np.array_equal(items[0]["position"], gt) and items[0]["position"].dtype == np.float64
    assert len(items[0]["noisy_candidates"]) == 2 and np.allclose(items[0]["noisy_candidates"][1], gt - 0.2)
    assert len(items[0]["context_lines"]) == 1                          # empty lines dropped (:183-186)
    assert items[1]["position"] is None and items[1]["context_lines"] == []
