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


def test_sample_stream_pads_once_per_epoch_like_distributed_sampler(tmp_path):
    """the sharding of SceneSampleStream follows DistributedSampler (train_dist.py:129): the
    whole epoch's sample list is padded ONCE to a multiple of world_size (fewer than world_size
    duplicates), every rank gets the same count, and together the ranks cover every sample.  Host logic
    only - the plan is made without touching the GPU."""
    from pointnet_refine_amd.io import SceneSampleStream
    p = lambda a: [{"x": float(x), "y": float(y), "z": float(z)} for x, y, z in a]
    gt = np.array([[0, 0, 0], [1, 0.5, 0], [2, 1, 0.1]])
    counts = [1, 3, 2, 5, 1, 1, 4]                      # small scenes: 17 samples in all
    for s, n in enumerate(counts):
        items = [{"position": p(gt), "noisy_candidates": [p(gt + 0.01 * k)]} for k in range(n)]
        (tmp_path / f"s{s}.json").write_text(json.dumps({"items": items}))
        (tmp_path / f"s{s}.pcd").write_bytes(b"")       # existence is all the index needs
    world, total = 8, sum(counts)
    plans = []
    for r in range(world):
        ds = SceneSampleStream(str(tmp_path), rank=r, world_size=world, seed=5)
        assert len(ds) == -(-total // world) == 3
        plan = ds._plan()
        assert [si for si, _ in plan] == [si for si, _ in plans[0]] if plans else True     # same scene order everywhere
        plans.append(plan)
        assert sum(len(pr) for _, pr in plan) == len(ds)
    drawn = [(si, pr) for plan in plans for si, prs in plan for pr in prs]
    assert len(drawn) == 3 * world == 24                # 7 duplicates < world_size, not up to 7 per scene
    assert len(set(drawn)) == total                     # every sample of the epoch is seen
    ds = SceneSampleStream(str(tmp_path), rank=0, world_size=world, seed=5)
    ds.set_epoch(1)
    assert ds._plan() != plans[0]                       # reshuffled per epoch
