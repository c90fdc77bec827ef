"""Checkpoint file round trip (train_dist.py:233-246 saves model.state_dict() with torch.save;
inference_whole_scene.py:202-204 loads it with torch.load + load_state_dict): after FlatAdam has
re-homed every parameter as a view into one flat buffer, the saved file must still be the reference's
205-entry state_dict, loadable with a weights-only loader into a fresh model, strict, and give the
identical eval output."""
import json
import os

import pytest
import torch

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_state_dict_file_round_trip_after_flat_adam(tmp_path):
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(11)
    model = LineRefineNet().to(dev).train()
    step = TrainStep(model, None, decoder_chunk=4)            # FlatAdam: parameters become views of one buffer
    batch = synthetic_batch(8, 256, dev, seed=3)
    for _ in range(2):
        step(*batch)
    w = model.context_proj.weight           # really a view into the flat buffer by now
    assert w.untyped_storage().nbytes() > w.numel() * 4
    path = str(tmp_path / "best_model.pth")
    torch.save(model.state_dict(), path)                       # train_dist.py:235
    assert os.path.getsize(path) < 60e6                        # the flat buffer is stored once, not once per view
    sd = torch.load(path, map_location=dev, weights_only=True)
    manifest = json.load(open(os.path.join(GOLDEN, "g6_state_dict_manifest.json")))["entries"]      # the reference's own
    assert [(k, list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in sd.items()] == \
        [(k, shp, dt) for k, shp, dt in manifest]
    fresh = LineRefineNet().to(dev)
    missing, unexpected = fresh.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    ctx, noisy, _ = synthetic_batch(4, 256, dev, seed=5)
    model.eval(); fresh.eval()
    with torch.no_grad():
        a = model(ctx, noisy)
        b = fresh(ctx, noisy)
        enc_a = model.context_encoder(ctx.transpose(2, 1))
        enc_b = fresh.context_encoder(ctx.transpose(2, 1))
    # same weights, same kernels: identical outputs (every tensor of the flat buffer starts on a 16-byte
    # boundary, so the re-homed parameters take the same code paths as freshly allocated ones)
    assert torch.equal(a, b)
    assert torch.equal(enc_a[0], enc_b[0]) and torch.equal(enc_a[1], enc_b[1])
    assert all(p.data_ptr() % 16 == 0 for p in model.parameters())
    for k, v in model.state_dict().items():
        assert torch.equal(v, fresh.state_dict()[k]), k
    # the loaded model trains on: a fresh TrainStep over it steps without touching the source model
    step2 = TrainStep(fresh.train(), None, decoder_chunk=4)
    step2(*batch)
    step.close(); step2.close()
