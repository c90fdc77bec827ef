"""The fused eval-mode encoder kernel (csrc/prh_fused.hpp):
context -> memory (+ fused, global_feat) in one launch with BatchNorm folded.
  precision "fp32" (two fp16 planes, three products): the north_star's 1e-4 against the golden
      vectors generated from the reference (G1) and against the oracle on ragged shapes;
  precision "fp16" (one plane, BASELINE config 5): the reduced-precision gate 5e-2 on `out`."""
import os

import numpy as np
import pytest
import torch

from conftest import maxdiff, rel_l2
from oracle import linerefine_oracle as O
from oracle import procedural as P

pytestmark = pytest.mark.gpu


def _model(sd=None):
    from pointnet_refine_amd.model import LineRefineNet
    m = LineRefineNet()
    if sd is not None:
        m.load_state_dict(sd, strict=True)
    return m.cuda().eval()


def _spy():
    """Count launches of the fused entry point through the ops layer."""
    from pointnet_refine_amd import ops
    calls = []
    orig = ops.encoder_eval_fused

    def wrapped(*a, **k):
        calls.append(k.get("precision", "fp32"))
        return orig(*a, **k)
    return ops, orig, wrapped, calls


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("fp16", 5e-2), (None, 1e-4)])
def test_g1_through_the_fused_kernel(golden_dir, precision, tol):
    g = np.load(os.path.join(golden_dir, "g1_eval_forward.npz"))
    m = _model(P.linerefine_state_dict(0))
    m.context_encoder.inference_precision = precision
    ctx, noisy, _ = P.synth_batch(8, 256, 4, 32, seed=1234)
    ops, orig, wrapped, calls = _spy()
    ops.encoder_eval_fused = wrapped
    try:
        with torch.no_grad():
            out = m(ctx.cuda(), noisy.cuda())
            memory = m.encode_context(ctx.cuda())
            gf, fu = m.context_encoder(ctx.cuda().transpose(2, 1))
    finally:
        ops.encoder_eval_fused = orig
    assert (len(calls) == 3 and set(calls) == {precision}) if precision else calls == []
    assert maxdiff(out, g["out"]) < tol
    if precision == "fp16":
        assert rel_l2(g["memory_sub"], memory[:, ::16, ::8]) < 1e-2
        assert rel_l2(g["global_feat"], gf) < 1e-2
        assert rel_l2(g["fused_sub"], fu.transpose(2, 1)[:, ::16, ::8]) < 1e-2
    else:
        assert maxdiff(memory[:, ::16, ::8], g["memory_sub"]) < 1e-4
        assert maxdiff(gf, g["global_feat"]) < 1e-4
        assert maxdiff(fu.transpose(2, 1)[:, ::16, ::8], g["fused_sub"]) < 1e-4


@pytest.mark.parametrize("B,N,C", [(1, 1, 4), (3, 7, 4), (2, 129, 4), (5, 1000, 4), (3, 160, 6), (2, 64, 4), (4, 33, 4)])
def test_fused_encoder_ragged_shapes_vs_oracle(B, N, C):
    """Tiles that end inside a segment (N not a multiple of 32 / 64), one-point segments, the C = 6
    encoder variant: the fp32-accurate kernel against the oracle, both returns."""
    from pointnet_refine_amd.model import MultiScalePointNetEncoder
    sd = P.encoder_state_dict(C, 1024, seed=3)
    sd["fusion.1.weight"][5] = 0.0          # a dead channel: exact zeros through the gate and the pooling
    sd["fusion.1.bias"][5] = -1.0
    m = MultiScalePointNetEncoder(in_channel=C, out_dim=1024)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    ctx, _, _ = P.synth_batch(B, N, C, 32, seed=B * 100 + N)
    o_g, o_f = O.encoder_forward(O.as_params(sd), ctx, "", False)
    with torch.no_grad():
        for prec, tol in (("fp32", 1e-4), ("fp16", None)):
            m.inference_precision = prec
            gf, fu = m(ctx.cuda().transpose(2, 1))
            assert gf.shape == (B, 2048) and fu.shape == (B, 1024, N)
            if tol is not None:
                assert maxdiff(gf, o_g) < tol and maxdiff(fu.transpose(2, 1), o_f) < tol
                assert float(fu[:, 5].abs().max()) == 0.0
            else:
                assert rel_l2(o_f, fu.transpose(2, 1)) < 1e-2 and rel_l2(o_g, gf) < 1e-2


def test_fused_kernel_matches_the_per_layer_path_and_tracks_weight_updates():
    """Same weights, 16 x 1000 points: the fused kernel (fp32-accurate) and the per-layer eval
    kernels agree to 1e-4 on memory; after an in-place weight change (optimizer-style) the cached
    image is rebuilt - stale weights would reproduce the old output."""
    m = _model(P.linerefine_state_dict(1))
    ctx, noisy, _ = P.synth_batch(16, 1000, 4, 32, seed=5)
    ctx = ctx.cuda()
    with torch.no_grad():
        m.context_encoder.inference_precision = "fp32"
        a = m.encode_context(ctx)
        m.context_encoder.inference_precision = None
        b = m.encode_context(ctx)
        assert maxdiff(a, b) < 1e-4
        m.context_encoder.inference_precision = "fp32"
        m.context_encoder.fusion[0].weight.mul_(1.01)
        m.context_encoder.bn3.running_var.mul_(1.1)
        c = m.encode_context(ctx)
        m.context_encoder.inference_precision = None
        d = m.encode_context(ctx)
    assert maxdiff(c, d) < 1e-4 and maxdiff(a, c) > 1e-3


def test_training_and_grad_mode_never_take_the_fused_kernel():
    ops, orig, wrapped, calls = _spy()
    ops.encoder_eval_fused = wrapped
    try:
        m = _model(P.linerefine_state_dict(0))
        ctx, noisy, _ = P.synth_batch(2, 64, 4, 32, seed=5)
        out = m(ctx.cuda(), noisy.cuda())               # eval mode, but autograd is on: per-layer path (it has a backward)
        out.sum().backward()
        m.train()
        with torch.no_grad():
            m(ctx.cuda(), noisy.cuda())
    finally:
        ops.encoder_eval_fused = orig
    assert calls == []
    assert m.context_encoder.conv1.weight.grad is not None


@pytest.mark.parametrize("precision", ["fp32", "fp16"])
def test_activations_beyond_the_fp16_range_are_flagged_not_clamped(precision):
    """the fused kernel carries activations between layers as fp16 planes and
    clamps at 65504 - fine for a trained checkpoint (post-BatchNorm values are O(1)), silent garbage
    for weights that are not.  bn1.weight = 3e5 puts h1 at ~1e5..1e6: the kernel counts the clamps,
    ops.encoder_eval_fused raises FusedSaturation, and the modules redo the call on the per-layer
    kernels (fp32 storage) with a warning - same numbers as inference_precision=None and as the oracle."""
    from pointnet_refine_amd import ops
    sd = P.linerefine_state_dict(0)
    sd["context_encoder.bn1.weight"] = sd["context_encoder.bn1.weight"] * 3e5
    m = _model(sd)
    enc = m.context_encoder
    ctx, noisy, _ = P.synth_batch(4, 200, 4, 32, seed=8)
    c = ctx.cuda()
    with torch.no_grad():
        enc.inference_precision = None
        gf_ref, fu_ref = enc(c.transpose(2, 1))
        out_ref = m(c, noisy.cuda())
        enc.inference_precision = precision
        with pytest.raises(ops.FusedSaturation):
            ops.encoder_eval_fused(c, enc._param_list(), enc._bn_buffer_list(), enc.bn1.eps, want_fused=True,
                                   want_global=True, precision=precision)
        assert ops.fused_saturation(c.device) == 0          # the raise consumed the counter
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            gf, fu = enc(c.transpose(2, 1))
        with pytest.warns(RuntimeWarning, match="fp16 range"):
            out = m(c, noisy.cuda())
    assert torch.equal(gf, gf_ref) and torch.equal(fu, fu_ref) and torch.equal(out, out_ref)
    g64, f64 = O.encoder_forward(O.as_params(sd, dtype=torch.float64), ctx.double(), "context_encoder.", False)
    assert rel_l2(f64, fu.transpose(2, 1)) < 1e-5 and rel_l2(g64, gf) < 1e-5
    # ordinary weights: nothing flagged, no warning
    m2 = _model(P.linerefine_state_dict(0))
    m2.context_encoder.inference_precision = precision
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            m2(c, noisy.cuda())
    assert ops.fused_saturation(c.device) == 0


def test_image_cache_sees_writes_through_data_and_raw_pointers():
    """`p.data.copy_()` and raw-pointer writers do not bump a parameter's version
    counter; the image cache key carries a device-side fingerprint of the values, so the next eval
    forward rebuilds the folded image."""
    m = _model(P.linerefine_state_dict(0))
    ctx, _, _ = P.synth_batch(2, 128, 4, 32, seed=4)
    c = ctx.cuda()
    with torch.no_grad():
        a = m.encode_context(c).clone()
        w = m.context_encoder.conv3.weight
        v0 = w._version
        w.data.mul_(1.5)                                   # no version bump
        assert w._version == v0
        b = m.encode_context(c).clone()
        m.context_encoder.inference_precision = None
        ref = m.encode_context(c)
    assert maxdiff(a, b) > 1e-3
    assert maxdiff(b, ref) < 1e-4
