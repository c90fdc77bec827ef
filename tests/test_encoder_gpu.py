"""HIP encoder / linear / point_mlp (through the nn.Module surface -> C ABI) against the
oracle on the same seeded inputs and against the golden vectors.

Tolerances: forward max-abs <= 1e-4 (north_star); gradients rel-L2 <= 5e-3 per tensor
with median <= 1e-3 (ReLU/max mask flips inside fp32 noise make max-abs on gradients
meaningless, see oracle/make_golden.py); BN running statistics rel 1e-5.
"""
import os
import re

import numpy as np
import pytest
import torch

from conftest import maxdiff, rel_l2
from oracle import linerefine_oracle as O
from oracle import procedural as P

pytestmark = pytest.mark.gpu


def _pre_bn_bias(k):
    return bool(re.search(r"(conv\d\.bias|fusion\.0\.bias|point_mlp\.[036]\.bias)$", k))


def _encoder(C, sd):
    from pointnet_refine_amd.model import MultiScalePointNetEncoder
    m = MultiScalePointNetEncoder(in_channel=C, out_dim=1024)
    m.load_state_dict(sd, strict=True)
    return m.cuda()


@pytest.mark.parametrize("name,C,B,N", [("g3_encoder_c4_train", 4, 4, 192), ("g4_encoder_c6_train", 6, 3, 160)])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_encoder_vs_oracle_and_golden(golden_dir, name, C, B, N, mode):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = P.encoder_state_dict(C, 1024, seed=3)
    sd["fusion.1.weight"][5] = 0.0     # dead channel: max-pool tie -> first index
    sd["fusion.1.bias"][5] = -1.0
    ctx, _, _ = P.synth_batch(B, N, C, 32, seed=77)
    r = np.random.default_rng(5)
    up_g = torch.from_numpy(r.normal(0, 1, (B, 2048)).astype(np.float32))
    up_f = torch.from_numpy(r.normal(0, 1, (B, N, 1024)).astype(np.float32))

    m = _encoder(C, sd)
    m.train(mode == "train")
    x = ctx.cuda().requires_grad_(True)
    gf, fu_cm = m(x.transpose(2, 1))                 # reference contract: (B,C,N) in, (B,out,N) out
    assert fu_cm.shape == (B, 1024, N) and gf.shape == (B, 2048)
    ((gf * up_g.cuda()).sum() + (fu_cm.transpose(2, 1) * up_f.cuda()).sum()).backward()

    p = O.as_params(sd, requires_grad=True)
    ox = ctx.clone().requires_grad_(True)
    ns = {}
    o_g, o_f = O.encoder_forward(p, ox, "", mode == "train", ns)
    ((o_g * up_g).sum() + (o_f * up_f).sum()).backward()

    # forward: oracle and golden
    assert maxdiff(gf, o_g) < 1e-4 and maxdiff(fu_cm.transpose(2, 1), o_f) < 1e-4
    assert maxdiff(gf, g[f"{mode}::gfeat"]) < 1e-4
    assert maxdiff(fu_cm.transpose(2, 1)[:, ::8, ::16], g[f"{mode}::fused_sub"]) < 1e-4
    # input gradient
    assert rel_l2(ox.grad, x.grad) < 5e-3
    assert rel_l2(g[f"{mode}::dx"], x.grad) < 5e-3
    # parameter gradients
    rels = {}
    named = dict(m.named_parameters())
    for k, nrm in zip(g[f"{mode}::grad_keys"], g[f"{mode}::grad_norms"]):
        k = str(k)
        if mode == "train" and _pre_bn_bias(k):
            assert float(named[k].grad.abs().max()) < 1e-3     # analytically zero
            continue
        rels[k] = rel_l2(p[k].grad.reshape(named[k].shape), named[k].grad)
        assert abs(float(named[k].grad.double().norm()) - nrm) <= 5e-3 * nrm + 1e-9, k
        assert rel_l2(g[f"{mode}::gh::{k}"], named[k].grad.reshape(-1)[:64]) < 2e-2, k
    # g4/train: the reference-vs-oracle comparison itself shows 2.2e-3 here (one max-pool
    # arg-max / ReLU flip inside fp32 noise moves every upstream gradient), so the gate is
    # the flip level, not the 1e-6 level the other three cases reach.
    assert max(rels.values()) < 1e-2, max(rels, key=rels.get)
    assert float(np.median(list(rels.values()))) < 5e-3
    # running statistics
    msd = m.state_dict()
    if mode == "train":
        for k, v in ns.items():
            assert maxdiff(msd[k], v) <= 1e-5 * float(v.double().abs().max()) + 1e-6, k
            assert maxdiff(msd[k], g["st::" + k]) <= 1e-5 * float(v.double().abs().max()) + 1e-6, k
    else:
        for k, v in sd.items():
            if "running" in k or "num_batches" in k:
                assert torch.equal(msd[k].cpu(), v), k


def test_encoder_backward_twice_raises():
    sd = P.encoder_state_dict(4, 1024, seed=3)
    m = _encoder(4, sd).train()
    ctx, _, _ = P.synth_batch(2, 64, 4, 32, seed=1)
    gf, fu = m(ctx.cuda().transpose(2, 1))
    loss = fu.sum() + gf.sum()
    loss.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="second"):
        loss.backward()


def test_encoder_backward_twice_with_the_retain_switch():
    """The reference allows loss.backward(retain_graph=True) twice (src/model.py is stock autograd);
    ops.allow_encoder_retain_graph(True) makes the backward work on a copy of the one buffer it overwrites:
    the second backward then ADDS the same gradients (autograd accumulation), as with the reference."""
    from pointnet_refine_amd import ops
    sd = P.encoder_state_dict(4, 1024, seed=3)
    m = _encoder(4, sd).train()
    ctx, _, _ = P.synth_batch(2, 64, 4, 32, seed=1)
    ops.allow_encoder_retain_graph(True)
    try:
        x = ctx.cuda().transpose(2, 1).contiguous().requires_grad_(True)
        gf, fu = m(x)
        loss = (fu * fu).sum() + gf.sum()
        loss.backward(retain_graph=True)
        g1 = {k: v.grad.clone() for k, v in m.named_parameters()}
        dx1 = x.grad.clone()
        loss.backward()
        for k, v in m.named_parameters():
            assert maxdiff(v.grad, 2 * g1[k]) <= 1e-6 * float(g1[k].abs().max()) + 1e-12, k
        assert maxdiff(x.grad, 2 * dx1) <= 1e-6 * float(dx1.abs().max())
    finally:
        ops.allow_encoder_retain_graph(False)


def test_encoder_channel_mismatch_raises():
    m = _encoder(4, P.encoder_state_dict(4, 1024, seed=3)).eval()
    with pytest.raises(RuntimeError, match="channels"):
        m(torch.randn(2, 3, 64, device="cuda"))
    with pytest.raises(ValueError, match="more than 1 value"):
        m.train()(torch.randn(1, 4, 1, device="cuda"))


@pytest.mark.parametrize("B,N", [(1, 1), (3, 7), (2, 129), (5, 1000)])
def test_encoder_ragged_sizes_eval(B, N):
    sd = P.encoder_state_dict(4, 1024, seed=3)
    m = _encoder(4, sd).eval()
    ctx, _, _ = P.synth_batch(B, N, 4, 32, seed=B * 100 + N)
    with torch.no_grad():
        gf, fu = m(ctx.cuda().transpose(2, 1))
        o_g, o_f = O.encoder_forward(O.as_params(sd), ctx, "", False)
    assert maxdiff(gf, o_g) < 1e-4 and maxdiff(fu.transpose(2, 1), o_f) < 1e-4


def test_point_mlp_c3_golden(golden_dir):
    from pointnet_refine_amd.model import LineRefineNet
    g = np.load(os.path.join(golden_dir, "g5_point_mlp_c3.npz"))
    sd = P.linerefine_state_dict(0)
    m = LineRefineNet()
    m.load_state_dict(sd, strict=True)
    m.cuda()
    r = np.random.default_rng(9)
    x = torch.from_numpy(r.normal(0, 1.5, (4, 1024, 3)).astype(np.float32))
    for mode in ("train", "eval"):
        m.load_state_dict(sd, strict=True)
        m.train(mode == "train")
        with torch.no_grad():
            y = m.encode_line(x.cuda())
        assert maxdiff(y[:, ::16, ::2], g[mode]) < 1e-4


def test_linear_fwd_bwd():
    from pointnet_refine_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(5, 77, 1024, generator=g)
    w = torch.randn(256, 1024, generator=g) / 32
    b = torch.randn(256, generator=g)
    up = torch.randn(5, 77, 256, generator=g)
    xs = [t.clone().requires_grad_(True) for t in (x, w, b)]
    (torch.nn.functional.linear(*[t.double() for t in xs]) * up.double()).sum().backward()
    ys = [t.cuda().requires_grad_(True) for t in (x, w, b)]
    out = ops.linear(*ys)
    (out * up.cuda()).sum().backward()
    assert maxdiff(out, torch.nn.functional.linear(x.double(), w.double(), b.double())) < 1e-4
    for a, bb in zip(xs, ys):
        assert rel_l2(a.grad, bb.grad) < 1e-5


def test_encoder_train_ragged_large_vs_oracle_on_split_cores():
    """Train-mode encoder at a size that engages the default split-fp16 cores (NT second
    generation, transposed-read wgrad, in-place dz, operand maxima from the statistics epilogue)
    with NOTHING aligned: P = 5 x 1999 = 9995 rows (not a multiple of the 256-row tile, of the
    128-row statistics block or of the 16/32-row k-tiles), against the oracle, forward and
    backward with gradients arriving on both outputs."""
    C, B, N = 4, 5, 1999
    sd = P.encoder_state_dict(C, 1024, seed=9)
    ctx, _, _ = P.synth_batch(B, N, C, 32, seed=31)
    r = np.random.default_rng(6)
    up_g = torch.from_numpy(r.normal(0, 1, (B, 2048)).astype(np.float32))
    up_f = torch.from_numpy(r.normal(0, 1, (B, N, 1024)).astype(np.float32))
    m = _encoder(C, sd).train()
    x = ctx.cuda().requires_grad_(True)
    gf, fu_cm = m(x.transpose(2, 1))
    ((gf * up_g.cuda()).sum() + (fu_cm.transpose(2, 1) * up_f.cuda()).sum()).backward()
    p = O.as_params(sd, requires_grad=True)
    ox = ctx.clone().requires_grad_(True)
    ns = {}
    o_g, o_f = O.encoder_forward(p, ox, "", True, ns)
    ((o_g * up_g).sum() + (o_f * up_f).sum()).backward()
    assert maxdiff(gf, o_g) < 1e-4 and maxdiff(fu_cm.transpose(2, 1), o_f) < 1e-4
    assert rel_l2(ox.grad, x.grad) < 5e-3
    rels = {}
    named = dict(m.named_parameters())
    for k, v in named.items():
        if _pre_bn_bias(k):
            continue
        rels[k] = rel_l2(p[k].grad.reshape(v.shape), v.grad)
    assert max(rels.values()) < 1e-2, max(rels, key=rels.get)
    assert float(np.median(list(rels.values()))) < 2e-3
    msd = m.state_dict()
    for k, v in ns.items():
        assert maxdiff(msd[k], v) <= 1e-5 * float(v.double().abs().max()) + 1e-6, k


@pytest.mark.parametrize("gemm_mode,B,N", [(3, 4, 512), (3, 2, 1024), (4, 3, 256)])
def test_pooling_fused_into_the_gate_epilogue(gemm_mode, B, N):
    """Segments of a whole number of 128-row wave tiles: the dual pooling (src/model.py:58-60)
    rides on the epilogue that writes `fused` (F_POOL + pool_tiles_kernel) instead of a pass of its
    own.  Forward against the oracle, and the backward with upstream gradient on global_feat ONLY,
    so every parameter gradient flows through the arg-max rows the epilogue recorded - incl. a
    dead channel, whose maximum ties at 0 on every point (first index wins)."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    old = lib.prh_get_gemm_mode()
    sd = P.encoder_state_dict(4, 1024, seed=3)
    sd["fusion.1.weight"][5] = 0.0
    sd["fusion.1.bias"][5] = -1.0
    ctx, _, _ = P.synth_batch(B, N, 4, 32, seed=31)
    up_g = torch.from_numpy(np.random.default_rng(2).normal(0, 1, (B, 2048)).astype(np.float32))
    try:
        lib.prh_set_gemm_mode(gemm_mode)
        m = _encoder(4, sd).train()
        x = ctx.cuda().requires_grad_(True)
        gf, fu = m(x.transpose(2, 1))
        (gf * up_g.cuda()).sum().backward()
    finally:
        lib.prh_set_gemm_mode(old)
    p = O.as_params(sd, requires_grad=True)
    ox = ctx.clone().requires_grad_(True)
    o_g, o_f = O.encoder_forward(p, ox, "", True, {})
    (o_g * up_g).sum().backward()
    named = dict(m.named_parameters())
    if gemm_mode == 3:
        assert maxdiff(gf, o_g) < 1e-4 and maxdiff(fu.transpose(2, 1), o_f) < 1e-4
        assert rel_l2(ox.grad, x.grad) < 5e-3
        assert rel_l2(p["fusion.0.weight"].grad.reshape(named["fusion.0.weight"].shape), named["fusion.0.weight"].grad) < 5e-3
        # the dead channel: gradient of its maximum went to point 0 of every segment, as torch.max does
        assert rel_l2(p["intensity_gate.2.bias"].grad, named["intensity_gate.2.bias"].grad) < 5e-3
    else:
        # bf16 storage: reduced-precision gates (tests/test_bf16_gpu.py).  With the gradient entering
        # through the max-pool only, a near-tie between two points that bf16 rounding resolves the
        # other way moves that channel's whole gradient to another point: measured 0.22 on fusion dW
        assert rel_l2(o_g, gf) < 3e-2
        assert rel_l2(p["fusion.0.weight"].grad.reshape(named["fusion.0.weight"].shape), named["fusion.0.weight"].grad) < 0.4


def test_config4_per_rank_share_c6_encoder_two_kernel_families_agree():
    """BASELINE config 4 names C=6 inputs (xyz + intensity + normals) at N=2048 over 8 GPUs: 512 segments per rank.
    LineRefineNet is built on the 4-channel encoder (src/model.py:142), so C=6 exists at the encoder API
    (MultiScalePointNetEncoder(in_channel=6), src/model.py:7).  That rank's share at size - B=512, N=2048,
    train mode, both returns used - is far beyond the CPU oracle, so it is a cross-check of two independent
    kernel families on identical inputs (each pinned to the C=6 golden vectors at fixture size above): the
    split-fp16 cores against the exact fp32 MFMA cores."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    B, N, C = 512, 2048, 6
    sd = P.encoder_state_dict(C, 1024, seed=3)
    ctx, _, _ = P.synth_batch(B, N, C, 32, seed=11)
    g = torch.Generator(device="cuda").manual_seed(4)
    # upstream gradient on the MEAN half of global_feat and on fused: the max half routes its gradient through ONE
    # point per (segment, channel), and where a channel's two largest activations sit inside fp32 noise of each other
    # the two families pick different points and every upstream gradient moves by ~1e-3 (the g4 / train fixture above
    # shows 2.2e-3 between the reference and the oracle themselves) - that would measure ties, not cores
    up_g = torch.randn(B, 2048, device="cuda", generator=g)
    up_g[:, :1024] = 0.0
    # (and a one-signed upstream gradient on fused: with zero-mean noise every parameter gradient is a cancelling
    # sum over a million points and both families sit ~1e-3 from each other in fp32 accumulation noise alone)
    up_f = (1.0 + 0.3 * torch.randn(B, N, 1024, device="cuda", generator=g)) * 0.05
    res = {}
    old = lib.prh_get_gemm_mode()
    try:
        for mode in (0, 3):
            lib.prh_set_gemm_mode(mode)
            m = _encoder(C, sd).train()
            x = ctx.cuda().requires_grad_(True)
            gf, fu_cm = m(x.transpose(2, 1))
            assert gf.shape == (B, 2048) and fu_cm.shape == (B, 1024, N)
            ((gf * up_g).sum() + (fu_cm.transpose(2, 1) * up_f).sum()).backward()
            res[mode] = (gf.detach().clone(), fu_cm.detach()[:, :, ::64].clone(), x.grad.clone(),
                         {k: v.grad.clone() for k, v in m.named_parameters()},
                         {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
            del m, x, gf, fu_cm
            torch.cuda.empty_cache()
    finally:
        lib.prh_set_gemm_mode(old)
    a, b = res[0], res[3]
    assert maxdiff(a[0], b[0]) < 1e-4 * max(1.0, float(a[0].abs().max()))
    assert maxdiff(a[1], b[1]) < 1e-4 * max(1.0, float(a[1].abs().max()))
    assert rel_l2(a[2], b[2]) < 2e-3
    rels = {k: rel_l2(a[3][k], b[3][k]) for k in a[3] if not _pre_bn_bias(k)}
    top = sorted(rels.items(), key=lambda kv: -kv[1])[:3]
    med = float(np.median(list(rels.values())))
    print(f"C=6, B=512, N=2048: split-fp16 vs exact fp32 cores, parameter-gradient rel-L2 median {med:.2e}, worst {top}")
    assert top[0][1] < 5e-4, top          # measured: median 1.7e-5, worst 3.5e-5 (bn1.weight)
    for k, v in a[4].items():
        assert maxdiff(v, b[4][k]) <= 1e-5 * float(v.abs().max()) + 1e-7, k
