"""Inference-only cross-attention with the key / value projections folded in (csrc/prh_attnfold.hpp,
SURVEY 8(f) f1, src/model.py:119-128 in eval mode): attention over the raw rows of memory + pos and
memory instead of per-layer projected buffers.  Checked against the written-out reference
arithmetic in fp64 (bf16 operand rounding: 5e-2 class, BASELINE config 5), through the whole model
against the golden vectors generated from the reference (G1), and against the unfolded path."""
import os

import numpy as np
import pytest
import torch

from conftest import maxdiff, rel_l2
from oracle import procedural as P

pytestmark = pytest.mark.gpu


def _ref(q, x, y, wk, bk, wv, bv, heads):
    """nn.MultiheadAttention's cross-attention core in fp64: K = x Wk^T + bk, V = y Wv^T + bv."""
    B, M, C = q.shape
    d = C // heads
    k = x.double() @ wk.double().t() + bk.double()
    v = y.double() @ wv.double().t() + bv.double()
    qh = q.double().view(B, M, heads, d).transpose(1, 2)
    kh = k.view(B, -1, heads, d).transpose(1, 2)
    vh = v.view(B, -1, heads, d).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / d ** 0.5, dim=-1)
    return (att @ vh).transpose(1, 2).reshape(B, M, C)


@pytest.mark.parametrize("B,M,N", [(2, 32, 64), (3, 32, 1000), (1, 32, 1), (2, 20, 33), (5, 32, 256), (2, 7, 95)])
def test_folded_attention_against_fp64(B, M, N):
    from pointnet_refine_amd import ops
    g = torch.Generator().manual_seed(B * 100 + N)
    q = torch.randn(B, M, 256, generator=g).cuda()
    x = torch.randn(B, N, 256, generator=g).cuda()
    y = torch.randn(B, N, 256, generator=g).cuda()
    wk = (torch.randn(256, 256, generator=g) / 16).cuda()
    wv = (torch.randn(256, 256, generator=g) / 16).cuda()
    bk = torch.randn(256, generator=g).cuda()          # must not matter: constant along the keys
    bv = torch.randn(256, generator=g).cuda()
    with torch.no_grad():
        o = ops.attention_folded(q, ops.cast_perm_bf16(x), ops.cast_perm_bf16(y), wk, wv, bv, 8)
    ref = _ref(q, x, y, wk, bk, wv, bv, 8)
    assert torch.isfinite(o).all()
    # bf16 operands (8 significant bits) through two 256-deep products and a softmax
    assert rel_l2(ref, o) < 2e-2, rel_l2(ref, o)
    assert maxdiff(o, ref) < 5e-2 * float(ref.abs().max())


def test_folded_attention_is_inference_only():
    from pointnet_refine_amd import ops
    q = torch.randn(1, 32, 256, device="cuda", requires_grad=True)
    x = ops.cast_perm_bf16(torch.randn(1, 64, 256, device="cuda"))
    w = torch.randn(256, 256, device="cuda")
    with pytest.raises(RuntimeError):
        ops.attention_folded(q, x, x, w, w, torch.zeros(256, device="cuda"), 8)


def test_g1_and_the_unfolded_path_through_the_model(golden_dir):
    """Config 5 (fused fp16 encoder + bf16 decoder GEMMs) with and without the fold: both within
    5e-2 of the reference's eval forward (G1), and close to each other."""
    from pointnet_refine_amd import ops
    from pointnet_refine_amd.model import LineRefineNet
    g = np.load(os.path.join(golden_dir, "g1_eval_forward.npz"))
    m = LineRefineNet()
    m.load_state_dict(P.linerefine_state_dict(0), strict=True)
    m = m.cuda().eval()
    m.context_encoder.inference_precision = "fp16"
    ctx, noisy, _ = P.synth_batch(8, 256, 4, 32, seed=1234)
    calls = []
    orig = ops.attention_folded
    ops.attention_folded = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    old = ops.set_gemm_mode("bf16")
    try:
        with torch.no_grad():
            out_fold = m(ctx.cuda(), noisy.cuda())
            m.fold_kv_inference = False
            out_plain = m(ctx.cuda(), noisy.cuda())
        m.fold_kv_inference = True
        m.train()                                   # training never takes the folded kernel
        n_eval = len(calls)
        m(ctx.cuda(), noisy.cuda())
        assert len(calls) == n_eval == 6
    finally:
        ops.attention_folded = orig
        ops.set_gemm_mode(old)
    assert maxdiff(out_fold, g["out"]) < 5e-2
    assert maxdiff(out_plain, g["out"]) < 5e-2
    assert maxdiff(out_fold, out_plain) < 3e-2


@pytest.mark.parametrize("B,N", [(2, 100), (1, 1), (3, 256), (5, 777)])
def test_posmem_images_match_the_separate_path(B, N):
    """ops.posmem_images = pos_hidden + Linear(256,256) with the memory residual + the two casts, in
    one pass: same images as the separate path up to bf16 rounding of the hidden layer."""
    from pointnet_refine_amd import ops
    g = torch.Generator().manual_seed(B + N)
    ctx = torch.randn(B, N, 4, generator=g).cuda()
    mem = torch.randn(B, N, 256, generator=g).cuda()
    w0 = torch.randn(256, 3, generator=g).cuda()
    b0 = torch.randn(256, generator=g).cuda()
    w2 = (torch.randn(256, 256, generator=g) / 16).cuda()
    b2 = torch.randn(256, generator=g).cuda()
    with torch.no_grad():
        x16, y16 = ops.posmem_images(ctx[:, :, :3], mem, w0, b0, w2, b2)
        h = torch.relu(ctx[:, :, :3].double() @ w0.double().t() + b0.double())
        mempos = mem.double() + h @ w2.double().t() + b2.double()
        # the opaque channel order is the one cast_perm_bf16 produces
        assert torch.equal(y16, ops.cast_perm_bf16(mem))
        want = ops.cast_perm_bf16(mempos.float()).float()
    got = x16.float()
    assert got.shape == (B * N, 256)
    assert rel_l2(want, got) < 1e-2          # bf16 hidden layer and weights, fp32 accumulation
    assert maxdiff(got, want) < 0.05 * float(want.abs().max())
