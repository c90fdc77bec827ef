"""LineRefineNet on the HIP path against the golden vectors generated from the reference
(G1 eval forward, G2 train-mode forward+backward with dropout forced to 0) and against
the oracle; plus size-independent properties at the BASELINE config-2 size."""
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


def _model(sd=None):
    from pointnet_refine_amd.model import LineRefineNet
    m = LineRefineNet()
    if sd is not None:
        m.load_state_dict(sd, strict=True)
    return m.cuda()


def _zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0


def test_g1_eval_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_eval_forward.npz"))
    sd = P.linerefine_state_dict(0)
    m = _model(sd).eval()
    ctx, noisy, _ = P.synth_batch(8, 256, 4, 32, seed=1234)
    with torch.no_grad():
        out = m(ctx.cuda(), noisy.cuda())
        memory = m.encode_context(ctx.cuda())
        gf, fu = m.context_encoder(ctx.cuda().transpose(2, 1))
    assert out.shape == (6, 8, 32, 3)
    assert maxdiff(out, g["out"]) < 1e-4                       # north_star gate
    assert maxdiff(gf, g["global_feat"]) < 1e-4
    assert maxdiff(memory[:, ::16, ::8], g["memory_sub"]) < 1e-4
    assert maxdiff(fu.transpose(2, 1)[:, ::16, ::8], g["fused_sub"]) < 1e-4


def test_g2_train_fwd_bwd(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_train_fwd_bwd.npz"))
    sd = P.linerefine_state_dict(0)
    m = _model(sd).train()
    _zero_dropout(m)
    ctx, noisy, target = P.synth_batch(8, 256, 4, 32, seed=1234)
    c = ctx.cuda().requires_grad_(True)
    nl = noisy.cuda().requires_grad_(True)
    out = m(c, nl)
    loss = sum(torch.nn.functional.l1_loss(out[l], target.cuda()) for l in range(6)) / 6
    loss.backward()
    assert maxdiff(out, g["out"]) < 2e-4
    assert abs(float(loss) - float(g["loss"])) < 2e-5
    assert rel_l2(g["dctx"], c.grad) < 5e-3
    assert rel_l2(g["dnoisy"], nl.grad) < 5e-3
    named = dict(m.named_parameters())
    rels = []
    for k, nrm in zip(g["grad_keys"], g["grad_norms"]):
        k = str(k)
        if _pre_bn_bias(k):
            continue
        gr = named[k].grad.reshape(-1).double()
        assert abs(float(gr.norm()) - nrm) <= 5e-3 * nrm + 1e-12, k
        rels.append(rel_l2(g["gh::" + k], gr[:64]))
    assert max(rels) < 2e-2 and float(np.median(rels)) < 1e-3
    msd = m.state_dict()
    for k in msd:
        if "running" in k or "num_batches" in k:
            ref = torch.from_numpy(g["st::" + k])
            assert maxdiff(msd[k], ref) <= 1e-5 * float(ref.double().abs().max()) + 1e-6, k


def test_full_size_properties():
    """BASELINE config 2 size (B=512, N=1024): properties that need no oracle run.
    (a) eval-mode segments are independent: a segment's output does not depend on its
        batch-mates; (b) global max/mean pooling is invariant to a permutation of the
        points, `fused` is equivariant; (c) train-mode BN output statistics: the
        pre-ReLU fusion activations have batch mean beta and variance gamma^2."""
    sd = P.linerefine_state_dict(0)
    m = _model(sd).eval()
    B, N = 512, 1024
    ctx, noisy, _ = P.synth_batch(B, N, 4, 32, seed=99)
    ctx, noisy = ctx.cuda(), noisy.cuda()
    with torch.no_grad():
        out = m(ctx, noisy)
        out_sub = m(ctx[37:41].contiguous(), noisy[37:41].contiguous())
        # different batch sizes pick different GEMM cores (exact fp32 vs split-bf16), so the
        # comparison is at the network's fp32 noise floor (~1e-5 through 6 layers), not bitwise
        assert maxdiff(out[:, 37:41], out_sub) < 5e-5
        enc = m.context_encoder
        gf, fu = enc(ctx[:64].transpose(2, 1))
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(7)).cuda()
        gf2, fu2 = enc(ctx[:64, perm].contiguous().transpose(2, 1))
        assert maxdiff(gf[:, :1024], gf2[:, :1024]) == 0.0           # max: exact
        # mean: fp32 running sums of N values in a different order differ by ~sqrt(N)*eps*|sum|
        # (a few 1e-6 per unit of magnitude at N=1024); both must sit that close to the fp64 mean
        mean64 = fu.double().mean(-1)
        tol = 1e-5 * max(1.0, float(fu.max()))
        assert maxdiff(gf[:, 1024:], mean64) < tol and maxdiff(gf2[:, 1024:], mean64) < tol
        assert maxdiff(fu[:, :, perm], fu2) < 1e-5
        assert float(fu.min()) >= 0.0
    # small oracle cross-check of a slice of the big batch (eval => batch independent)
    with torch.no_grad():
        o = O.linerefine_forward(O.as_params(sd), ctx[100:102].cpu(), noisy[100:102].cpu())
    assert maxdiff(out[:, 100:102], o) < 1e-4


@pytest.mark.parametrize("mode", [0, 1, 3], ids=["fp32-cores", "split-bf16-cores", "split-fp16-cores"])
def test_g1_g2_on_both_gemm_cores(golden_dir, mode):
    """The golden vectors hold on the exact fp32 MFMA cores, the split-bf16 cores and the
    split-fp16 cores."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    old = lib.prh_get_gemm_mode()
    lib.prh_set_gemm_mode(mode)
    try:
        g = np.load(os.path.join(golden_dir, "g1_eval_forward.npz"))
        sd = P.linerefine_state_dict(0)
        m = _model(sd).eval()
        ctx, noisy, target = P.synth_batch(8, 256, 4, 32, seed=1234)
        with torch.no_grad():
            out = m(ctx.cuda(), noisy.cuda())
        assert maxdiff(out, g["out"]) < 1e-4
        g2 = np.load(os.path.join(golden_dir, "g2_train_fwd_bwd.npz"))
        m = _model(sd).train()
        _zero_dropout(m)
        out = m(ctx.cuda(), noisy.cuda())
        loss = sum(torch.nn.functional.l1_loss(out[l], target.cuda()) for l in range(6)) / 6
        loss.backward()
        assert maxdiff(out, g2["out"]) < 2e-4
        k = "context_encoder.fusion.0.weight"
        gr = dict(m.named_parameters())[k].grad.reshape(-1).double()
        nrm = float(g2["grad_norms"][list(g2["grad_keys"]).index(k)])
        assert abs(float(gr.norm()) - nrm) <= 5e-3 * nrm
    finally:
        lib.prh_set_gemm_mode(old)


def test_batch_4096_points_2048_shapes_run():
    """BASELINE config 4 shape per rank scaled to what one test may hold (B=64, N=2048):
    eval forward agrees with the oracle on a slice."""
    sd = P.linerefine_state_dict(0)
    m = _model(sd).eval()
    ctx, noisy, _ = P.synth_batch(64, 2048, 4, 32, seed=5)
    with torch.no_grad():
        out = m(ctx.cuda(), noisy.cuda())
        o = O.linerefine_forward(O.as_params(sd), ctx[10:12], noisy[10:12])
    assert out.shape == (6, 64, 32, 3)
    assert maxdiff(out[:, 10:12], o) < 1e-4


def test_bf16_mode_loose_gate(golden_dir):
    """BASELINE config 3 (bf16 arithmetic): eval forward within the LOOSER 5e-2 gate the
    survey sets for reduced precision (CPU bf16 autocast of the reference shows 1.7e-2)."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    old = lib.prh_get_gemm_mode()
    lib.prh_set_gemm_mode(2)
    try:
        g = np.load(os.path.join(golden_dir, "g1_eval_forward.npz"))
        m = _model(P.linerefine_state_dict(0)).eval()
        ctx, noisy, _ = P.synth_batch(8, 256, 4, 32, seed=1234)
        with torch.no_grad():
            out = m(ctx.cuda(), noisy.cuda())
        err = maxdiff(out, g["out"])
        print("bf16-mode max abs error on out:", err)
        assert err < 5e-2
    finally:
        lib.prh_set_gemm_mode(old)


def test_config2_fwd_bwd_two_kernel_families_agree():
    """BASELINE config 2 (B=512, N=1024, fp32 forward+backward, train-mode BN) is too large
    for the CPU oracle, so the full-size check is a cross-check of two independent kernel
    families on identical inputs: the exact fp32 MFMA cores against the split-bf16 cores and
    the split-fp16 cores (each individually pinned to the golden vectors at B=8).  Outputs
    within 1e-4, every parameter gradient within 2e-3 rel-L2, BN running statistics within
    1e-5 relative."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(512, 1024, 4, 32, seed=2)
    ctx, noisy, target = ctx.cuda(), noisy.cuda(), target.cuda()
    res = {}
    old = lib.prh_get_gemm_mode()
    try:
        for mode in (0, 1, 3):
            lib.prh_set_gemm_mode(mode)
            m = _model(sd).train()
            _zero_dropout(m)
            out = m(ctx, noisy)
            loss = (out - target.unsqueeze(0)).abs().mean()
            loss.backward()
            res[mode] = (out.detach(), {k: v.grad for k, v in m.named_parameters()},
                         {k: v.clone() for k, v in m.state_dict().items() if "running" in k})
            del m, out, loss
    finally:
        lib.prh_set_gemm_mode(old)
    for other in (1, 3):
        assert maxdiff(res[0][0], res[other][0]) < 1e-4
        worst = 0.0
        for k, g0 in res[0][1].items():
            if _pre_bn_bias(k):
                continue
            worst = max(worst, rel_l2(g0, res[other][1][k]))
        print(f"mode {other} vs exact fp32 cores: out max|d| {maxdiff(res[0][0], res[other][0]):.2e}, "
              f"worst gradient rel-L2 {worst:.2e}")
        assert worst < 2e-3, (other, worst)
        for k, v in res[0][2].items():
            assert maxdiff(v, res[other][2][k]) <= 1e-5 * float(v.abs().max()) + 1e-7, (other, k)


def test_decoder_layer_standalone_call_contract():
    """DetrTransformerDecoderLayer(tgt, memory, query_pos, pos) — the reference's own call
    signature (src/model.py:103) — against the oracle's decoder_layer, forward and backward."""
    from pointnet_refine_amd.model import DetrTransformerDecoderLayer
    torch.manual_seed(5)
    layer = DetrTransformerDecoderLayer(d_model=256, nhead=8, dim_feedforward=1024, dropout=0.0).cuda().train()
    B, M, N = 3, 32, 192
    tgt = torch.randn(B, M, 256, device="cuda", requires_grad=True)
    mem = torch.randn(B, N, 256, device="cuda", requires_grad=True)
    qp = torch.randn(B, M, 256, device="cuda")
    pp = torch.randn(B, N, 256, device="cuda")
    w = torch.randn(B, M, 256, device="cuda")
    out = layer(tgt, mem, query_pos=qp, pos=pp)
    (out * w).sum().backward()

    p = {"L." + k: v.detach().double().cpu().requires_grad_(True) for k, v in layer.state_dict().items()}
    t2 = tgt.detach().double().cpu().requires_grad_(True)
    m2 = mem.detach().double().cpu().requires_grad_(True)
    ref = O.decoder_layer(p, "L", t2, m2, qp.double().cpu(), pp.double().cpu())
    (ref * w.double().cpu()).sum().backward()
    assert maxdiff(out, ref) < 1e-4
    assert rel_l2(tgt.grad, t2.grad) < 1e-4
    assert rel_l2(mem.grad, m2.grad) < 1e-4
    for k, v in layer.named_parameters():
        assert rel_l2(v.grad, p["L." + k].grad) < 1e-4, k
    # pos / query_pos may be omitted (with_pos_embed(None))
    out2 = layer(tgt, mem)
    ref2 = O.decoder_layer(p, "L", t2, m2, torch.zeros_like(t2), torch.zeros_like(m2))
    assert maxdiff(out2, ref2) < 1e-4


def test_default_model_never_takes_a_stock_pytorch_branch():
    """The product model has no stock-PyTorch arithmetic branch (unsupported shapes raise): a train
    step and an eval forward never reach torch's own Linear / attention / LayerNorm / BatchNorm / conv
    kernels (patched to raise), and a shape the kernels do not take raises RuntimeError instead of
    quietly running on torch."""
    from pointnet_refine_amd import model as M
    import torch.nn.functional as F
    m = _model(P.linerefine_state_dict(0))
    ctx, noisy, target = P.synth_batch(4, 128, 4, 32, seed=2)
    ctx, noisy, target = ctx.cuda(), noisy.cuda(), target.cuda()

    def boom(name):
        def f(*a, **k):
            raise AssertionError(f"stock torch.nn.functional.{name} reached from the default model")
        return f

    saved = {n: getattr(F, n) for n in ("linear", "scaled_dot_product_attention", "layer_norm",
                                        "multi_head_attention_forward", "batch_norm", "conv1d")}
    try:
        for n in saved:
            setattr(F, n, boom(n))
        m.train()
        out = m(ctx, noisy)
        (out - target.unsqueeze(0)).abs().mean().backward()
        m.eval()
        with torch.no_grad():
            m(ctx, noisy)
            m.context_encoder(ctx.transpose(2, 1).contiguous())
    finally:
        for n, f in saved.items():
            setattr(F, n, f)
    assert all(p.grad is not None for p in m.parameters())
    assert not hasattr(M, "FALLBACKS")
    # shapes outside the kernels' range fail loudly
    odd = M.DetrTransformerDecoderLayer(d_model=192, nhead=8, dim_feedforward=256, dropout=0.0).cuda()
    with pytest.raises(RuntimeError):
        odd(torch.randn(2, 32, 192, device="cuda"), torch.randn(2, 64, 192, device="cuda"))
    with pytest.raises(RuntimeError):
        M.PositionalEncoding(in_dim=2, out_dim=256).cuda()(torch.randn(2, 8, 2, device="cuda"))


@pytest.mark.parametrize("B,N,M", [(3, 77, 20), (1, 1, 32), (2, 130, 7), (5, 33, 48)])
def test_ragged_shapes_forward_backward_vs_oracle(B, N, M):
    """The reference takes any B >= 1, N >= 1 and any number of line points M (src/model.py:181-234; its
    callers use M = 32).  Odd sizes - N not a multiple of any tile, a single context point, M below and above
    32 (two query tiles in the attention kernels, 1..2 row tiles in the small GEMM cores) - eval forward and
    train-mode forward + backward (dropout off) against the oracle."""
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(B, N, 4, M, seed=B * 100 + N + M)
    m = _model(sd).eval()
    with torch.no_grad():
        out = m(ctx.cuda(), noisy.cuda())
        ref = O.linerefine_forward(O.as_params(sd), ctx, noisy)
    assert out.shape == (6, B, M, 3)
    assert maxdiff(out, ref) < 1e-4
    if B * N < 2:
        with pytest.raises(ValueError):
            m.train()(ctx.cuda(), noisy.cuda())              # one value per channel: BatchNorm refuses, as the reference does
        return
    m = _model(sd).train()
    _zero_dropout(m)
    out = m(ctx.cuda(), noisy.cuda())
    loss = (out - target.cuda().unsqueeze(0)).abs().mean()
    loss.backward()
    p = O.as_params(sd, dtype=torch.float64, requires_grad=True)
    o64 = O.linerefine_forward(p, ctx.double(), noisy.double(), training=True, new_stats={})
    (o64 - target.double().unsqueeze(0)).abs().mean().backward()
    assert maxdiff(out, o64) < 2e-4
    worst = max((rel_l2(p[k].grad, v.grad), k) for k, v in m.named_parameters() if not _pre_bn_bias(k))
    # 60..240 query rows: ONE ReLU unit whose pre-activation sits inside fp32 noise of zero (fp32 here, fp64 in
    # the oracle) moves its layer's gradients by 1 / sqrt(rows x active units) ~ 5e-3 (tests/diag_ragged.py: the
    # same 5.13e-3 on layer 4 in all three GEMM modes, whose query-side arithmetic is identical; 5.36e-3 on
    # layer 0 only in the modes whose encoder rounds differently) - so the gate here is 1e-2, against 2e-3 at
    # the fixture and benchmark sizes (tests/test_oracle_fp64_gpu.py)
    assert worst[0] < 1e-2, worst
