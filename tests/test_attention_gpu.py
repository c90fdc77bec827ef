"""Fused cross-attention core (prh_attn_forward/backward through ops.attention) against an
fp64 written-out reference (the oracle's mha core: softmax(q k^T / sqrt(32)) v per head),
on strided K/V column blocks like the decoder uses, with ragged N and with dropout (the
kernel's counter-based keep-mask is re-created by ops.attention_keep_mask)."""
import pytest
import torch

from conftest import maxdiff, rel_l2

pytestmark = pytest.mark.gpu


def _ref(q, k, v, H, keep=None, p=0.0):
    B, M, C = q.shape
    N = k.shape[1]
    d = C // H
    qh = q.double().view(B, M, H, d).transpose(1, 2)
    kh = k.double().view(B, N, H, d).transpose(1, 2)
    vh = v.double().view(B, N, H, d).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / d ** 0.5, dim=-1)
    if keep is not None:
        att = att * keep.to(att.dtype) / (1.0 - p)
    return (att @ vh).transpose(1, 2).reshape(B, M, C)


@pytest.mark.parametrize("B,M,N,p", [(3, 32, 64, 0.0), (2, 32, 1000, 0.0), (5, 32, 33, 0.0),
                                     (2, 32, 257, 0.1), (1, 64, 96, 0.25), (3, 20, 70, 0.0),
                                     (2, 45, 130, 0.2),
                                     # key-split mode (B x H <= 1024, N >= 128): 8 waves per head;
                                     # and the same key count with too many heads for it
                                     (32, 32, 2048, 0.1), (130, 32, 256, 0.1)])
def test_attention_fwd_bwd(B, M, N, p):
    from pointnet_refine_amd import ops
    H, C = 8, 256
    g = torch.Generator().manual_seed(B * 1000 + N)
    q = torch.randn(B, M, C, generator=g)
    kv = torch.randn(B, N, 6 * C, generator=g)          # K/V as column blocks of a wide buffer
    up = torch.randn(B, M, C, generator=g)
    seed = 1234567
    keep = ops.attention_keep_mask(B, H, M, N, p, seed) if p > 0 else None

    qr, kvr = q.clone().requires_grad_(True), kv.clone().requires_grad_(True)
    o_ref = _ref(qr, kvr[..., 2 * C:3 * C], kvr[..., 4 * C:5 * C], H, keep, p)
    (o_ref * up.double()).sum().backward()

    qg, kvg = q.cuda().requires_grad_(True), kv.cuda().requires_grad_(True)
    o = ops.attention(qg, kvg[..., 2 * C:3 * C], kvg[..., 4 * C:5 * C], H, p, seed)
    (o * up.cuda()).sum().backward()
    assert maxdiff(o, o_ref) < 2e-5
    assert rel_l2(qr.grad, qg.grad) < 2e-5
    assert rel_l2(kvr.grad, kvg.grad) < 2e-5
    # untouched column blocks receive exactly zero
    assert float(kvg.grad[..., :2 * C].abs().max()) == 0.0


def test_attention_dropout_rate_and_determinism():
    from pointnet_refine_amd import ops
    keep = ops.attention_keep_mask(4, 8, 32, 512, 0.1, 99)
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - 0.1) < 5e-3
    q = torch.randn(2, 32, 256, device="cuda")
    k = torch.randn(2, 300, 256, device="cuda")
    v = torch.randn(2, 300, 256, device="cuda")
    a = ops.attention(q, k, v, 8, 0.1, 7)
    b = ops.attention(q, k, v, 8, 0.1, 7)
    c = ops.attention(q, k, v, 8, 0.1, 8)
    assert torch.equal(a, b) and not torch.equal(a, c)


def test_attention_arena_blocks_match_plain_path():
    """Six attention calls against column blocks of one wide K/V buffer with the gradient
    arena give the same outputs and the same dK_all/dV_all as six plain calls on slices."""
    from pointnet_refine_amd import ops
    B, M, N, H, C = 2, 32, 200, 8, 256
    g = torch.Generator().manual_seed(11)
    q = [torch.randn(B, M, C, generator=g).cuda().requires_grad_(True) for _ in range(6)]
    k_all = torch.randn(B, N, 6 * C, generator=g).cuda()
    v_all = torch.randn(B, N, 6 * C, generator=g).cuda()
    up = torch.randn(B, M, C, generator=g).cuda()

    ka, va = k_all.clone().requires_grad_(True), v_all.clone().requires_grad_(True)
    token, arena = ops.kv_token(ka, va, C)
    outs = [ops.attention_block(q[i], ka, va, token, arena, i, H) for i in (0, 1, 2, 4, 5)]   # block 3 unused
    sum((o * up).sum() for o in outs).backward()
    dq_a = [t.grad.clone() if t.grad is not None else None for t in q]
    for t in q:
        t.grad = None

    kb, vb = k_all.clone().requires_grad_(True), v_all.clone().requires_grad_(True)
    outs_b = [ops.attention(q[i], kb[..., i * C:(i + 1) * C], vb[..., i * C:(i + 1) * C], H) for i in (0, 1, 2, 4, 5)]
    sum((o * up).sum() for o in outs_b).backward()
    for a, b in zip(outs, outs_b):
        assert torch.equal(a, b)
    assert maxdiff(ka.grad, kb.grad) < 1e-6 and maxdiff(va.grad, vb.grad) < 1e-6
    assert float(ka.grad[..., 3 * C:4 * C].abs().max()) == 0.0       # unused block is zero-filled
    for i in (0, 1, 2, 4, 5):
        assert maxdiff(dq_a[i], q[i].grad) < 1e-6


def test_add_dropout_layernorm_fused():
    """y = LayerNorm(x + dropout(r)) (src/model.py:117,128,133) against torch in fp64, forward and
    backward, without dropout and with the kernel's own keep-mask re-created in torch."""
    import torch
    from pointnet_refine_amd import ops
    torch.manual_seed(2)
    for rows, p, seed in ((7, 0.0, 0), (4099, 0.0, 0), (1000, 0.25, 12345), (65, 0.1, 77)):
        ln = torch.nn.LayerNorm(256).cuda()
        with torch.no_grad():
            ln.weight.uniform_(0.5, 1.5); ln.bias.normal_(0, 0.2)
        x = torch.randn(rows, 256, device="cuda", requires_grad=True)
        r = (torch.randn(rows, 256, device="cuda") * 3 + 1).requires_grad_(True)
        w = torch.randn(rows, 256, device="cuda")
        y = ops.AddDropoutLayerNormFn.apply(x, r, ln.weight, ln.bias, ln.eps, p, seed)
        (y * w).sum().backward()
        keep = ops.layernorm_keep_mask(rows, 256, p, seed, "cuda").double() / (1.0 - p) if p > 0 else 1.0
        x2 = x.detach().double().requires_grad_(True)
        r2 = r.detach().double().requires_grad_(True)
        g2 = ln.weight.detach().double().requires_grad_(True)
        b2 = ln.bias.detach().double().requires_grad_(True)
        ref = torch.nn.functional.layer_norm(x2 + r2 * keep, (256,), g2, b2, ln.eps)
        (ref * w.double()).sum().backward()
        assert maxdiff(y, ref) < 2e-5
        assert rel_l2(x.grad, x2.grad) < 2e-6 and rel_l2(r.grad, r2.grad) < 2e-6
        assert rel_l2(ln.weight.grad, g2.grad) < 2e-6 and rel_l2(ln.bias.grad, b2.grad) < 2e-6
        if p > 0:
            frac = float(ops.layernorm_keep_mask(rows, 256, p, seed).float().mean())
            assert abs(frac - (1 - p)) < 0.01


def test_linear_with_fused_relu():
    """relu(x W^T + b) with the ReLU in the GEMM epilogue (FFN of the decoder layer,
    src/model.py:131), forward and backward against torch fp64, on both GEMM paths (small: exact
    fp32 core, large: split core)."""
    from pointnet_refine_amd import ops
    torch.manual_seed(4)
    for rows in (40, 65536):
        x = torch.randn(rows, 256, device="cuda", requires_grad=True)
        w = (torch.randn(1024, 256, device="cuda") * 0.05).requires_grad_(True)
        b = (torch.randn(1024, device="cuda") * 0.1).requires_grad_(True)
        g = torch.randn(rows, 1024, device="cuda")
        y = ops.linear(x, w, b, None, True)
        (y * g).sum().backward()
        x2, w2, b2 = (t.detach().double().requires_grad_(True) for t in (x, w, b))
        ref = torch.relu(torch.nn.functional.linear(x2, w2, b2))
        (ref * g.double()).sum().backward()
        assert float(y.detach().min()) >= 0.0 and maxdiff(y, ref) < 2e-5
        # entries within fp32 noise of zero may take either side of the ReLU: rel-L2 per tensor
        assert rel_l2(x.grad, x2.grad) < 1e-3 and rel_l2(w.grad, w2.grad) < 1e-3 and rel_l2(b.grad, b2.grad) < 1e-3
