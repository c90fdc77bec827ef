"""Positional-encoding MLP on the HIP path (src/model.py:64-75): the 3-wide first layer as an
elementwise kernel over points read in place, the second Linear with the residual (memory +
pos, src/model.py:123-126) in its epilogue - against the written-out torch arithmetic."""
import pytest
import torch
import torch.nn.functional as F

from conftest import maxdiff

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,N,C,H", [(3, 100, 4, 256), (1, 7, 3, 256), (2, 513, 6, 64), (5, 32, 3, 1024), (1, 1, 4, 4)])
def test_pos_hidden_fwd_bwd(B, N, C, H):
    from pointnet_refine_amd import ops
    g = torch.Generator().manual_seed(B * 100 + N + H)
    ctx = torch.randn(B, N, C, generator=g).cuda()
    w0 = (torch.randn(H, 3, generator=g) * 0.5).cuda().requires_grad_(True)
    b0 = (torch.randn(H, generator=g) * 0.1).cuda().requires_grad_(True)
    up = torch.randn(B, N, H, generator=g).cuda()
    xyz = ctx[:, :, :3]                                   # a strided view unless C == 3
    h = ops.pos_hidden(xyz, w0, b0)
    h.backward(up)
    wr, br = w0.detach().double().requires_grad_(True), b0.detach().double().requires_grad_(True)
    ref = F.relu(F.linear(xyz.double(), wr, br))
    ref.backward(up.double())
    assert h.shape == (B, N, H)
    assert maxdiff(h, ref) < 2e-6 * (1 + float(ref.abs().max()))
    rows = B * N
    assert maxdiff(w0.grad, wr.grad) < 2e-6 * rows ** 0.5 * (1 + float(wr.grad.abs().max()))
    assert maxdiff(b0.grad, br.grad) < 2e-6 * rows ** 0.5 * (1 + float(br.grad.abs().max()))


@pytest.mark.parametrize("B,M,H", [(3, 32, 256), (1, 5, 64), (7, 33, 128), (2, 32, 4)])
def test_pos_hidden_point_gradients(B, M, H):
    """The decoder's query positions carry gradients (src/model.py:209-231, no detach)."""
    from pointnet_refine_amd import ops
    g = torch.Generator().manual_seed(B + M + H)
    xyz = torch.randn(B, M, 3, generator=g).cuda().requires_grad_(True)
    w0 = (torch.randn(H, 3, generator=g) * 0.5).cuda().requires_grad_(True)
    b0 = (torch.randn(H, generator=g) * 0.1).cuda().requires_grad_(True)
    up = torch.randn(B, M, H, generator=g).cuda()
    ops.pos_hidden(xyz, w0, b0).backward(up)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (xyz, w0, b0))
    F.relu(F.linear(xr, wr, br)).backward(up.double())
    assert maxdiff(xyz.grad, xr.grad) < 1e-5 * (1 + float(xr.grad.abs().max()))
    assert maxdiff(w0.grad, wr.grad) < 1e-5 * (1 + float(wr.grad.abs().max()))
    assert maxdiff(b0.grad, br.grad) < 1e-5 * (1 + float(br.grad.abs().max()))


def test_pos_hidden_rejects_wide_point_gradients_and_bad_width():
    from pointnet_refine_amd import ops
    xyz = torch.randn(4, 3, device="cuda", requires_grad=True)
    with pytest.raises(RuntimeError, match="hidden <= 256"):
        ops.pos_hidden(xyz, torch.randn(512, 3, device="cuda"), None).sum().backward()
    with pytest.raises(RuntimeError, match="power of two"):
        ops.pos_hidden(xyz.detach(), torch.randn(96, 3, device="cuda"), None)


@pytest.mark.parametrize("rows,k,n", [(65536, 128, 3), (1, 128, 3), (1000, 256, 4), (333, 16, 1), (77, 64, 2), (5, 4, 3)])
def test_linear_small_fwd_bwd(rows, k, n):
    from pointnet_refine_amd import ops
    g = torch.Generator().manual_seed(rows + k + n)
    x = torch.randn(rows, k, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(n, generator=g).cuda().requires_grad_(True)
    up = torch.randn(rows, n, generator=g).cuda()
    y = ops.linear_small(x, w, b)
    y.backward(up)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    ref = F.linear(xr, wr, br)
    ref.backward(up.double())
    assert y.shape == (rows, n)
    assert maxdiff(y, ref) < 1e-5
    assert maxdiff(x.grad, xr.grad) < 1e-5
    assert maxdiff(w.grad, wr.grad) < 2e-6 * rows ** 0.5 * (1 + float(wr.grad.abs().max()))
    assert maxdiff(b.grad, br.grad) < 2e-6 * rows ** 0.5 * (1 + float(br.grad.abs().max()))


def test_linear_small_rejects_unsupported_shapes():
    from pointnet_refine_amd import ops
    with pytest.raises(RuntimeError, match="power of two"):
        ops.linear_small(torch.randn(8, 96, device="cuda"), torch.randn(3, 96, device="cuda"), None)
    assert not ops.linear_small_supported(128, 5) and ops.linear_small_supported(128, 3)


@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("rows,k,n,relu", [(1000, 256, 256, False), (8192, 256, 256, True), (77, 64, 132, False)])
def test_linear_with_residual(mode, rows, k, n, relu):
    from pointnet_refine_amd import _lib as L, ops
    lib = L.lib()
    old = lib.prh_get_gemm_mode()
    lib.prh_set_gemm_mode(mode)
    try:
        g = torch.Generator().manual_seed(rows + k + n)
        x = torch.randn(rows, k, generator=g).cuda().requires_grad_(True)
        w = (torch.randn(n, k, generator=g) / k ** 0.5).cuda().requires_grad_(True)
        b = torch.randn(n, generator=g).cuda().requires_grad_(True)
        r = torch.randn(rows, n, generator=g).cuda().requires_grad_(True)
        up = torch.randn(rows, n, generator=g).cuda()
        y = ops.linear(x, w, b, None, relu, r)
        y.backward(up)
        xr, wr, br, rr = (t.detach().double().requires_grad_(True) for t in (x, w, b, r))
        ref = F.linear(xr, wr, br) + rr
        if relu:
            ref = F.relu(ref)
        ref.backward(up.double())
        assert maxdiff(y, ref) < 2e-5
        assert maxdiff(r.grad, rr.grad) < 1e-6
        assert maxdiff(x.grad, xr.grad) < 2e-5
        assert maxdiff(w.grad, wr.grad) < 2e-5 * rows ** 0.5
        assert maxdiff(b.grad, br.grad) < 2e-5 * rows ** 0.5
    finally:
        lib.prh_set_gemm_mode(old)


def test_positional_encoding_module_with_residual():
    """PositionalEncoding(xyz, resid=memory) == memory + mlp(xyz) of the plain torch modules."""
    from pointnet_refine_amd.model import PositionalEncoding
    torch.manual_seed(5)
    pe = PositionalEncoding(3, 256).cuda()
    ctx = torch.randn(4, 300, 4, device="cuda")
    mem = torch.randn(4, 300, 256, device="cuda", requires_grad=True)
    out = pe(ctx[:, :, :3], resid=mem)
    up = torch.randn_like(out)
    out.backward(up)
    got = [p.grad.clone() for p in pe.parameters()] + [mem.grad.clone()]
    for p in pe.parameters():
        p.grad = None
    mem2 = mem.detach().clone().requires_grad_(True)
    ref = mem2 + pe.mlp(ctx[:, :, :3])          # torch modules: library GEMMs
    ref.backward(up)
    want = [p.grad for p in pe.parameters()] + [mem2.grad]
    assert maxdiff(out, ref) < 2e-5
    for a, b in zip(got, want):
        assert maxdiff(a, b) < 3e-5 * (1 + float(b.abs().max()))
