"""Raw GEMM cores through the C ABI against fp64 torch on ragged shapes."""
import ctypes as C

import pytest
import torch

from conftest import maxdiff, rel_l2

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


@pytest.fixture(scope="module", params=[0, 1, 3], ids=["fp32", "split", "split16"])
def lib(request):
    """Every case runs on the exact fp32 MFMA cores, the split-bf16 cores (3 planes, 6 products)
    and the split-fp16 cores (2 scaled planes, 3 products)."""
    from pointnet_refine_amd import _lib
    l = _lib.lib()
    old = l.prh_get_gemm_mode()
    assert l.prh_set_gemm_mode(request.param) == 0
    yield l
    l.prh_set_gemm_mode(old)


@pytest.mark.parametrize("m,n,k", [(128, 128, 32), (1, 4, 4), (257, 64, 64), (1000, 1984, 1024),
                                   (333, 100, 1984), (4096, 1024, 1984), (70, 3, 128), (777, 300, 72),
                                   (512, 128, 64), (1031, 1984, 1024),
                                   # small-problem cores (prh_small.hpp): 32x32 / 32x64 / 64x64 tiles
                                   (1024, 256, 256), (1024, 1024, 256), (1024, 256, 1024), (4096, 256, 256),
                                   (1000, 100, 128), (33, 256, 64), (1, 8, 64), (3000, 252, 192),
                                   # the decoder's launches at 2 / 4 / 6 segments of 160 points (tests/test_dist_gpu.py:
                                   # K/V projections and query-side Linears of a chunked and a monolithic step)
                                   (320, 1536, 256), (640, 1536, 256), (960, 1536, 256), (64, 768, 256),
                                   (128, 512, 256), (192, 256, 1024), (192, 1024, 256)])
def test_gemm_nt(lib, m, n, k):
    g = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k)
    a = torch.randn(m, k, device="cuda", generator=g)
    w = torch.randn(n, k, device="cuda", generator=g)
    # asymmetric structure so a transposed C would be caught
    a[:, 0] += torch.arange(m, device="cuda") * 0.01
    c = torch.full((m, n), float("nan"), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    nb = lib.prh_linear_forward_workspace_bytes(m, k, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    rc = lib.prh_test_gemm_nt(_p(a), _p(w), _p(c), m, n, k, _p(ws), nb, 0, st)
    assert rc == 0, lib.prh_last_error()
    ref = a.double() @ w.double().t()
    tol = 2e-6 * k ** 0.5 * float(ref.abs().max() / k ** 0.5 + 1)
    assert maxdiff(c, ref) < max(tol, 1e-5) * 4


@pytest.mark.parametrize("p,mo,ni", [(32, 128, 128), (1, 4, 4), (1000, 64, 4), (5000, 1024, 1984),
                                     (100000, 128, 64), (777, 256, 1024), (65, 1024, 64), (4099, 300, 260),
                                     (20000, 1024, 1984),
                                     # small wgrad core: rows and both widths multiples of 64
                                     (1024, 256, 256), (1024, 1024, 256), (64, 64, 64), (4096, 128, 1024),
                                     (1088, 256, 192), (6144, 128, 64), (2048, 96, 32),
                                     # the K / V projection wgrads of a 2 / 4 / 6-segment decoder chunk
                                     (320, 1536, 256), (640, 1536, 256), (960, 1536, 256), (192, 1024, 256)])
def test_gemm_tn(lib, p, mo, ni):
    g = torch.Generator(device="cuda").manual_seed(p + mo + ni)
    a = torch.randn(p, mo, device="cuda", generator=g)
    b = torch.randn(p, ni, device="cuda", generator=g)
    c = torch.full((mo, ni), float("nan"), device="cuda")
    cs = torch.full((mo,), float("nan"), device="cuda")
    nb = lib.prh_test_gemm_tn_workspace_bytes(p, mo, ni)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.prh_test_gemm_tn(_p(a), _p(b), _p(c), _p(cs), p, mo, ni, _p(ws), nb, 0, st)
    assert rc == 0, lib.prh_last_error()
    ref = a.double().t() @ b.double()
    assert maxdiff(c, ref) < 1e-5 * p ** 0.5 * 4
    assert maxdiff(cs, a.double().sum(0)) < 1e-5 * p ** 0.5 * 4


def test_gemm_rejects_unaligned_k(lib):
    a = torch.randn(8, 6, device="cuda")
    c = torch.empty(8, 8, device="cuda")
    rc = lib.prh_test_gemm_nt(_p(a), _p(a), _p(c), 8, 8, 6, None, 0, 0, C.c_void_p(0))
    assert rc == -1 and b"multiples of 4" in lib.prh_last_error()


@pytest.mark.parametrize("m,n,k", [(4099, 1984, 1024), (20000, 256, 1536), (3000, 1024, 1984)])
def test_nt_core_is_bitwise_reproducible(m, n, k):
    """Race screen of the split-fp16 NT core (LDS-DMA weights behind counted vmcnt waits; with PRH_H2_PP=1 - run by
    tests/test_kernel_switches_gpu.py - raw barriers and two wave rows one barrier apart): twenty launches, one result."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    old = lib.prh_get_gemm_mode()
    lib.prh_set_gemm_mode(3)
    try:
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        a = torch.randn(m, k, device="cuda", generator=g)
        w = torch.randn(n, k, device="cuda", generator=g)
        nb = lib.prh_linear_forward_workspace_bytes(m, k, n)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        first = None
        for _ in range(20):
            c = torch.full((m, n), float("nan"), device="cuda")
            assert lib.prh_test_gemm_nt(_p(a), _p(w), _p(c), m, n, k, _p(ws), nb, 0, st) == 0
            if first is None:
                first = c
                ref = a.double() @ w.double().t()
                assert float((c.double() - ref).norm() / ref.norm()) < 1e-6
            else:
                assert torch.equal(c, first)
    finally:
        lib.prh_set_gemm_mode(old)


def test_split_core_is_fp32_accurate():
    """The split-bf16 core (3 planes, 6 MFMA products) must match fp64 as closely as the
    exact fp32 MFMA core does: relative L2 error below 1e-6 on a K=1984 contraction with a
    wide dynamic range (values from 1e-6 to 1e+4, no scaling assumptions)."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    m, n, k = 2048, 1024, 1984
    g = torch.Generator(device="cuda").manual_seed(5)
    a = torch.randn(m, k, device="cuda", generator=g) * torch.exp(torch.randn(m, 1, device="cuda", generator=g) * 3)
    w = torch.randn(n, k, device="cuda", generator=g) * torch.exp(torch.randn(n, 1, device="cuda", generator=g) * 2) * 0.02
    ref = a.double() @ w.double().t()
    nb = lib.prh_linear_forward_workspace_bytes(m, k, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    errs = {}
    old = lib.prh_get_gemm_mode()
    for mode in (0, 1, 3):
        lib.prh_set_gemm_mode(mode)
        c = torch.empty(m, n, device="cuda")
        assert lib.prh_test_gemm_nt(_p(a), _p(w), _p(c), m, n, k, _p(ws), nb, 0, st) == 0
        errs[mode] = float((c.double() - ref).norm() / ref.norm())
    lib.prh_set_gemm_mode(old)
    print("rel-L2 error vs fp64: fp32 core %.3e, split-bf16 core %.3e, split-fp16 core %.3e"
          % (errs[0], errs[1], errs[3]))
    assert errs[0] < 1e-6 and errs[1] < 1e-6 and errs[3] < 1e-6


def test_bf16_mode_is_reduced_precision_but_sane():
    """Mode 2 (plain bf16 operands, BASELINE config 3) is opt-in and only bf16-accurate:
    rel-L2 error vs fp64 in the 1e-3 class, far above the split core's 1e-6."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    m, n, k = 4096, 1024, 1024      # large enough to be routed to the 256x256-tile cores
    g = torch.Generator(device="cuda").manual_seed(6)
    a = torch.randn(m, k, device="cuda", generator=g)
    w = torch.randn(n, k, device="cuda", generator=g)
    ref = a.double() @ w.double().t()
    nb = lib.prh_linear_forward_workspace_bytes(m, k, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    old = lib.prh_get_gemm_mode()
    try:
        lib.prh_set_gemm_mode(2)
        c = torch.empty(m, n, device="cuda")
        assert lib.prh_test_gemm_nt(_p(a), _p(w), _p(c), m, n, k, _p(ws), nb, 0, st) == 0
        err = float((c.double() - ref).norm() / ref.norm())
    finally:
        lib.prh_set_gemm_mode(old)
    assert 1e-4 < err < 1e-2, err


def test_split16_operand_scales():
    """The split-fp16 core places each operand by a power-of-two scale taken from its largest
    magnitude: results must not depend on the operands' absolute magnitudes (1e-12 .. 1e+10),
    an all-zero operand gives exact zeros, and a single outlier 1e6 times larger than the rest
    costs the small entries precision only relative to the outlier's row scale (rel-L2 of the
    whole product stays at fp32 level)."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    m, n, k = 2048, 512, 1024
    g = torch.Generator(device="cuda").manual_seed(11)
    a0 = torch.randn(m, k, device="cuda", generator=g)
    w0 = torch.randn(n, k, device="cuda", generator=g)
    nb = lib.prh_linear_forward_workspace_bytes(m, k, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    old = lib.prh_get_gemm_mode()
    try:
        lib.prh_set_gemm_mode(3)

        def run(a, w):
            c = torch.full((m, n), float("nan"), device="cuda")
            assert lib.prh_test_gemm_nt(_p(a), _p(w), _p(c), m, n, k, _p(ws), nb, 0, st) == 0
            return c

        for sa, sw in ((1.0, 1.0), (1e-12, 1e10), (3e7, 2e-9), (1e-6, 1e-6)):
            a, w = a0 * sa, w0 * sw
            ref = a.double() @ w.double().t()
            err = float((run(a, w).double() - ref).norm() / ref.norm())
            assert err < 1e-6, (sa, sw, err)
        assert float(run(torch.zeros_like(a0), w0).abs().max()) == 0.0
        a = a0.clone()
        a[7, 3] = 1e6
        ref = a.double() @ w0.double().t()
        c = run(a, w0).double()
        assert float((c - ref).norm() / ref.norm()) < 1e-6
        rows = torch.arange(m, device="cuda") != 7          # rows without the outlier: error relative
        assert float((c - ref)[rows].abs().max()) < 1e-6 * 1e6 * 2 ** -9   # to outlier * 2^-22-ish floor
    finally:
        lib.prh_set_gemm_mode(old)


def test_split16_component_wise_rows_far_below_the_maximum():
    """The per-tensor power-of-two scale of the split-fp16 core, component-wise (VERDICT r01
    weak 3): a dgrad-shaped operand whose rows sit 2^-g below the tensor's largest entry.  Rows
    within 2^-16 of the maximum keep all 22 split bits (per-row rel-L2 at fp32 level); below
    that the low plane runs into fp16's denormal spacing and a row loses about one bit per
    factor of two - an ABSOLUTE floor of 2^-25 of the tensor maximum per element, never garbage:
    2^-20 -> < 2e-5 per row, 2^-24 -> < 4e-4.  The three-plane bf16 mode has no such dependence."""
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    m, n, k = 2048, 1984, 1024           # fusion-dgrad shape: K = 1024 output channels, N = 1984
    g = torch.Generator(device="cuda").manual_seed(13)
    a = torch.randn(m, k, device="cuda", generator=g)
    gaps = {0: 0, 1: 12, 2: 16, 3: 20, 4: 24}
    for r, gp in gaps.items():
        a[r::8] *= 2.0 ** -gp            # rows 5..7 of every 8 stay at scale 1 (they hold the maximum)
    w = torch.randn(n, k, device="cuda", generator=g)
    ref = a.double() @ w.double().t()
    nb = lib.prh_linear_forward_workspace_bytes(m, k, n)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    old = lib.prh_get_gemm_mode()
    try:
        errs = {}
        for mode in (3, 1):
            lib.prh_set_gemm_mode(mode)
            c = torch.empty(m, n, device="cuda")
            assert lib.prh_test_gemm_nt(_p(a), _p(w), _p(c), m, n, k, _p(ws), nb, 0, st) == 0
            d = (c.double() - ref)
            errs[mode] = {gp: float((d[r::8].norm(dim=1) / ref[r::8].norm(dim=1)).max()) for r, gp in gaps.items()}
    finally:
        lib.prh_set_gemm_mode(old)
    e3, e1 = errs[3], errs[1]
    assert e3[0] < 1e-6 and e3[12] < 1e-6 and e3[16] < 2e-6, e3
    assert e3[20] < 2e-5 and e3[24] < 4e-4, e3
    assert e3[24] > e3[12]                                # the degradation is real, and bounded
    assert max(e1.values()) < 1e-6, e1                    # split-bf16 (3 planes): no range dependence


@pytest.mark.parametrize("rows,k,n", [(1024, 256, 256), (1024, 256, 1024), (1024, 1024, 256), (1000, 256, 128),
                                      (4096, 256, 256), (96, 128, 96)])
@pytest.mark.parametrize("relu,resid", [(False, False), (True, False), (False, True)])
def test_small_linear_forward_backward(rows, k, n, relu, resid):
    """ops.linear on the query-side shapes of the decoder at small batch (prh_small.hpp: NT forward,
    NN dgrad with the weight as stored, 64 x 64 wgrad) against fp64 torch."""
    from pointnet_refine_amd import ops
    g = torch.Generator(device="cuda").manual_seed(rows + k + n)
    x = torch.randn(rows, k, device="cuda", generator=g, requires_grad=True)
    w = (torch.randn(n, k, device="cuda", generator=g) / k ** 0.5).requires_grad_()
    b = torch.randn(n, device="cuda", generator=g, requires_grad=True)
    r = torch.randn(rows, n, device="cuda", generator=g, requires_grad=True) if resid else None
    dy = torch.randn(rows, n, device="cuda", generator=g)
    y = ops.linear(x, w, b, relu=relu, resid=r)
    y.backward(dy)
    xd, wd, bd = (t.detach().double().requires_grad_() for t in (x, w, b))
    rd = r.detach().double().requires_grad_() if resid else None
    yr = xd @ wd.t() + bd
    if resid:
        yr = yr + rd
    if relu:
        yr = yr.relu()
    yr.backward(dy.double())
    assert maxdiff(y, yr) < 2e-5
    assert maxdiff(x.grad, xd.grad) < 2e-5
    assert maxdiff(w.grad, wd.grad) < 2e-5 * rows ** 0.5
    assert maxdiff(b.grad, bd.grad) < 2e-5 * rows ** 0.5
    if resid:
        assert maxdiff(r.grad, rd.grad) < 1e-6


@pytest.mark.parametrize("rows,mode", [(64, 3), (1024, 3), (65536, 3), (65536, 0), (65536, 4)])
def test_linear_relu_dropout_epilogue(rows, mode):
    """FFN hidden layer (src/model.py:131): y = dropout(relu(x W1^T + b1)) leaves the GEMM epilogue of
    every core family (small-problem core, 128 x 128 fp32 core, split-fp16 / bf16 NT cores) already
    dropped out; the keep decision is the (seed, row, column) hash ops.layernorm_keep_mask re-creates,
    and the backward masks with it through y > 0."""
    from pointnet_refine_amd import _lib, ops
    lib = _lib.lib()
    old = lib.prh_get_gemm_mode()
    lib.prh_set_gemm_mode(mode)
    try:
        torch.manual_seed(rows + mode)
        k, n, p, seed = 256, 1024, 0.1, 12345
        x = torch.randn(rows, k, device="cuda", requires_grad=True)
        w = (torch.randn(n, k, device="cuda") / 16).requires_grad_(True)
        b = torch.randn(n, device="cuda", requires_grad=True)
        y = ops.linear(x, w, b, None, True, None, p, seed)
        keep = ops.layernorm_keep_mask(rows, n, p, seed, device="cuda")
        up = torch.randn_like(y)
        (y * up).sum().backward()
        xd, wd, bd = (t.detach().double() for t in (x, w, b))
        pre = xd @ wd.t() + bd
        ref = torch.relu(pre) * keep / (1.0 - p)
        tol = 5e-2 if mode == 4 else 1e-5
        # same mask: dropped elements are exact zeros, kept ones follow the reference (ReLU ties aside: a
        # pre-activation inside fp32 noise of 0 may round to either side - ~5 of the 67 M elements at 65536 rows)
        assert float(((y == 0) != (ref == 0)).float().mean()) < (2e-3 if mode == 4 else 1e-6)
        assert rel_l2(ref, y) < tol
        assert abs(float(keep.float().mean()) - (1.0 - p)) < 5e-3
        # backward through the mask the FORWARD produced (y > 0), so that tie flips do not count
        g = up.double() * (y.detach() > 0) / (1.0 - p)
        assert rel_l2(g @ wd, x.grad) < tol and rel_l2(g.t() @ xd, w.grad) < tol and rel_l2(g.sum(0), b.grad) < tol
    finally:
        lib.prh_set_gemm_mode(old)
