"""bf16 mode (prh_set_gemm_mode(4); BASELINE config 3 "B=4096, N=1024 bf16 training"): bf16 MFMA
operands AND bf16 activation storage in the encoder (csrc/prh_b16.hpp).

Two kinds of checks:
  * the kernels compute what they say - against fp64 arithmetic on the SAME bf16-rounded
    operands the errors are at fp32-accumulation level (1e-5), which pins indexing, swizzles,
    prologues, epilogues and tails independently of the precision question;
  * the mode as a whole against the reference's fp32 results (oracle, golden vectors G1-G3):
    the looser gate SURVEY 8(d) states for reduced precision - max-abs on `out` <= 5e-2 - and
    the gradient error the mode really has, stated below per test.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import maxdiff, rel_l2
from oracle import linerefine_oracle as O
from oracle import procedural as P

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


@pytest.fixture()
def bf16_mode():
    from pointnet_refine_amd import _lib
    lib = _lib.lib()
    old = lib.prh_get_gemm_mode()
    assert lib.prh_set_gemm_mode(4) == 0
    yield lib
    lib.prh_set_gemm_mode(old)


def _bf(t):
    return t.to(torch.bfloat16)


# K % 64 == 0: the plain-operand launches run on the DMA + phase-split core (gemm_nt_b16d_kernel: 1, 2, 3, 16 and 31
# k-tiles, ragged row tiles, N off the 256-column tile); K = 8 / 72 stay on the register-staged core
@pytest.mark.parametrize("rows,k,n", [(1000, 1024, 256), (4099, 64, 1984), (257, 8, 64), (2048, 1984, 1024), (70000, 256, 1536),
                                      (515, 128, 264), (300, 192, 72), (1, 64, 64), (777, 72, 128)])
def test_linear_bf16_kernels_are_exact_on_rounded_operands(bf16_mode, rows, k, n):
    lib = bf16_mode
    g = torch.Generator(device="cuda").manual_seed(rows + k + n)
    x = _bf(torch.randn(rows, k, device="cuda", generator=g))
    w = torch.randn(n, k, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(n, device="cuda", generator=g)
    dy = torch.randn(rows, n, device="cuda", generator=g)
    y = torch.full((rows, n), float("nan"), device="cuda")
    nb = lib.prh_linear_bf16_workspace_bytes(rows, k, n, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.prh_linear_forward_bf16(_p(x), k, _p(w), _p(b), _p(y), rows, k, n, 0, _p(ws), nb, 0, st) == 0
    wr, dyr = _bf(w).double(), _bf(dy).double()
    ref = x.double() @ wr.t() + b.double()
    assert float((y.double() - ref).norm() / ref.norm()) < 2e-6
    assert float((y.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max())
    dx = torch.full((rows, k), float("nan"), device="cuda", dtype=torch.bfloat16)
    dw = torch.full((n, k), float("nan"), device="cuda")
    db = torch.full((n,), float("nan"), device="cuda")
    assert lib.prh_linear_backward_bf16(_p(x), k, _p(w), _p(dy), _p(dx), _p(dw), _p(db), rows, k, n, _p(ws), nb, 0, st) == 0
    rdx = dyr @ wr                                   # stored in bf16: one rounding, 2^-9 relative
    assert float((dx.double() - rdx).norm() / rdx.norm()) < 3e-3
    assert torch.equal(dx, _bf(rdx.float())) or float((dx.double() - _bf(rdx.float()).double()).abs().max()) <= \
        2.0 ** -7 * float(rdx.abs().max())          # at most one bf16 ulp apart (accumulation order)
    rdw = dyr.t() @ x.double()
    assert float((dw.double() - rdw).norm() / rdw.norm()) < 2e-6
    assert float((db.double() - dyr.sum(0)).abs().max()) < 1e-4 * float(dyr.sum(0).abs().max()) + 1e-4


@pytest.mark.parametrize("rows,k,n", [(4099, 1024, 1984), (70000, 256, 1536), (2048, 1984, 1024)])
def test_dma_core_is_bitwise_reproducible(bf16_mode, rows, k, n):
    """Race screen of the DMA + phase-split core (counted vmcnt waits, raw barriers, two wave rows one barrier apart):
    the same launch twenty times must give the same bits - an LDS image read before its DMA has landed, or overwritten
    before its last read, shows up as a run that differs."""
    lib = bf16_mode
    g = torch.Generator(device="cuda").manual_seed(rows + k)
    x = _bf(torch.randn(rows, k, device="cuda", generator=g))
    w = torch.randn(n, k, device="cuda", generator=g) / k ** 0.5
    b = torch.randn(n, device="cuda", generator=g)
    nb = lib.prh_linear_bf16_workspace_bytes(rows, k, n, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    first = None
    for _ in range(20):
        y = torch.full((rows, n), float("nan"), device="cuda")
        assert lib.prh_linear_forward_bf16(_p(x), k, _p(w), _p(b), _p(y), rows, k, n, 0, _p(ws), nb, 0, st) == 0
        if first is None:
            first = y
            ref = x.double() @ _bf(w).double().t() + b.double()
            assert float((y.double() - ref).norm() / ref.norm()) < 2e-6
        else:
            assert torch.equal(y, first)


def test_generic_linear_in_bf16_mode_runs_on_the_bf16_core(bf16_mode):
    """fp32-storage Linears (decoder) in mode 4: input converted in flight, same one-product core."""
    from pointnet_refine_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.randn(5000, 256, device="cuda", generator=g).requires_grad_(True)
    w = (torch.randn(1536, 256, device="cuda", generator=g) / 16).requires_grad_(True)
    b = torch.randn(1536, device="cuda", generator=g).requires_grad_(True)
    r = torch.randn(5000, 1536, device="cuda", generator=g)
    y = ops.linear(x, w, b, None, True, r)
    ref = torch.relu(_bf(x).double() @ _bf(w).double().t() + b.double() + r.double())
    assert float((y.double() - ref).norm() / ref.norm()) < 2e-6
    y.backward(torch.ones_like(y))
    full = torch.relu(x.double() @ w.double().t() + b.double() + r.double())
    assert 1e-4 < float((y.double() - full).norm() / full.norm()) < 1e-2      # really one bf16 product
    assert x.grad is not None and w.grad is not None and torch.isfinite(w.grad).all()


def _pre_bn_bias(k):
    return bool(re.search(r"(conv\d\.bias|fusion\.0\.bias|point_mlp\.[036]\.bias)$", k))


@pytest.mark.parametrize("mode", ["eval", "train"])
def test_encoder_bf16_vs_oracle_g3(bf16_mode, golden_dir, mode):
    """Encoder alone, both returns used (G3 incl. the dead max-pool channel), random upstream
    gradients over only 768 points (little averaging).  Yardstick: the reference encoder under
    torch.autocast(bfloat16) on the same inputs (g9 fixture: train fused rel-L2 1.6e-2, dx 0.19,
    gradient heads median 0.13 / worst 0.23; eval 5e-3, 0.10, 0.073 / 0.29).  Gates: each figure
    <= 1.25 x the reference's own bf16 figure (+ a floor of 3e-2 on the forward: pre-BN
    activations are STORED in bf16 here, the autocast BatchNorm reads them the same way);
    gradient norms within 5 %, running statistics within 1e-2.  Measured: train 1.5e-2, 0.18,
    0.127 / 0.18; eval 1.5e-2 (gfeat 4e-3), 0.10, 0.069 / 0.12."""
    g9 = np.load(os.path.join(golden_dir, "g9_bf16_autocast.npz"))
    y = lambda name: float(g9[f"enc_{mode}_{name}"])
    from pointnet_refine_amd.model import MultiScalePointNetEncoder
    B, N, Cc = 4, 192, 4
    g = np.load(os.path.join(golden_dir, "g3_encoder_c4_train.npz"))
    sd = P.encoder_state_dict(Cc, 1024, seed=3)
    sd["fusion.1.weight"][5] = 0.0
    sd["fusion.1.bias"][5] = -1.0
    ctx, _, _ = P.synth_batch(B, N, Cc, 32, seed=77)
    r = np.random.default_rng(5)
    up_g = torch.from_numpy(r.normal(0, 1, (B, 2048)).astype(np.float32))
    up_f = torch.from_numpy(r.normal(0, 1, (B, N, 1024)).astype(np.float32))
    m = MultiScalePointNetEncoder(in_channel=Cc, out_dim=1024)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train(mode == "train")
    x = ctx.cuda().requires_grad_(True)
    gf, fu_cm = m(x.transpose(2, 1))
    assert fu_cm.dtype == torch.float32 and fu_cm.shape == (B, 1024, N)
    ((gf * up_g.cuda()).sum() + (fu_cm.transpose(2, 1) * up_f.cuda()).sum()).backward()
    assert rel_l2(g[f"{mode}::gfeat"], gf) < max(3e-2, 1.25 * y("gfeat_rel_l2"))
    assert rel_l2(g[f"{mode}::fused_sub"], fu_cm.transpose(2, 1)[:, ::8, ::16]) < max(3e-2, 1.25 * y("fused_rel_l2"))
    assert rel_l2(g[f"{mode}::dx"], x.grad) < 1.25 * y("dx_rel_l2")
    named = dict(m.named_parameters())
    rels = {}
    for k, nrm in zip(g[f"{mode}::grad_keys"], g[f"{mode}::grad_norms"]):
        k = str(k)
        if mode == "train" and _pre_bn_bias(k):
            assert float(named[k].grad.abs().max()) < 0.3       # analytically zero; bf16 dz keeps O(2^-9 * |dy|) of it
            continue
        assert abs(float(named[k].grad.double().norm()) - nrm) <= 5e-2 * nrm + 1e-9, k
        rels[k] = rel_l2(g[f"{mode}::gh::{k}"], named[k].grad.reshape(-1)[:64])
    assert max(rels.values()) < 1.25 * y("grad_head_rel_l2_worst"), (max(rels, key=rels.get), max(rels.values()))
    assert float(np.median(list(rels.values()))) < 1.25 * y("grad_head_rel_l2_median")
    if mode == "train":
        msd = m.state_dict()
        for k in msd:
            if "running" in k:
                v = torch.from_numpy(g["st::" + k])
                assert maxdiff(msd[k], v) <= 1e-2 * float(v.double().abs().max()) + 1e-4, k


def _model(sd):
    from pointnet_refine_amd.model import LineRefineNet
    m = LineRefineNet()
    m.load_state_dict(sd, strict=True)
    return m.cuda()


def test_model_bf16_eval_g1(bf16_mode, golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_eval_forward.npz"))
    m = _model(P.linerefine_state_dict(0)).eval()
    ctx, noisy, _ = P.synth_batch(8, 256, 4, 32, seed=1234)
    with torch.no_grad():
        out = m(ctx.cuda(), noisy.cuda())
    assert maxdiff(out, g["out"]) < 5e-2                       # SURVEY 8(d) reduced-precision gate


def test_model_bf16_training_step_vs_g2(bf16_mode, golden_dir):
    """Config 3's arithmetic against the reference's train-mode forward+backward (G2, dropout
    off).  Yardstick: the REFERENCE ITSELF under torch.autocast(bfloat16) on the same inputs and
    weights (tests/golden/g9_bf16_autocast.npz, oracle/make_golden_bf16.py): `out` max-abs 0.397 /
    rel-L2 5.9e-2; gradients, on the 64-entry heads the G2 fixture keeps: rel-L2 median 6.8e-2,
    90th percentile 0.28, worst 1.55 - SURVEY 8(d)'s "5e-2" is that measurement in eval mode
    (1.7e-2 rel-L2, 7.9e-2 max-abs).  This mode must be at least as close to the fp32 reference
    as the reference's own bf16 run: `out` rel-L2 <= 5e-2 and max-abs <= 0.2 (measured: 2.4e-2 /
    0.125), loss within 2e-2, gradient-head rel-L2 median / 90th percentile / worst <= the
    autocast figures (measured: 5.7e-2 / - / 1.50 with the K / V buffers in bf16; 5.1e-2 / - / 0.76
    before that)."""
    g = np.load(os.path.join(golden_dir, "g2_train_fwd_bwd.npz"))
    g9 = np.load(os.path.join(golden_dir, "g9_bf16_autocast.npz"))
    m = _model(P.linerefine_state_dict(0)).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    ctx, noisy, target = P.synth_batch(8, 256, 4, 32, seed=1234)
    nl = noisy.cuda().requires_grad_(True)
    out = m(ctx.cuda(), nl)
    loss = sum(torch.nn.functional.l1_loss(out[l], target.cuda()) for l in range(6)) / 6
    loss.backward()
    assert rel_l2(g["out"], out) < 5e-2
    assert maxdiff(out, g["out"]) < float(g9["train_out_maxabs"])
    assert maxdiff(out, g["out"]) < 0.2                       # measured 0.125
    assert abs(float(loss) - float(g["loss"])) < 2e-2 * float(g["loss"])
    assert rel_l2(g["dnoisy"], nl.grad) < 0.2
    named = dict(m.named_parameters())
    rels = {}
    for k, nrm in zip(g["grad_keys"], g["grad_norms"]):
        k = str(k)
        if _pre_bn_bias(k):
            continue
        gr = named[k].grad.reshape(-1).double()
        assert torch.isfinite(gr).all(), k
        rels[k] = rel_l2(g["gh::" + k], gr[:64])
    worst = max(rels, key=rels.get)
    print(f"bf16 mode vs G2: out {maxdiff(out, g['out']):.3e}, grad rel-L2 median {np.median(list(rels.values())):.3e}, "
          f"worst {rels[worst]:.3e} ({worst})")
    vals = list(rels.values())
    assert float(np.median(vals)) < float(g9["train_grad_head_rel_l2_median"])
    assert float(np.quantile(vals, 0.9)) < float(g9["train_grad_head_rel_l2_p90"])
    # (the worst head is the same tensor for both: the first rows of layer 0's cross-attention
    # in_proj_weight, a near-cancelling sum; bf16 K/V storage puts this mode at 1.50 against 1.55)
    assert rels[worst] < float(g9["train_grad_head_rel_l2_worst"]), (worst, rels[worst])


def test_train_step_bf16_mode_learns(bf16_mode):
    """TrainStep in mode 4 (chunked decoder, fused Adam): finite, and the loss goes down."""
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    torch.manual_seed(5)
    m = LineRefineNet().cuda().train()
    ctx, noisy, target = synthetic_batch(32, 512, torch.device("cuda", 0), seed=9)
    step = TrainStep(m, None, decoder_chunk=16)
    losses = [float(step(ctx, noisy, target)) for _ in range(12)]
    assert all(np.isfinite(losses)), losses
    assert min(losses[-3:]) < 0.8 * losses[0], losses


def test_model_bf16_every_gradient_vs_fp64_oracle(bf16_mode):
    """absolute gates, on WHOLE tensors, against the oracle in fp64 on the G2 inputs
    (B=8, N=256, train mode, dropout off) - not "no worse than the reference under autocast" on 64-entry
    heads.  Only 2,048 context points stand behind every BatchNorm statistic and every weight gradient
    here, so the figures are far noisier than at the benchmark's 4.19 M points (worst tensor 1.9e-2 there:
    bench.py's at_init gate).  Measured on this fixture (r03): out rel-L2 2.8e-2; per-tensor gradient rel-L2
    median 4.3e-2, 90th percentile 0.14, worst 0.22 (decoder_layers.1.self_attn.in_proj_weight); the gates
    sit ~1.5x above that."""
    from oracle import linerefine_oracle as O
    sd = P.linerefine_state_dict(0)
    m = _model(sd).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    ctx, noisy, target = P.synth_batch(8, 256, 4, 32, seed=1234)
    out = m(ctx.cuda(), noisy.cuda())
    loss = sum(torch.nn.functional.l1_loss(out[l], target.cuda()) for l in range(6)) / 6
    loss.backward()
    p = O.as_params(sd, dtype=torch.float64, requires_grad=True)
    o64 = O.linerefine_forward(p, ctx.double(), noisy.double(), training=True, new_stats={})
    O.deep_supervision_l1(o64, target.double()).backward()
    rels = {k: rel_l2(p[k].grad, v.grad) for k, v in m.named_parameters()
            if not _pre_bn_bias(k)}                  # (analytically zero in front of a train-mode BatchNorm)
    vals = sorted(rels.values())
    worst = max(rels, key=rels.get)
    print(f"bf16 mode vs fp64 oracle (G2 inputs): out rel-L2 {rel_l2(o64, out):.3e}; per-tensor gradient rel-L2 median "
          f"{vals[len(vals) // 2]:.3e}, p90 {vals[int(len(vals) * 0.9)]:.3e}, worst {rels[worst]:.3e} ({worst})")
    assert rel_l2(o64, out) < 4e-2
    assert vals[len(vals) // 2] < 6e-2
    assert vals[int(len(vals) * 0.9)] < 2.2e-1
    assert rels[worst] < 3.5e-1, (worst, rels[worst])
