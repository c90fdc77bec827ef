"""Where does the mode-0 (exact fp32 cores) gradient of a decoder chunk depend on its batch-mates?
(tests/diag_fp32_chunk.py: chunked vs monolithic gradients differ by 5e-4 in mode 0, 6e-6 in mode 3.)
Attention kernel alone, then the decoder alone, segments 0..3 computed inside a batch of 6 and alone."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet_refine_amd import _lib, ops
from pointnet_refine_amd.model import LineRefineNet

lib = _lib.lib()
dev = torch.device("cuda", 0)
PROC = os.environ.get("PRH_DIAG_PROC") == "1"


def rel(a, b):
    return float((a.double() - b.double()).norm() / (a.double().norm() + 1e-30))


for mode in (0, 3):
    lib.prh_set_gemm_mode(mode)
    torch.manual_seed(0)
    B, M, N, C, H = 6, 32, 160, 256, 8
    q = torch.randn(B, M, C, device=dev)
    kall = torch.randn(B, N, 6 * C, device=dev)
    vall = torch.randn(B, N, 6 * C, device=dev)
    up = torch.randn(B, M, C, device=dev)
    res = []
    for sl in (slice(0, 6), slice(0, 4)):
        qq = q[sl].clone().requires_grad_(True)
        kk = kall[sl].clone().requires_grad_(True)
        vv = vall[sl].clone().requires_grad_(True)
        tok, arena = ops.kv_token(kk, vv, C)
        o = ops.attention_block(qq, kk, vv, tok, arena, 2, H, 0.0, 0)
        (o * up[sl]).sum().backward()
        res.append((o.detach()[:4], qq.grad[:4], kk.grad[:4], vv.grad[:4]))
    print(f"mode {mode} attention alone, segments 0..3 in a batch of 6 vs alone: o {rel(*[r[0] for r in res]):.2e} "
          f"dq {rel(*[r[1] for r in res]):.2e} dk {rel(*[r[2] for r in res]):.2e} dv {rel(*[r[3] for r in res]):.2e}")
    # decoder alone
    torch.manual_seed(3)
    m = LineRefineNet()
    if PROC:
        from oracle import procedural as P
        m.load_state_dict(P.linerefine_state_dict(0), strict=True)
    m = m.to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    from pointnet_refine_amd.synth import synthetic_batch
    ctx, noisy, target = synthetic_batch(B, N, dev, seed=50)
    if PROC:
        ctx, noisy, target = [t.to(dev) for t in P.synth_batch(B, N, 4, 32, seed=50)]
    with torch.no_grad():
        memory = m.encode_context(ctx)
        tgt0 = m.encode_line(noisy)
    # record the output of every Linear with a fused ReLU (FFN hidden layers, regression-head hidden layers)
    relu_outs = []
    orig_linear = ops.linear

    def spy(x, w, b=None, x_amax=None, relu=False, resid=None, dropout_p=0.0, seed=0):
        y = orig_linear(x, w, b, x_amax, relu, resid, dropout_p, seed)
        if relu:
            relu_outs.append(y.detach())
        return y
    ops.linear = spy
    orig_ph = ops.pos_hidden

    def spy_ph(xyz, w0, b0=None):
        h = orig_ph(xyz, w0, b0)
        if h.shape[-2] == 32:            # the query-side positional MLP (the memory side is the same in both runs)
            relu_outs.append(h.detach())
        return h
    ops.pos_hidden = spy_ph
    for lo, hi in ((0, 4), (4, 6), (2, 4)):
        outs = []
        masks = []
        for sl in (slice(0, 6), slice(lo, hi)):
            relu_outs.clear()
            mem = memory[sl].clone().requires_grad_(True)
            t0 = tgt0[sl].clone().requires_grad_(True)
            for p_ in m.parameters():
                p_.grad = None
            out = m.decode(ctx[sl], noisy[sl], mem, t0)
            pick = slice(lo, hi) if sl.stop - sl.start == 6 else slice(0, hi - lo)
            w = torch.linspace(0.5, 1.5, out[:, pick].numel(), device=dev).view_as(out[:, pick])
            (out[:, pick] * w).sum().backward()          # smooth loss on the chosen segments only
            grads = {n: p_.grad.clone() for n, p_ in m.named_parameters() if p_.grad is not None}
            outs.append((out.detach()[:, pick], mem.grad[pick].clone(), t0.grad[pick].clone(), grads))
            masks.append([(y.reshape(sl.stop - sl.start, -1, y.shape[-1])[pick] > 0) for y in relu_outs])
        flips = sum(int((a != b).sum()) for a, b in zip(*masks))
        total = sum(a.numel() for a in masks[0])
        print(f"      ReLU masks (FFN, head and query-side positional hidden layers) that differ between the two evaluations: {flips} of {total}")
        worst = sorted(((rel(outs[0][3][n], outs[1][3][n]), n) for n in outs[0][3]), reverse=True)[:4]
        print(f"mode {mode} decoder alone, segments {lo}..{hi - 1} in a batch of 6 vs alone: out max|d| {float((outs[0][0] - outs[1][0]).abs().max()):.2e}  "
              f"d_memory {rel(outs[0][1], outs[1][1]):.2e}  d_tgt0 {rel(outs[0][2], outs[1][2]):.2e}  worst params: " + ", ".join(f"{n} {r:.1e}" for r, n in worst))
    ops.linear = orig_linear
    ops.pos_hidden = orig_ph
lib.prh_set_gemm_mode(3)
