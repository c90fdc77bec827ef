"""Which tensors of the ragged case (B=3, N=77, M=20) differ from the fp64 oracle, per GEMM mode?"""
import os, sys, re
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import linerefine_oracle as O
from oracle import procedural as P
from pointnet_refine_amd import _lib
from pointnet_refine_amd.model import LineRefineNet
lib = _lib.lib()
B, N, M = 3, 77, 20
sd = P.linerefine_state_dict(0)
ctx, noisy, target = P.synth_batch(B, N, 4, M, seed=B * 100 + N + M)
p = O.as_params(sd, dtype=torch.float64, requires_grad=True)
o64 = O.linerefine_forward(p, ctx.double(), noisy.double(), training=True, new_stats={})
(o64 - target.double().unsqueeze(0)).abs().mean().backward()
def rel(a, b): return float((a.double().cpu() - b.double().cpu()).norm() / (a.double().norm() + 1e-30))
for mode in (0, 1, 3):
    lib.prh_set_gemm_mode(mode)
    m = LineRefineNet(); m.load_state_dict(sd, strict=True); m = m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention): mod.dropout = 0.0
    out = m(ctx.cuda(), noisy.cuda())
    (out - target.cuda().unsqueeze(0)).abs().mean().backward()
    rows = sorted(((rel(p[k].grad, v.grad), k) for k, v in m.named_parameters()
                   if not re.search(r"(conv\d\.bias|fusion\.0\.bias|point_mlp\.[036]\.bias)$", k)), reverse=True)[:5]
    sign = int(((out.detach().cpu().double() - target.double()).sign() != (o64.detach() - target.double()).sign()).sum())
    print(f"mode {mode}: out max|d| {float((out.detach().cpu().double() - o64.detach()).abs().max()):.2e}; L1 sign flips vs oracle {sign}; worst: "
          + ", ".join(f"{k} {r:.2e}" for r, k in rows))
lib.prh_set_gemm_mode(3)
