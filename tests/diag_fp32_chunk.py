"""under PRH_GEMM=fp32 the chunked-decoder gradient differed from the monolithic
one by 4.2e-4 (default cores 5.7e-6).  One process, the product model: which tensors differ, in which
GEMM mode, and does the forward already differ?  python tests/diag_fp32_chunk.py  (GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet_refine_amd import _lib
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch
from pointnet_refine_amd.train_step import TrainStep

lib = _lib.lib()
dev = torch.device("cuda", 0)


PROC = os.environ.get("PRH_DIAG_PROC") == "1"      # the procedural weights / inputs of tests/test_dist_gpu.py


def make():
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
    return m


B, N = 6, 160
batch = synthetic_batch(B, N, dev, seed=50)
if PROC:
    from oracle import procedural as P
    batch = [t.to(dev) for t in P.synth_batch(B, N, 4, 32, seed=50)]
SHIFT = float(os.environ.get("PRH_DIAG_SHIFT", "0"))      # move the target away from the predictions: no sign(0) ambiguity
batch = [batch[0], batch[1], batch[2] + SHIFT]
for mode in ((0,) if os.environ.get('PRH_DIAG_MODE0') else (0, 3)):
    lib.prh_set_gemm_mode(mode)
    res = {}
    for tag, chunk in (("mono", None), ("chunk4", 4), ("chunk2", 2)):
        m = make()
        st = TrainStep(m, torch.optim.SGD(m.parameters(), lr=0.0), decoder_chunk=chunk)
        st.keep_out = True
        st.grads.zero()
        st.forward_backward(*batch)
        torch.cuda.synchronize()
        res[tag] = (st.last_out.clone(), st.grads.flat.clone(), [n for n, p in m.named_parameters()],
                    [(o, p.numel()) for o, p in zip(st.grads.offsets, st.grads.params)])
        st.close()
    for tag in ("chunk4", "chunk2"):
        o0, g0, names, sizes = res["mono"]
        o1, g1, _, _ = res[tag]
        tgt = batch[2].unsqueeze(0)
        flips = int(((o0 - tgt).sign() != (o1 - tgt).sign()).sum())
        near = int(((o0 - tgt).abs() < 1e-4).sum())
        print(f"mode {mode} {tag}: out max|d| {float((o0 - o1).abs().max()):.3e}  all grads rel-L2 {float((g0 - g1).norm() / g0.norm()):.3e}"
              f"  L1 sign flips between the two runs: {flips} of {o0.numel()} (|pred - target| < 1e-4: {near})")
        rows = []
        for n, (off, k) in zip(names, sizes):
            a, b = g0[off:off + k].double(), g1[off:off + k].double()
            rows.append((float((a - b).norm() / max(float(a.norm()), 1e-30)), n, float(a.norm())))
        gmax = max(r[2] for r in rows)
        rows = [r for r in rows if r[2] > 1e-7 * gmax]          # tensors that receive a gradient at all
        for r, n, nn in sorted(rows, reverse=True)[:10]:
            print(f"      {n:48s} rel-L2 {r:.3e}  |g| {nn:.3e}")
lib.prh_set_gemm_mode(3)
