"""Is the first forward/backward of a process bitwise equal to the second (same inputs, same weights)?
A difference means some kernel reads workspace it did not write.  Also poisons the workspace with NaN."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet_refine_amd import _lib, ops
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch
from pointnet_refine_amd.train_step import TrainStep

lib = _lib.lib()
dev = torch.device("cuda", 0)
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B, N = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (6, 160)
lib.prh_set_gemm_mode(mode)


def make():
    torch.manual_seed(3)
    m = LineRefineNet().to(dev).train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return m


batch = synthetic_batch(B, N, dev, seed=50)
runs = []
for it in range(4):
    if it == 2:      # poison every cached workspace: a kernel that reads what it did not write shows NaN / a change
        for t in ops._workspaces.values():
            t.view(torch.float32)[: t.numel() // 4].fill_(float("nan")) if t.numel() % 4 == 0 else t.fill_(255)
    m = make()
    st = TrainStep(m, torch.optim.SGD(m.parameters(), lr=0.0), decoder_chunk=None)
    st.keep_out = True
    st.grads.zero()
    with torch.no_grad():
        mem = m.encode_context(batch[0]).clone() if False else None
    st.forward_backward(*batch)
    torch.cuda.synchronize()
    runs.append((st.last_out.clone(), st.grads.flat.clone()))
    st.close()
for i in range(1, 4):
    o0, g0 = runs[0]
    o1, g1 = runs[i]
    print(f"mode {mode} B={B} N={N} run {i} vs run 0: out max|d| {float((o0 - o1).abs().max()):.3e}  grads rel-L2 {float((g0 - g1).norm() / g0.norm()):.3e}"
          f"  nan in out {bool(torch.isnan(o1).any())} nan in grads {bool(torch.isnan(g1).any())}")
o1, g1 = runs[1]
o2, g2 = runs[3]
print(f"run 3 vs run 1: out max|d| {float((o2 - o1).abs().max()):.3e} grads rel-L2 {float((g2 - g1).norm() / g1.norm()):.3e}")
