"""Where the cycles of gemm_nt_b16d_kernel go (diagnostic build of the library, -DPRH_STAMP; GPU box).
usage: PRH_LIB_PATH=pointnet_refine_amd/libprh_stamp.so [PRH_GEMM=bf16] python scripts/diag_b16d_stamps.py [B] [N]
(bf16 mode: gemm_nt_b16d_kernel per phase; default mode: prologue / k-loop / epilogue of four gemm_nt_h2_kernel launches)
Prints, for one workgroup of the fusion dgrad (K=1024, N=1984) and one wave of each wave row, the s_memtime
ticks (shader-clock cycles) accumulated in each part of the k-loop."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointnet_refine_amd import _lib
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda", 0)
lib = _lib.lib()
lib.prh_debug_stamp_buffer.argtypes = [C.c_void_p]
lib.prh_debug_stamp_buffer.restype = C.c_int
buf = torch.zeros(128, dtype=torch.int32, device=dev)
assert lib.prh_debug_stamp_buffer(C.c_void_p(buf.data_ptr())) == 0
torch.manual_seed(0)
m = LineRefineNet().to(dev).train()
ctx, noisy, target = synthetic_batch(B, N, dev)
up = torch.randn(B, N, 256, device=dev)
for _ in range(3):
    for p in m.parameters():
        p.grad = None
    mem = m.encode_context(ctx)
    mem.backward(up)
    del mem
for p in m.parameters():
    p.grad = None
out = m(ctx, noisy)                 # the whole model once: the K / V projections' launches
out.sum().backward()
del out
torch.cuda.synchronize()
t = buf.cpu().numpy().astype("int64") & 0xFFFFFFFF
names = []
for mp in range(4):
    names += [f"phase {mp} memory", f"phase {mp} barrier A", f"phase {mp} MFMA issue", f"phase {mp} barrier B"]
names += ["prologue", "loop exit", "epilogue (issue)", "store drain (vmcnt 0)"]
for row in range(2):
    o = t[row * 32:row * 32 + 32]
    kt = int(o[24])
    tot = int(o[:20].sum())
    if tot == 0:
        continue
    print(f"wave row {row}: k-tiles {kt}, total {tot} cycles (s_memtime)")
    for i, nm in enumerate(names):
        v = int(o[i])
        per = f"{v / kt:8.1f} per k-tile" if i < 16 and kt else ""
        print(f"   {nm:26s} {v:8d} ticks  {100.0 * v / max(tot, 1):5.1f} %  {per}")

# split-fp16 NT core (any mode but bf16): prologue / k-loop / epilogue / store drain of one workgroup per launch kind
kinds = ["fusion dgrad K=1024 N=1984", "fusion forward K=1984 N=1024", "K/V projection forward K=256 N=1536", "conv5 forward K=512 N=1024"]
for w, nm in enumerate(kinds):
    for row in range(2):
        o = t[64 + w * 16 + row * 8: 64 + w * 16 + row * 8 + 8]
        tot = int(o[:4].sum())
        if tot == 0:
            continue
        print(f"h2 {nm}, wave row {row}: {int(o[4])} k-tiles, {tot} cycles: prologue {int(o[0])} ({100.0 * o[0] / tot:.1f} %), k-loop {int(o[1])} "
              f"({100.0 * o[1] / tot:.1f} %, {o[1] / max(int(o[4]), 1):.0f} per k-tile), epilogue {int(o[2])} ({100.0 * o[2] / tot:.1f} %), store drain {int(o[3])} ({100.0 * o[3] / tot:.1f} %)")
