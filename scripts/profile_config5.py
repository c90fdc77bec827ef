"""Config-5 forward (fused fp16 encoder kernel + bf16 decoder with the folded cross-attention) for a kernel profile:
rocprofv3 --kernel-trace --stats -- python scripts/profile_config5.py [B] [N] [batch] [reps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointnet_refine_amd import ops
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = LineRefineNet().to(dev).eval()
ctx, noisy, _ = synthetic_batch(B, N, dev)
m.context_encoder.inference_precision = "fp16"


def fwd():
    with torch.no_grad(), ops.gemm_mode_scope("bf16"):
        return [m(ctx[s:s + bs], noisy[s:s + bs]) for s in range(0, B, bs)]


fwd()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fwd()
torch.cuda.synchronize()
print(f"config-5 forward, B={B} N={N} batches of {bs}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms")
