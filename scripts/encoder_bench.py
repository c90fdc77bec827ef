#!/usr/bin/env python
"""Kernel-tuning harness: encoder + context_proj forward/backward at (B,N) through the HIP
path, per-GEMM-launch durations from the library profiler (HIP events on the launch stream).
usage: python scripts/encoder_bench.py [B] [N] [iters]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointnet_refine_amd import _lib
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda", 0)
lib = _lib.lib()
torch.manual_seed(0)
m = LineRefineNet().to(dev).train()
ctx, noisy, target = synthetic_batch(B, N, dev)
up = torch.randn(B, N, 256, device=dev)


def step():
    for p in m.parameters():
        p.grad = None
    mem = m.encode_context(ctx)
    mem.backward(up)


step()
torch.cuda.synchronize()
lib.prh_profile_enable(4096)
t0 = time.perf_counter()
for _ in range(iters):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
agg = {}
order = []
name = C.create_string_buffer(64)
ms, fl, by = C.c_float(), C.c_double(), C.c_double()
for i in range(lib.prh_profile_count()):
    lib.prh_profile_read(i, name, 64, C.byref(ms), C.byref(fl), C.byref(by))
    k = name.value.decode()
    if k not in agg:
        agg[k] = [0, 0.0, fl.value, by.value]
        order.append(k)
    agg[k][0] += 1
    agg[k][1] += ms.value
lib.prh_profile_enable(0)
tot = 0.0
print(f"B={B} N={N}: encoder+proj fwd+bwd {dt*1e3:.2f} ms/iter = {B/dt:.0f} seg/s (encoder only)")
print(f"{'kernel':38s} {'n':>3s} {'avg ms':>9s} {'TFLOP/s':>8s} {'%peak':>6s} {'alg GB/s':>9s}")
for k in order:
    n, t, f, b = agg[k]
    a = t / n
    per_iter = n // iters
    tot += t / iters
    print(f"{k:38s} {per_iter:3d} {a:9.3f} {f/a/1e9:8.1f} {f/a/1e9/157.3*100:6.1f} {b/a/1e6:9.0f}")
flops = sum(v[2] * v[0] for v in agg.values()) / iters
print(f"GEMM launches: {tot:.2f} ms/iter of {dt*1e3:.2f}; {flops/1e12:.2f} TFLOP/iter -> {flops/tot/1e9:.1f} TFLOP/s in-GEMM, "
      f"{flops/dt/1e12:.1f} TFLOP/s end-to-end")
