#!/usr/bin/env python
"""Inference-side measurement of the encoder: context (B,N,4) -> memory
(B,N,256) in eval mode, per-layer kernels against the fused kernel (csrc/prh_fused.hpp), both
precisions; then the whole eval forward.  Reports ms, segments/s, algorithmic TFLOP/s (2 x
3,055,936 MAC per point for encoder + context_proj) and HBM bytes per point by design.
usage: python scripts/infer_bench.py [B] [N] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pointnet_refine_amd import _lib
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
bs = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = torch.device("cuda", 0)
lib = _lib.lib()
torch.manual_seed(0)
m = LineRefineNet().to(dev).eval()
ctx, noisy, _ = synthetic_batch(B, N, dev)
flop_pt = 2.0 * (2793792 + 262144)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def enc_all():
    with torch.no_grad():
        for s in range(0, B, bs):
            m.encode_context(ctx[s:s + bs])


def full_all():
    with torch.no_grad():
        for s in range(0, B, bs):
            m(ctx[s:s + bs], noisy[s:s + bs])


print(f"B={B} N={N} batches of {bs}; encoder + context_proj = {flop_pt * B * N / 1e12:.2f} TFLOP per pass")
ref = None
for prec, mode, what in ((None, 3, "per-layer kernels, split-fp16 (fp32-accurate)"),
                         ("fp32", 3, "FUSED kernel, 2 fp16 planes (fp32-accurate)"),
                         ("fp16", 3, "FUSED kernel, 1 fp16 plane")):
    m.context_encoder.inference_precision = prec
    lib.prh_set_gemm_mode(mode)
    t = timed(enc_all)
    with torch.no_grad():
        mem = m.encode_context(ctx[:64])
    if ref is None:
        ref = mem
    err = float((mem - ref).abs().max())
    print(f"  encode_context [{what:48s}] {t * 1e3:8.2f} ms  {B / t:9.0f} seg/s  {flop_pt * B * N / t / 1e12:7.1f} TFLOP/s  "
          f"max|memory - per-layer| {err:.2e}")
for prec, mode, what in ((None, 3, "per-layer encoder, decoder split-fp16"), ("fp32", 3, "fused encoder (fp32-accurate), decoder split-fp16"),
                         ("fp16", 4, "fused encoder fp16, decoder GEMMs bf16 (config 5)")):
    m.context_encoder.inference_precision = prec
    lib.prh_set_gemm_mode(mode)
    t = timed(full_all)
    print(f"  full eval forward [{what:48s}] {t * 1e3:8.2f} ms  {B / t:9.0f} seg/s")
lib.prh_set_gemm_mode(3)
print("HBM bytes per point by design: fused kernel 16 B in + 1,024 B out (memory); per-layer eval path additionally writes "
      "and re-reads z_cat 7,936 B + z_fus 4,096 B + fused 4,096 B")
