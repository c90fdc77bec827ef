#!/usr/bin/env python
"""Row f2 measurement: whole-scene context building (config 5 shape: 100k-point cloud, 4096
polylines, N=1024, crop radius 0.3 m) on the HIP path, with the CPU oracle (the reference's
algorithm) timed beside it on a bounded sample of the same lines.
usage: python tests/diag_context_bench.py [n_points] [n_lines]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import context_oracle as O
from pointnet_refine_amd.context import build_contexts_resampled, resample_polyline

P = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N, R, DECAY = 1024, 0.3, 2.0
rng = np.random.default_rng(0)
lines = []
for i in range(L):
    x = np.sort(rng.uniform(-60, 60, 6))
    lines.append(np.stack([x, rng.uniform(-40, 40) + 0.2 * np.sin(x / 9), rng.normal(0, 0.02, 6)], 1))
xyz = np.stack([rng.uniform(-60, 60, P), rng.uniform(-40, 40, P), rng.normal(0, 0.05, P)], 1)
cloud = np.column_stack([xyz, np.clip(rng.exponential(12, P), 0, 255)]).astype(np.float32)
t0 = time.perf_counter()
dense = np.stack([resample_polyline(l, 200) for l in lines]).astype(np.float32)
line = np.stack([resample_polyline(l, 32) for l in lines]).astype(np.float32)
t_host = time.perf_counter() - t0
dev = torch.device("cuda", 0)
ct, dt_, lt = (torch.from_numpy(a).to(dev) for a in (cloud, dense, line))
build_contexts_resampled(ct, dt_, lt, N, R, DECAY, seed=0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
iters = 5
e0.record()
for s in range(iters):
    ctx, counts = build_contexts_resampled(ct, dt_, lt, N, R, DECAY, seed=s)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
evals = 2.0 * L * P * 200          # count pass + fill pass
print(f"GPU: {L} lines x {P} points, N={N}: {ms:.2f} ms = {L / ms * 1e3:.0f} lines/s; "
      f"{evals / ms / 1e6:.1f} G brute-force-equivalent point-to-sample distance evaluations/s "
      f"(~{evals * 8 / ms / 1e9:.1f} TFLOP/s fp32 VALU of 157.3; bounding boxes skip most of them); host resampling {t_host * 1e3:.0f} ms; "
      f"mean points in tube {float(counts.float().mean()):.0f}")
# whole-scene refinement (config 5 shape): contexts + batched eval forward, random-init weights
from pointnet_refine_amd.io import refine_scene
from pointnet_refine_amd.model import LineRefineNet
torch.manual_seed(0)
model = LineRefineNet().to(dev).eval()
ref = None
for prec, what in (("layers", "per-layer eval kernels (round-1 path), fp32-accurate"),
                   ("fp32", "fused encoder kernel, fp32-accurate (2 fp16 planes)"),
                   ("fp16", "fused encoder kernel fp16 + decoder GEMMs bf16 (config 5)")):
    BL = int(os.environ.get("BATCH_LINES", "2048"))
    refine_scene(model, ct, lines[:BL], batch_lines=BL, precision=prec)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        refined, _ = refine_scene(model, ct, lines, batch_lines=BL, seed=1, precision=prec)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    if ref is None:
        ref = refined
    print(f"refine_scene [{what}]: {L} lines end to end (device resampling + contexts + eval forward in batches "
          f"of {BL} + copy back) {best * 1e3:.0f} ms = {L / best:.0f} lines/s; max |refined - per-layer| "
          f"{float(np.abs(refined - ref).max()):.2e}")
nb = 16
t0 = time.perf_counter()
for i in range(nb):
    O.build_context(cloud, dense[i].astype(np.float64), line[i].astype(np.float64), R, DECAY, N)
t_cpu = (time.perf_counter() - t0) / nb
print(f"CPU oracle (numpy, 1 thread, brute-force distances): {t_cpu * 1e3:.1f} ms/line = {1 / t_cpu:.1f} lines/s "
      f"(sample: first {nb} lines)")
try:
    from scipy.spatial import KDTree
    t0 = time.perf_counter()
    for i in range(nb):
        d, _ = KDTree(dense[i]).query(cloud[:, :3])
    t_kd = (time.perf_counter() - t0) / nb
    print(f"CPU crop alone with scipy KDTree as in the reference: {t_kd * 1e3:.1f} ms/line = {1 / t_kd:.1f} lines/s")
except Exception as e:      # scipy missing on the box
    print("scipy KDTree not available:", e)
