"""Per-tensor gradient error of a GEMM mode against the exact-fp32 cores at a given size, at
initialisation and after some optimiser steps (conv1.weight in bf16 mode).
python scripts/bf16_grad_budget.py [B] [N] [steps] [mode]     (on the GPU box)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pointnet_refine_amd import _lib
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch
from pointnet_refine_amd.train_step import TrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 25
MODE = bench.GEMM_MODES[sys.argv[4]] if len(sys.argv) > 4 else 4
dev = torch.device("cuda", 0)
lib = _lib.lib()
lib.prh_set_gemm_mode(MODE)
torch.manual_seed(0)
model = LineRefineNet().to(dev).train()
step = TrainStep(model, None, decoder_chunk=2048)
batch = synthetic_batch(B, N, dev, seed=1234)


def report(tag, batch_a=None, mode=MODE):
    r = bench.parity_check(step, model, batch, lib, mode, dev, batch_bench=batch_a)
    per = sorted(r["per_tensor"], key=lambda t: -t[2] / max(t[1], 1e-30))
    gmax = max(t[1] for t in per)
    print(f"== {tag}: out max-abs {r['out_max_abs_vs_exact_fp32']:.3e} rel-L2 {r['out_rel_l2_vs_exact_fp32']:.3e} "
          f"all grads {r['all_grads_rel_l2']:.3e} worst {r['worst_grad']} {r['worst_grad_rel_l2']:.3e}")
    shown = 0
    for n, bn, dn, *_ in per:
        if bn > 1e-7 * gmax and shown < 10:
            print(f"   {n:48s} |g| {bn:.3e}  rel-L2 {dn / bn:.3e}")
            shown += 1
    for n, bn, dn, *_ in per:
        if n in ("context_encoder.conv1.weight", "point_mlp.0.weight", "context_encoder.conv2.weight"):
            print(f"   [{n}] |g| {bn:.3e} rel-L2 {dn / max(bn, 1e-30):.3e}")
    sys.stdout.flush()


report("initialisation")
qb = (batch[0].bfloat16().float(), batch[1], batch[2])
if MODE != 0:
    report("exact cores, context rounded to bf16 vs exact context", batch_a=qb, mode=0)
    lib.prh_set_gemm_mode(MODE)
for i in range(STEPS):
    step(*batch)
torch.cuda.synchronize()
report(f"after {STEPS} Adam steps")
