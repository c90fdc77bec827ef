"""Which stock torch kernels are left in one training step, and where they come from.
usage: python scripts/diag_torch_ops.py [B] [N] [chunk]   (GPU box)
Prints the aten ops that launch kernels (add / copy / fill / sum ...) grouped by input shapes, with counts and
device time, for ONE TrainStep call after warm-up."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    chunk = int(sys.argv[3]) if len(sys.argv) > 3 else B
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = LineRefineNet().to(dev).train()
    st = TrainStep(m, None, decoder_chunk=chunk)
    batch = synthetic_batch(B, N, dev, seed=1)
    for _ in range(2):
        st(*batch)
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
        st(*batch)
        torch.cuda.synchronize()
    rows = {}
    for e in prof.events():
        if not e.name.startswith("aten::"):
            continue
        dt = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
        if dt <= 0:
            continue
        key = (e.name, str(e.input_shapes)[:110])
        c = rows.setdefault(key, [0, 0.0])
        c[0] += 1
        c[1] += dt
    tot = sum(v[1] for v in rows.values())
    print(f"B={B} N={N} chunk={chunk}: stock aten ops with device time: {sum(v[0] for v in rows.values())} calls, {tot / 1e3:.3f} ms")
    for (name, shp), (n, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{t / 1e3:9.3f} ms  {n:4d} x  {name:28s} {shp}")


if __name__ == "__main__":
    main()
