"""Worker of tests/test_dist_gpu.py: one rank of a 2-rank group that pushes the PRODUCT
LineRefineNet (HIP path) through TrainStep.  Both ranks share the box's single GPU, so the
group runs over gloo (CUDA tensors are staged through the host); the exchange code is the same
TrainStep uses over RCCL.  Writes its findings as JSON to the path in argv[1]."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import procedural as P
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.train_step import TrainStep
    dev = torch.device("cuda", 0)

    def make():
        m = LineRefineNet()
        m.load_state_dict(P.linerefine_state_dict(0), strict=True)
        m = m.to(dev).train()
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
        return m

    model, ref = make(), make()
    if rank == 1:                       # must be overwritten by the rank-0 buffer broadcast
        with torch.no_grad():
            model.context_encoder.bn3.running_mean.add_(5.0)
    ctx, noisy, target = [t.to(dev) for t in P.synth_batch(6, 160, 4, 32, seed=50 + rank)]
    step = TrainStep(model, None, decoder_chunk=4, world_size=world)       # fused Adam, HIP loss, chunked decoder
    loss = step(ctx, noisy, target)
    # expected exchange: mean over ranks of the per-rank gradients of the common pre-step weights
    rstep = TrainStep(ref, torch.optim.SGD(ref.parameters(), lr=0.0), decoder_chunk=None, world_size=None)
    rstep.grads.zero()
    rstep.forward_backward(ctx, noisy, target)
    want = rstep.grads.flat.clone()
    dist.all_reduce(want)
    want /= world
    got = step.grads.flat
    gerr = float((want - got).norm() / want.norm())
    weights = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(weights) for _ in range(world)]
    dist.all_gather(gathered, weights)
    bn = model.context_encoder.bn3.running_mean.detach().clone()
    bns = [torch.zeros_like(bn) for _ in range(world)]
    dist.all_gather(bns, bn)
    if rank == 0:
        json.dump({"grad_rel_l2": gerr, "weights_equal": bool(torch.equal(gathered[0], gathered[1])),
                   "bn_gap": float((bns[0] - bns[1]).abs().max()), "loss": float(loss),
                   "n_params": int(weights.numel())}, open(sys.argv[1], "w"))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
