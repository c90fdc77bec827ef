"""The reference wraps the model in DistributedDataParallel(find_unused_parameters=True) over
backend "nccl" (train_dist.py:147).  One-rank RCCL group on the GPU box: the wrapped HIP model
trains, and its gradients equal the unwrapped model's."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from conftest import rel_l2
from oracle import procedural as P

pytestmark = pytest.mark.gpu


def test_ddp_wrapper_matches_plain_module():
    from torch.nn.parallel import DistributedDataParallel as DDP
    from pointnet_refine_amd.model import LineRefineNet
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    try:
        sd = P.linerefine_state_dict(0)
        ctx, noisy, target = P.synth_batch(4, 128, 4, 32, seed=3)
        ctx, noisy, target = ctx.cuda(), noisy.cuda(), target.cuda()
        grads = []
        for wrap in (False, True):
            m = LineRefineNet()
            m.load_state_dict(sd, strict=True)
            m = m.cuda().train()
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.0
            net = DDP(m, device_ids=[0], find_unused_parameters=True) if wrap else m
            opt = torch.optim.Adam(net.parameters(), lr=1e-3)
            opt.zero_grad()
            out = net(ctx, noisy)
            loss = sum(torch.nn.functional.l1_loss(out[l], target) for l in range(6)) / 6
            loss.backward()
            grads.append({k: v.grad.clone() for k, v in m.named_parameters()})
            opt.step()
            assert len((net.module if wrap else net).state_dict()) == 205
        for k in grads[0]:
            if float(grads[0][k].abs().max()) > 1e-6:
                assert rel_l2(grads[0][k], grads[1][k]) < 1e-5, k
    finally:
        dist.destroy_process_group()
