#!/usr/bin/env python
"""The reference's training loop (train_dist.py:143-242) on the MI355X path with synthetic
data: the ONLY model-side change against the reference is the import line.

    python examples/train_synthetic.py                       # one GPU
    torchrun --nproc_per_node=8 --master-addr 127.0.0.1 examples/train_synthetic.py

DistributedDataParallel over backend "nccl" (= RCCL on ROCm) is used exactly as the reference
does (find_unused_parameters=True, per-rank BatchNorm statistics, rank-0 logging/checkpoint)."""
import os
import sys

import torch
import torch.distributed as dist
import torch.optim as optim
from torch.nn.parallel import DistributedDataParallel as DDP

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pointnet_refine_amd.model import LineRefineNet          # was: from src.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch


def main():
    distributed = "LOCAL_RANK" in os.environ
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    rank = int(os.environ.get("RANK", "0"))
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl")
    torch.cuda.set_device(local_rank)
    device = torch.device(f"cuda:{local_rank}")

    BATCH_SIZE_PER_GPU, EPOCHS, STEPS_PER_EPOCH, LR = 32, 2, 10, 1e-3      # train_dist.py:118-121
    model = LineRefineNet().to(device)
    if distributed:
        model = DDP(model, device_ids=[local_rank], find_unused_parameters=True)   # train_dist.py:147
    optimizer = optim.Adam(model.parameters(), lr=LR)
    criterion = torch.nn.L1Loss()

    for epoch in range(EPOCHS):
        model.train()
        total = 0.0
        for it in range(STEPS_PER_EPOCH):
            context, noisy_line, target_offset = synthetic_batch(
                BATCH_SIZE_PER_GPU, 2048, device, seed=1000 * epoch + 10 * it + rank)
            optimizer.zero_grad()
            pred_offsets_stack = model(context, noisy_line)          # (6, B, 32, 3)
            loss = sum(criterion(pred_offsets_stack[i], target_offset)
                       for i in range(pred_offsets_stack.shape[0])) / pred_offsets_stack.shape[0]
            loss.backward()
            optimizer.step()
            total += loss.item()
        if rank == 0:
            print(f"Epoch [{epoch + 1}/{EPOCHS}] avg loss {total / STEPS_PER_EPOCH:.4f}")
    if rank == 0:
        sd = (model.module if distributed else model).state_dict()
        print(f"state_dict entries: {len(sd)} (reference: 205)")
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
