"""N>1 path on CPU: 2 gloo ranks exercise TrainStep's data-parallel exchange (flat-gradient
all-reduce = mean of per-rank gradients, rank-0 BatchNorm buffer broadcast, identical
weights after the step) with a small CPU stand-in module - the HIP model itself cannot run
on CPU by design (no fallback)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Tiny(torch.nn.Module):
    """Same structure as the hot path in miniature: shared MLP + BatchNorm, per-rank stats."""

    def __init__(self):
        super().__init__()
        self.l1 = torch.nn.Linear(4, 16)
        self.bn = torch.nn.BatchNorm1d(16)
        self.l2 = torch.nn.Linear(16, 3)

    def forward(self, context, noisy_line):
        h = torch.relu(self.bn(self.l1(context.reshape(-1, 4))))
        g = h.reshape(context.shape[0], -1, 16).max(dim=1)[0]
        off = self.l2(g).unsqueeze(1) + 0 * noisy_line
        return torch.stack([off + noisy_line * 0.1 * l for l in range(6)])


def _torch_l1(out, target, denom=None):
    """The reference's loss in torch ops for the CPU stand-in models of this file (the product's
    default loss is the HIP kernel, which - like the model - has no CPU path)."""
    return (out - target.unsqueeze(0)).abs().sum() / (out.numel() if denom is None else denom)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pointnet_refine_amd.train_step import TrainStep
    torch.manual_seed(0)
    model = Tiny().train()
    if rank == 1:                       # rank 1 starts with different BN buffers: must be overwritten
        model.bn.running_mean.add_(5.0)
    ref = Tiny().train()
    ref.load_state_dict({k: v.clone() for k, v in model.state_dict().items()})
    g = torch.Generator().manual_seed(100 + rank)
    ctx = torch.randn(6, 20, 4, generator=g)
    line = torch.randn(6, 32, 3, generator=g)
    tgt = torch.randn(6, 32, 3, generator=g)
    opt = torch.optim.SGD(model.parameters(), lr=0.1)
    step = TrainStep(model, opt, decoder_chunk=None, world_size=world, loss_fn=_torch_l1)
    loss = step(ctx, line, tgt)
    # expected: mean over ranks of the per-rank gradients of the pre-step weights
    if rank == 1:
        ref.bn.running_mean.sub_(5.0)
    _torch_l1(ref(ctx, line), tgt).backward()
    flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()])
    dist.all_reduce(flat)
    flat /= world
    got = torch.cat([v.reshape(-1) for v in step.grads.views])      # the flat buffer pads tensors to 16-byte boundaries
    weights = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(weights) for _ in range(world)]
    dist.all_gather(gathered, weights)
    bufs = torch.cat([model.bn.running_mean, model.bn.running_var])
    gb = [torch.zeros_like(bufs) for _ in range(world)]
    dist.all_gather(gb, bufs)
    if rank == 0:
        out.put({"grad_err": float((flat - got).abs().max()),
                 "weights_equal": bool(torch.equal(gathered[0], gathered[1])),
                 "buf_rank1_minus_rank0_mean": float((gb[1] - gb[0]).abs().max()),
                 "loss": float(loss)})
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_buffer_broadcast():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res["grad_err"] < 1e-6
    assert res["weights_equal"]
    # per-rank BN statistics (no SyncBN): after the step the running stats differ between
    # ranks by their local batch statistics only, not by rank 1's +5 offset (it was
    # overwritten by the rank-0 broadcast before the forward)
    assert res["buf_rank1_minus_rank0_mean"] < 1.0


def test_chunked_decoder_matches_monolithic_step():
    """TrainStep's decoder micro-batching is exact: same loss and gradients as one pass."""
    sys.path.insert(0, ROOT)
    from pointnet_refine_amd.train_step import TrainStep

    class Split(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.enc = torch.nn.Linear(4, 8)
            self.bn = torch.nn.BatchNorm1d(8)
            self.line = torch.nn.Linear(3, 8)
            self.dec = torch.nn.Linear(8, 3)

        def encode_context(self, c):
            B, N, _ = c.shape
            return torch.relu(self.bn(self.enc(c.reshape(B * N, 4)))).reshape(B, N, 8)

        def encode_line(self, l):
            return self.line(l)

        def decode(self, c, l, memory, tgt):
            h = tgt + memory.mean(dim=1, keepdim=True)
            return torch.stack([self.dec(h) * (i + 1) for i in range(6)])

        def forward(self, c, l):
            return self.decode(c, l, self.encode_context(c), self.encode_line(l))

    torch.manual_seed(1)
    a, b = Split().train(), Split().train()
    b.load_state_dict(a.state_dict())
    ctx, line, tgt = torch.randn(8, 10, 4), torch.randn(8, 32, 3), torch.randn(8, 32, 3)
    sa = TrainStep(a, torch.optim.SGD(a.parameters(), lr=0.0), decoder_chunk=None, loss_fn=_torch_l1)
    sb = TrainStep(b, torch.optim.SGD(b.parameters(), lr=0.0), decoder_chunk=3, loss_fn=_torch_l1)
    la, lb = sa(ctx, line, tgt), sb(ctx, line, tgt)
    assert abs(float(la) - float(lb)) < 1e-6
    assert float((sa.grads.flat - sb.grads.flat).abs().max()) < 1e-6
