"""One data-parallel training iteration of LineRefineNet, MI355X-first.

What ``train_dist.py:173-189`` does per iteration (zero_grad, forward, deep-supervision L1,
backward with DDP gradient averaging, Adam step), restructured for one process per GPU
with 288 GB of HBM and point-to-point xGMI:

* **flat gradient bucket**: every ``param.grad`` is a view into ONE contiguous fp32 buffer
  (38.8 MB for the full model), so the data-parallel exchange is a single RCCL all-reduce
  per step instead of DDP's 25 MB buckets - the payload is tiny against a step of hundreds
  of ms, so fewer/larger collectives beat overlap.
* **per-rank BatchNorm statistics, rank-0 buffers broadcast before each forward** - the
  reference's DDP(broadcast_buffers=True) semantics without SyncBN (train_dist.py:143-147).
* **decoder micro-batching**: the encoder (whose BatchNorm needs the whole per-rank batch)
  runs once at full batch through the HIP path; the BatchNorm-free decoder is run over
  batch chunks, each chunk's graph freed after its backward.  Mathematically identical to
  the monolithic step (the loss is a mean, gradients add), it bounds the stock-PyTorch
  decoder's saved activations so B=4096 x N=1024 fits next to the encoder's 100+ GB.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from . import ops


class FlatGrads:
    """All parameter gradients as views into one buffer; one collective per step.  Every tensor starts
    on a 16-byte boundary (offsets padded to multiples of 4 elements; the padding stays zero): the
    parameters that FlatAdam re-homes with this same layout keep the alignment the kernels' 16-byte
    loads and LDS-DMA need (the 3-wide regression biases would otherwise shift everything behind them)."""

    ALIGN = 4

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        self.offsets = []
        off = 0
        for p in self.params:
            off = -(-off // self.ALIGN) * self.ALIGN
            self.offsets.append(off)
            off += p.numel()
        n = -(-off // self.ALIGN) * self.ALIGN
        p0 = self.params[0]
        self.flat = torch.zeros(n, dtype=p0.dtype, device=p0.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        self.rebind()

    def rebind(self):
        for p, v in zip(self.params, self.views):
            if p.grad is not v:        # optimizer.zero_grad(set_to_none=True) may have dropped it
                p.grad = v

    def zero(self):
        self.flat.zero_()
        self.rebind()

    def offset_of(self, param) -> int:
        """Element offset of `param`'s gradient inside the flat buffer."""
        for p, off in zip(self.params, self.offsets):
            if p is param:
                return off
        raise KeyError("parameter is not in this gradient buffer")

    def all_reduce_mean(self, world_size: int, group=None, start: int = 0, stop=None, async_op: bool = False):
        """Mean over the group of flat[start:stop] (default: everything).  Runs whenever a process group
        exists (also at world_size 1 under torchrun, so the single-GPU launch exercises the same RCCL
        calls as the 8-GPU one).  async_op: returns a handle whose wait() also applies the 1/n."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        n = dist.get_world_size(group)
        if world_size not in (None, 1, n):
            raise RuntimeError(f"TrainStep: world_size={world_size} disagrees with the process group ({n} ranks)")
        part = self.flat[start:stop]
        if part.numel() == 0:
            return None
        work = dist.all_reduce(part, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if not async_op:
            if n > 1:                  # the mean DDP takes (train_dist.py:147), whatever the caller passed
                part.div_(n)
            return None

        class _Pending:
            def wait(self_inner):
                work.wait()
                if n > 1:
                    part.div_(n)
        return _Pending()


class FlatBuffers:
    """BatchNorm running statistics for the rank-0 broadcast DDP performs (broadcast_buffers=True,
    train_dist.py:143-147).  The floating buffers are RE-HOMED as views into one flat fp32 buffer and
    the `num_batches_tracked` counters into one flat int64 buffer (the kernels write them through raw
    pointers, so nothing else changes): the per-step exchange is two collectives on tensors that already
    exist - no concatenation, no scatter back."""

    def __init__(self, module: torch.nn.Module):
        self.bufs = [b for _, b in module.named_buffers() if b.is_floating_point()]
        self.ints = [b for _, b in module.named_buffers() if not b.is_floating_point()]
        self.flat = self._rehome(self.bufs, torch.float32)
        self.flat_int = self._rehome(self.ints, torch.int64)

    @staticmethod
    def _rehome(tensors, dtype):
        tensors = [t for t in tensors if t.dtype == dtype]
        if not tensors:
            return None
        n = sum(t.numel() for t in tensors)
        flat = torch.empty(n, dtype=dtype, device=tensors[0].device)
        off = 0
        with torch.no_grad():
            for t in tensors:
                k = t.numel()
                flat[off:off + k].copy_(t.reshape(-1))
                t.data = flat[off:off + k].view(t.shape)
                off += k
        return flat

    def broadcast(self, group=None):
        if self.flat is not None:
            dist.broadcast(self.flat, 0, group=group)
        if self.flat_int is not None:
            dist.broadcast(self.flat_int, 0, group=group)
        for b in self.bufs + self.ints:          # anything of another dtype stayed where it was
            if b.dtype not in (torch.float32, torch.int64):
                dist.broadcast(b, 0, group=group)


def deep_supervision_l1(out, target, denom=None, geometry=None, points=None):
    """sum_l sum |pred_l - target| / denom, denom = out.numel() by default: (1/L) sum_l
    nn.L1Loss(pred_l, target) of train.py:63-68 / train_dist.py:180-186, as one HIP pass that
    also produces the gradient and, into `geometry`, the initial / refined point-to-point errors
    the reference logs (train_dist.py:190-203) (row f3).  GPU tensors only."""
    return ops.deep_supervision_l1(out, target, denom, geometry, points)


class FlatAdam:
    """torch.optim.Adam semantics (lr 1e-3, betas (0.9, 0.999), eps 1e-8, no amsgrad: the
    reference's optimiser, train.py:40 / train_dist.py:150) as ONE HIP launch per step over flat
    buffers (row f3): the parameters become views into one fp32 buffer laid out like the
    FlatGrads gradient buffer, with flat first/second-moment buffers next to it.  Duck-types the
    optimizer calls the reference's loops make (step, zero_grad, param_groups[0]['lr'])."""

    def __init__(self, grads: "FlatGrads", lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.grads = grads
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                              "params": grads.params}]
        flat = torch.zeros_like(grads.flat)
        with torch.no_grad():
            for p, off in zip(grads.params, grads.offsets):
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view_as(p)          # parameters now live in the flat buffer
        self.flat = flat
        self.exp_avg = torch.zeros_like(flat)
        self.exp_avg_sq = torch.zeros_like(flat)
        self.steps = 0

    def zero_grad(self, set_to_none: bool = False):
        self.grads.zero()

    @torch.no_grad()
    def step(self):
        import ctypes as C
        from . import _lib as L
        self.steps += 1
        from . import ops
        ops.invalidate_fused_images()          # this kernel writes the weights behind autograd's version counters
        g = self.param_groups[0]
        dev = self.flat.device
        p = lambda t: C.c_void_p(t.data_ptr())
        L.check(L.lib().prh_adam_step(p(self.flat), p(self.grads.flat), p(self.exp_avg), p(self.exp_avg_sq),
                                      self.flat.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                      float(g["eps"]), float(g["weight_decay"]), self.steps, dev.index,
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "prh_adam_step")


class TrainStep:
    def __init__(self, model, optimizer=None, decoder_chunk: Optional[int] = None, world_size: int = 1,
                 group=None, broadcast_buffers: bool = True, lr: float = 1e-3, loss_fn=None,
                 graph: bool = False):
        """optimizer = None: the fused flat-buffer Adam (FlatAdam, the reference's Adam(lr=1e-3));
        any torch optimizer over model.parameters() also works.  loss_fn(out, target, denom):
        defaults to the HIP deep-supervision L1 (the reference's loss).
        graph = True: forward + loss + backward are captured once into a HIP graph (fixed batch
        shape) and replayed - for the launch-bound small-batch regime (the reference trains at 32
        segments per GPU: ~2,000 launches of a few microseconds each).  Dropout stays random per
        step: the library mixes a device counter, advanced inside the graph, into every seed."""
        self.model = model
        self.loss_fn = loss_fn if loss_fn is not None else deep_supervision_l1
        self.geometry = None
        self.keep_out = False         # True: forward_backward stashes the detached predictions in last_out
        self.last_out = None
        self.use_graph, self._graph, self._static, self._seed = bool(graph), None, None, None
        self.chunk = decoder_chunk
        self.world, self.group = world_size, group
        self.grads = FlatGrads(model.parameters())
        self._sinks_on = False
        self._ensure_sinks()
        self.opt = optimizer if optimizer is not None else FlatAdam(self.grads, lr=lr)
        distributed = dist.is_available() and dist.is_initialized()
        self.bufs = FlatBuffers(model) if (distributed and broadcast_buffers) else None

    def _loss(self, out, target, denom, points):
        if self.loss_fn is deep_supervision_l1:        # the HIP loss also fills the step metrics
            return deep_supervision_l1(out, target, denom, self.geometry, points)
        return self.loss_fn(out, target, denom)

    def forward_backward(self, context, noisy_line, target, accumulate: bool = False, decoder_done=None):
        """One (micro-batched) forward + loss + backward into the flat gradient buffer.  Inside, the
        library's backward kernels write parameter gradients straight into the buffer (ops gradient
        sinks) instead of going through autograd's accumulation: with accumulate=False (default) the
        first contribution to a tensor OVERWRITES what the buffer held (the buffer should be zero on
        entry: grads.zero(), as __call__ does); accumulate=True ADDS this call's gradients to the
        buffer's content - gradient accumulation over several calls before one optimiser step.
        decoder_done: called (no arguments) once the gradients of every decoder-side parameter are
        final and only the encoder-side backward is left (see decoder_grad_offset)."""
        from . import ops
        self._ensure_sinks()
        with ops.sinks_active(new_step=True, owner=self, preset_written=accumulate):
            return self._forward_backward(context, noisy_line, target, decoder_done)

    def decoder_grad_offset(self) -> int:
        """flat[offset:] holds the gradients of pos_emb, decoder_layers and reg_branches - final as soon as
        the decoder micro-batches are done (parameter order of LineRefineNet: context_encoder,
        context_proj, point_mlp, then these).  len(flat) when the model has no such split."""
        m = self.model
        first = getattr(getattr(m, "pos_emb", None), "mlp", None)
        if first is None or not hasattr(m, "decode"):
            return self.grads.flat.numel()
        names = [n for n, p in m.named_parameters() if p.requires_grad]
        i0 = next((i for i, n in enumerate(names) if n.startswith("pos_emb.")), None)
        if i0 is None or any(n.split(".")[0] in ("context_encoder", "context_proj", "point_mlp") for n in names[i0:]):
            return self.grads.flat.numel()
        return self.grads.offsets[i0]

    def _ensure_sinks(self):
        from . import ops
        if not getattr(self, "_sinks_on", False):
            ops.register_grad_sinks(zip(self.grads.params, self.grads.views), owner=self)
            self._sinks_on = True

    def _forward_backward(self, context, noisy_line, target, decoder_done=None):
        m = self.model
        if self.loss_fn is deep_supervision_l1:
            if self.geometry is None:
                self.geometry = torch.zeros(2, dtype=torch.float32, device=context.device)
            self.geometry.zero_()                   # [init_err, refine_err] of this step (device; no sync)
        B = context.shape[0]
        if self.chunk is None or self.chunk >= B or not hasattr(m, "decode"):
            out = m(context, noisy_line)
            loss = self._loss(out, target, float(out.numel()), float(target.numel() // 3))
            loss.backward()
            if decoder_done is not None:
                decoder_done()
            if self.keep_out:
                self.last_out = out.detach()
            return loss.detach()
        # encoder side at full batch (BatchNorm statistics span the whole per-rank batch)
        memory = m.encode_context(context)
        tgt0 = m.encode_line(noisy_line)
        d_memory = torch.empty_like(memory)
        d_tgt0 = torch.empty_like(tgt0)
        mem_d, tgt_d = memory.detach(), tgt0.detach()
        total = torch.zeros((), device=context.device, dtype=torch.float32)
        denom = float(6 * B * noisy_line.shape[1] * noisy_line.shape[2])
        outs = [] if self.keep_out else None
        for s in range(0, B, self.chunk):
            e = min(s + self.chunk, B)
            mem_c = mem_d[s:e].requires_grad_()
            tgt_c = tgt_d[s:e].requires_grad_()
            out = m.decode(context[s:e], noisy_line[s:e], mem_c, tgt_c)
            loss_c = self._loss(out, target[s:e], denom, float(target.numel() // 3))
            loss_c.backward()
            d_memory[s:e] = mem_c.grad
            d_tgt0[s:e] = tgt_c.grad
            total += loss_c.detach()
            if outs is not None:
                outs.append(out.detach())
            del out, loss_c, mem_c, tgt_c
        if decoder_done is not None:        # pos_emb / decoder_layers / reg_branches gradients are final
            decoder_done()
        torch.autograd.backward([memory, tgt0], [d_memory, d_tgt0])
        if outs is not None:
            self.last_out = torch.cat(outs, dim=1)
        return total

    def close(self):
        """Drop the captured graph (its private memory pool goes back to the allocator) and
        unregister the device seed counter (the library keeps the pointer it was given; call
        this - or drop the TrainStep - before freeing GPU memory).  The step falls back to eager
        launches; a later call with graph=True captures again."""
        if getattr(self, "_seed", None) is not None:
            try:
                from . import _lib as L
                L.lib().prh_set_dropout_seed_source(None)
            except Exception:
                pass
            self._seed = None
        self._graph, self._static, self._static_loss = None, None, None
        if getattr(self, "grads", None) is not None and getattr(self, "_sinks_on", False):
            try:
                from . import ops
                ops.clear_grad_sinks(self.grads.params, owner=self)      # only the entries this step registered
            except Exception:
                pass
            self._sinks_on = False

    def __del__(self):
        self.close()

    def _capture(self, context, noisy_line, target):
        """Warm up on a side stream, then capture zero-grad + forward + loss + backward."""
        import ctypes as C
        from . import _lib as L
        dev = context.device
        self._seed = torch.zeros(1, dtype=torch.int32, device=dev)
        L.check(L.lib().prh_set_dropout_seed_source(C.c_void_p(self._seed.data_ptr())), "prh_set_dropout_seed_source")
        self._static = [t.clone() for t in (context, noisy_line, target)]
        if self.loss_fn is deep_supervision_l1 and self.geometry is None:
            self.geometry = torch.zeros(2, dtype=torch.float32, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # the warm-up passes are train-mode forwards: they would push this batch into every
        # BatchNorm running statistic (and num_batches_tracked) twice more than an eager step
        # does - snapshot the buffers and put them back, so graph and eager runs checkpoint alike
        bufs = [b for b in self.model.buffers()]
        saved = [b.detach().clone() for b in bufs]
        with torch.cuda.stream(side):
            for _ in range(2):                      # allocations, workspaces, lazy initialisation
                self._seed.add_(1)
                self.grads.flat.zero_()
                self.forward_backward(*self._static)
            with torch.no_grad():
                for b, s0 in zip(bufs, saved):
                    b.copy_(s0)
        torch.cuda.current_stream(dev).wait_stream(side)
        del saved
        # the capture allocates from a private pool, which cannot reuse what the warm-up left in
        # the default pool - its cached blocks and the per-stream scratch buffers of ops._ws
        # (tens of GB at B=4096): hand them back first, or the two pools together exceed 288 GB
        torch.cuda.synchronize(dev)
        from . import ops as _ops
        _ops.release_workspaces()
        torch.cuda.empty_cache()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._seed.add_(1)
            self.grads.flat.zero_()
            self._static_loss = self.forward_backward(*self._static)

    def __call__(self, context, noisy_line, target):
        if self.bufs is not None:
            with torch.no_grad():
                self.bufs.broadcast(self.group)
        if self.use_graph:
            if self._graph is None:
                self.grads.rebind()
                self._capture(context, noisy_line, target)
            for dst, src in zip(self._static, (context, noisy_line, target)):
                if dst.shape != src.shape:
                    raise RuntimeError("TrainStep(graph=True): the batch shape is fixed at capture time")
                dst.copy_(src)
            self._graph.replay()
            loss = self._static_loss
            self.grads.all_reduce_mean(self.world, self.group)
        else:
            self.grads.zero()
            # the gradient exchange of the decoder side (26 MB of the 38.8 MB) is issued as soon as the
            # decoder micro-batches are done and runs under the encoder's backward - what DDP's bucket
            # hooks do for the reference (train_dist.py:147,188); the encoder side follows at the end
            pending = []
            split = self.decoder_grad_offset() if (dist.is_available() and dist.is_initialized()) else None

            def decoder_done():
                h = self.grads.all_reduce_mean(self.world, self.group, start=split, async_op=True)
                if h is not None:
                    pending.append(h)
            overlap = split is not None and 0 < split < self.grads.flat.numel()
            loss = self.forward_backward(context, noisy_line, target, decoder_done=decoder_done if overlap else None)
            self.grads.all_reduce_mean(self.world, self.group, stop=split if overlap else None)
            for h in pending:
                h.wait()
        self.opt.step()
        return loss
