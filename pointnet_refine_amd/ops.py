"""torch.autograd.Functions over the C ABI (``include/pointnet_refine_hip.h``).

PyTorch is plumbing here: it owns device memory, streams and the autograd graph; all
arithmetic of the shared-MLP path runs in the HIP library.  Every op raises on non-GPU
tensors - there is no CPU or eager fallback.
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from typing import List, Optional, Sequence

import torch

from . import _lib as L

_workspaces = {}


def _ws(device: torch.device, nbytes: int) -> torch.Tensor:
    """Grow-only scratch buffer per device.  Safe to reuse across calls because all work
    of one device is enqueued on its current stream in program order."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _workspaces.get(key)
    if t is None or t.numel() < nbytes:
        _workspaces.pop(key, None)
        t = None
        t = torch.empty(int(nbytes * 1.05) + 4096, dtype=torch.uint8, device=device)
        _workspaces[key] = t
    return t


def bf16_mode() -> bool:
    """True in GEMM mode 4 (bf16 operands and bf16 activation storage, BASELINE config 3)."""
    return L.lib().prh_get_gemm_mode() == 4


GEMM_MODES = {"fp32": 0, "split": 1, "bf16op": 2, "split16": 3, "bf16": 4}


def set_gemm_mode(mode) -> int:
    """Select the GEMM core family (`prh_set_gemm_mode`): a name of GEMM_MODES or its number.
    Returns the previous mode."""
    m = GEMM_MODES[mode] if isinstance(mode, str) else int(mode)
    old = L.lib().prh_get_gemm_mode()
    L.check(L.lib().prh_set_gemm_mode(m), "prh_set_gemm_mode")
    return old


_MODE_LOCK = threading.RLock()


class gemm_mode_scope:
    """with ops.gemm_mode_scope("bf16"): ... - the GEMM core family is ONE setting per process (the
    library reads it in the forward and again in the backward, which runs on autograd's device thread,
    so it cannot be thread-local); this scope takes a process-wide lock, switches the mode and puts the
    previous one back on exit, exception or not.  Two threads that want different modes serialise here
    instead of silently running each other's kernels."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        _MODE_LOCK.acquire()
        try:
            self.old = set_gemm_mode(self.mode)
        except Exception:
            _MODE_LOCK.release()
            raise
        return self

    def __exit__(self, *exc):
        try:
            set_gemm_mode(self.old)
        finally:
            _MODE_LOCK.release()
        return False


def release_workspaces():
    _workspaces.clear()


# ------------------------------------------------------------------------------------------
# Gradient sinks.  Autograd adds every parameter gradient a Function returns into the existing
# `.grad` with one elementwise kernel per parameter (~250 launches per step of this model).  A
# training loop that owns the gradient buffers (train_step.TrainStep: one flat fp32 buffer, zeroed
# at the start of a step) registers them here; while `sinks_active` the backward kernels then write
# a parameter's gradient straight into its buffer - the first contribution of a step overwrites,
# later ones (decoder micro-batches, shared weights) are added - and the Function returns None for
# it.  Views of a registered parameter (the q / k / v row blocks of an in_proj_weight) resolve to
# the matching slice.  Invariant: within a step, a region of the buffer is written either by these
# Functions or by autograd, never both before the Functions' first write.
# ------------------------------------------------------------------------------------------
_GRAD_SINKS = {}          # id(parameter) -> (parameter, gradient tensor, id(owner) or None): no strong reference to the owner
_SINK_WRITTEN = set()     # (data_ptr, numel) of the regions written in the current sinks_active scope
_SINKS_ACTIVE = False
_DIRECT = object()


def register_grad_sinks(params_and_grads, owner=None):
    """owner: whoever registers (a TrainStep); clear_grad_sinks(params, owner) later removes only the
    entries that are still that owner's - a newer registration for the same parameter survives."""
    for prm, g in params_and_grads:
        if g is not None and g.is_contiguous() and g.dtype == torch.float32:
            _GRAD_SINKS[id(prm)] = (prm, g, None if owner is None else id(owner))


def clear_grad_sinks(params=None, owner=None):
    if params is None:
        _GRAD_SINKS.clear()
        return
    for prm in params:
        ent = _GRAD_SINKS.get(id(prm))
        if ent is not None and (owner is None or ent[2] == id(owner)):
            del _GRAD_SINKS[id(prm)]


class sinks_active:
    """with ops.sinks_active(new_step=True): forward + backward of one (micro-)batch.  The set of regions
    already written belongs to the scope (saved and restored around it, so a scope opened and closed
    elsewhere - another TrainStep's __del__, a nested step - cannot make this one overwrite where it
    should add).  owner: only that owner's sinks are live inside the scope.  preset_written: treat every
    live sink as already written, i.e. ADD to the buffers' content (gradient accumulation)."""

    def __init__(self, new_step=True, owner=None, preset_written=False):
        self.new_step, self.owner, self.preset = new_step, owner, preset_written

    def __enter__(self):
        global _SINKS_ACTIVE, _SINK_WRITTEN, _SINK_OWNER
        self.prev = (_SINKS_ACTIVE, _SINK_WRITTEN, _SINK_OWNER)
        _SINK_OWNER = None if self.owner is None else id(self.owner)
        _SINKS_ACTIVE = any(_SINK_OWNER is None or e[2] == _SINK_OWNER for e in _GRAD_SINKS.values())
        if self.new_step:
            _SINK_WRITTEN = set()
            if self.preset:
                for _, g, own in _GRAD_SINKS.values():
                    if _SINK_OWNER is None or own == _SINK_OWNER:
                        _SINK_WRITTEN.add((g.data_ptr(), g.numel()))
                _SINK_WRITTEN.add("preset")
        return self

    def __exit__(self, *exc):
        global _SINKS_ACTIVE, _SINK_WRITTEN, _SINK_OWNER, LAST_SINK_WRITES
        LAST_SINK_WRITES = len(_SINK_WRITTEN - {"preset"})
        _SINKS_ACTIVE, _SINK_WRITTEN, _SINK_OWNER = self.prev
        return False


LAST_SINK_WRITES = 0      # regions written through sinks in the scope that closed last (tests)


_SINK_OWNER = None


def _written(key) -> bool:
    """True when the region was written in this scope (or the scope adds to everything: preset)."""
    return key in _SINK_WRITTEN or "preset" in _SINK_WRITTEN


class CatRowsFn(torch.autograd.Function):
    """torch.cat of row blocks of several parameters (the key / value rows of the six in_proj weights,
    model.py cross-attention projections) whose backward hands every block's gradient to its sink
    with one copy (first contribution of the step) or add, instead of autograd's zero-padded
    full-size gradient per slice plus an accumulation each."""

    @staticmethod
    def forward(ctx, *blocks):
        ctx.sinks = [_sink_view(b) for b in blocks]
        ctx.sizes = [b.shape[0] for b in blocks]
        return torch.cat(blocks)

    @staticmethod
    def backward(ctx, g):
        outs, o = [], 0
        for sink, n in zip(ctx.sinks, ctx.sizes):
            gi = g[o:o + n]
            o += n
            if sink is None or not _SINKS_ACTIVE:
                outs.append(gi)
                continue
            key = (sink.data_ptr(), sink.numel())
            if _written(key):
                sink.add_(gi)
            else:
                _SINK_WRITTEN.add(key)
                sink.copy_(gi)
            outs.append(None)
        return tuple(outs)


def cat_rows(blocks):
    return CatRowsFn.apply(*blocks)


def _sink_view(w):
    """Forward-time lookup: the gradient region of `w` (a registered parameter or a contiguous view
    of one), or None."""
    if not _SINKS_ACTIVE or w is None or not w.requires_grad:
        return None
    base = w._base if w._base is not None else w
    ent = _GRAD_SINKS.get(id(base))
    if ent is None or (_SINK_OWNER is not None and ent[2] != _SINK_OWNER):
        return None
    g = ent[1]
    if w is base:
        return g if g.shape == w.shape else None
    if not w.is_contiguous() or not base.is_contiguous():
        return None
    off = w.storage_offset() - base.storage_offset()
    if off < 0 or off + w.numel() > g.numel():
        return None
    return g.reshape(-1)[off:off + w.numel()].view(w.shape)


def _grad_buf(sink, shape, dev, need=True):
    """(fp32 tensor of `shape` the backward kernel writes, token for _grad_ret)."""
    if not need:
        return None, None
    shape = tuple(int(d) for d in shape)
    n = 1
    for d in shape:
        n *= d
    if sink is None or not _SINKS_ACTIVE or sink.numel() != n:
        return torch.empty(shape, dtype=torch.float32, device=dev), None
    sink = sink.view(shape)
    key = (sink.data_ptr(), sink.numel())
    if _written(key):
        return torch.empty(shape, dtype=torch.float32, device=dev), sink
    _SINK_WRITTEN.add(key)
    return sink, _DIRECT


def _grad_ret(buf, token):
    if token is None:
        return buf
    if token is not _DIRECT:
        token.add_(buf)
    return None


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _req_gpu_f32(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"pointnet_refine_amd: {name} must be a GPU tensor (got {t.device}); "
                           "the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"pointnet_refine_amd: {name} must be float32 (got {t.dtype})")


def _bn_layer(w, b, g, beta, rm, rv, nbt) -> L.BnLayer:
    cout, cin = w.shape[0], w.shape[1]
    return L.BnLayer(_p(w), _p(b), _p(g), _p(beta), _p(rm), _p(rv), _p(nbt), cin, cout)


# ------------------------------------------------------------------------------------------
# nn.Linear
# ------------------------------------------------------------------------------------------
def operand_absmax(t):
    """1-element device tensor max|t| of a contiguous 2-D fp32 tensor (prh_operand_absmax)."""
    out = torch.empty(1, dtype=torch.float32, device=t.device)
    ws = _ws(t.device, L.lib().prh_operand_absmax_workspace_bytes())
    L.check(L.lib().prh_operand_absmax(_p(t), t.shape[1], t.shape[0], t.shape[1], _p(out), _p(ws), ws.numel(),
                                       t.device.index, _stream(t.device)), "prh_operand_absmax")
    return out


class LinearFn(torch.autograd.Function):
    """y = x W^T + b on the fp32 MFMA GEMM core (context_proj, src/model.py:147,194)."""

    @staticmethod
    def forward(ctx, x, w, b, x_amax=None, relu=False, resid=None, dropout_p=0.0, seed=0):
        """x_amax: optional 1-element device tensor >= max|x| (saves the read pass that places the
        operand for the split-fp16 core; e.g. the encoder's bound for its output).  relu: apply
        ReLU in the GEMM epilogue (the backward masks dy with y > 0).  resid: tensor of the output's
        shape added in the epilogue, y = act(x W^T + b + resid).  dropout_p > 0 (with relu): the output
        leaves the epilogue dropped out, y = dropout(relu(.)), decided by a hash of (seed, row, column) -
        the FFN hidden layer (src/model.py:131); the backward reads the mask off y."""
        ctx.w_sink, ctx.b_sink = _sink_view(w), _sink_view(b)
        if dropout_p > 0.0 and not relu:
            raise RuntimeError("pointnet_refine_amd.linear: epilogue dropout comes with the fused ReLU (the backward reads the mask off y > 0)")
        ctx.drop_scale = 1.0 / (1.0 - dropout_p) if dropout_p > 0.0 else 1.0
        if x.dtype == torch.bfloat16:
            if dropout_p > 0.0:
                raise RuntimeError("pointnet_refine_amd.linear (bf16 input): no epilogue dropout")
            return LinearFn._forward_bf16(ctx, x, w, b, relu, resid)
        _req_gpu_f32(x, "input")
        _req_gpu_f32(w, "weight")
        n, k = w.shape
        if x.shape[-1] != k:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.shape} and {k}x{n})")
        if k % 4:
            raise RuntimeError("pointnet_refine_amd.linear: in_features must be a multiple of 4")
        ctx.bf16_in = False
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = w.contiguous()
        rows = x2.shape[0]
        y = torch.empty((rows, n), dtype=torch.float32, device=x.device)
        ws = _ws(x.device, L.lib().prh_linear_forward_workspace_bytes(rows, k, n))
        r2 = None
        if resid is not None:
            _req_gpu_f32(resid, "residual")
            if tuple(resid.shape) != (*x.shape[:-1], n):
                raise RuntimeError(f"residual shape {tuple(resid.shape)} does not match the output {(*x.shape[:-1], n)}")
            r2 = resid.reshape(rows, n)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        # operand maxima of the split-fp16 cores: measured once here and kept for the backward
        # GEMMs, which read x (wgrad) and W^T (dgrad) again
        w_amax = None
        if n % 4 == 0 and L.lib().prh_linear_uses_operand_maxima(rows, k, n):
            if x_amax is None:
                x_amax = operand_absmax(x2)
            w_amax = operand_absmax(w)
        L.check(L.lib().prh_linear_forward_full(_p(x2), k, _p(w), _p(b), _p(r2), n, _p(y), rows, k, n,
                                                int(bool(relu)), _p(x_amax), _p(w_amax), float(dropout_p),
                                                int(seed) & 0xFFFFFFFF, _p(ws), ws.numel(),
                                                x.device.index, _stream(x.device)),
                "prh_linear_forward")
        ctx.has_resid = resid is not None
        ctx.w_amax = w_amax
        if relu:
            ctx.save_for_backward(x2, w, y)
        else:
            ctx.save_for_backward(x2, w)
        ctx.relu = bool(relu)
        ctx.x_amax = x_amax
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def _forward_bf16(ctx, x, w, b, relu, resid):
        """x is a bf16 activation of the bf16 mode (the encoder's `fused`): y fp32 on the bf16 core."""
        if not x.is_cuda:
            raise RuntimeError("pointnet_refine_amd.linear: bf16 input must be a GPU tensor")
        _req_gpu_f32(w, "weight")
        n, k = w.shape
        if x.shape[-1] != k or k % 8 or n % 8 or resid is not None or relu:
            raise RuntimeError("pointnet_refine_amd.linear (bf16 input): needs in/out features % 8 == 0, no residual, no ReLU")
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = w.contiguous()
        rows = x2.shape[0]
        y = torch.empty((rows, n), dtype=torch.float32, device=x.device)
        ws = _ws(x.device, L.lib().prh_linear_bf16_workspace_bytes(rows, k, n, 0))
        L.check(L.lib().prh_linear_forward_bf16(_p(x2), k, _p(w), _p(b), _p(y), rows, k, n, 0, _p(ws), ws.numel(),
                                                x.device.index, _stream(x.device)), "prh_linear_forward_bf16")
        ctx.save_for_backward(x2, w)
        ctx.bf16_in, ctx.relu, ctx.has_resid, ctx.has_bias, ctx.xshape = True, False, False, b is not None, x.shape
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def _backward_bf16(ctx, dy):
        x2, w = ctx.saved_tensors
        n, k = w.shape
        rows, dev = x2.shape[0], x2.device
        dy2 = dy.reshape(rows, n).float().contiguous()
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_dx else None
        dw, tw = _grad_buf(ctx.w_sink, w.shape, dev, need_dw)
        db, tb = _grad_buf(ctx.b_sink, (n,), dev, need_db)
        ws = _ws(dev, L.lib().prh_linear_bf16_workspace_bytes(rows, k, n, 1))
        L.check(L.lib().prh_linear_backward_bf16(_p(x2), k, _p(w), _p(dy2), _p(dx), _p(dw), _p(db), rows, k, n, _p(ws),
                                                 ws.numel(), dev.index, _stream(dev)), "prh_linear_backward_bf16")
        return (dx.reshape(ctx.xshape) if need_dx else None), _grad_ret(dw, tw), _grad_ret(db, tb), None, None, None, None, None

    @staticmethod
    def backward(ctx, dy):
        if ctx.bf16_in:
            return LinearFn._backward_bf16(ctx, dy)
        if ctx.relu:
            x2, w, y = ctx.saved_tensors
        else:
            x2, w = ctx.saved_tensors
        n, k = w.shape
        rows = x2.shape[0]
        if n % 4:
            raise RuntimeError("pointnet_refine_amd.linear backward: out_features must be a multiple of 4")
        dy2 = dy.reshape(rows, n)
        hint = _DY_AMAX.pop(dy.data_ptr(), None) if _DY_AMAX else None
        dy_amax = hint[0] if (hint is not None and hint[1] == tuple(dy.shape) and dy2.data_ptr() == dy.data_ptr()) else None
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dev = x2.device
        if ctx.relu:          # mask by the saved output and take max|masked dy| in the same pass
            if dy2.dtype != torch.float32:
                dy2 = dy2.float()
            masked = torch.empty_like(dy2)
            dy_amax = torch.empty(1, dtype=torch.float32, device=dev)
            wsm = _ws(dev, L.lib().prh_operand_absmax_workspace_bytes())
            L.check(L.lib().prh_relu_mask_absmax(_p(dy2), _p(y), _p(masked), dy2.numel(), float(ctx.drop_scale), _p(dy_amax), _p(wsm),
                                                 wsm.numel(), dev.index, _stream(dev)), "prh_relu_mask_absmax")
            dy2 = masked
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_dx else None
        dw, tw = _grad_buf(ctx.w_sink, w.shape, dev, need_dw)
        db, tb = _grad_buf(ctx.b_sink, (n,), dev, need_db)
        nb = L.lib().prh_linear_backward_workspace_bytes(rows, k, n)
        ws = _ws(dev, nb)
        L.check(L.lib().prh_linear_backward_full(_p(x2), k, _p(w), _p(dy2), _p(dx), _p(dw), _p(db), rows,
                                                 k, n, _p(ctx.x_amax), _p(dy_amax), _p(ctx.w_amax), _p(ws),
                                                 ws.numel(), dev.index, _stream(dev)),
                "prh_linear_backward")
        dres = None
        if ctx.has_resid and ctx.needs_input_grad[5]:
            dres = dy2.reshape(*ctx.xshape[:-1], n)      # (masked by the ReLU when there is one)
        return (dx.reshape(ctx.xshape) if need_dx else None), _grad_ret(dw, tw), _grad_ret(db, tb), None, None, dres, None, None


def linear(x, w, b=None, x_amax=None, relu=False, resid=None, dropout_p=0.0, seed=0):
    return LinearFn.apply(x, w, b, x_amax, relu, resid, dropout_p, seed)


class LinearOut16Fn(torch.autograd.Function):
    """y (bf16) = x W^T + b from an fp32 x, with a bf16 gradient coming back: the wide cross-attention
    key / value projections of the bf16 mode (src/model.py:123-126 for all six layers at once)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.w_sink, ctx.b_sink = _sink_view(w), _sink_view(b)
        _req_gpu_f32(x, "input")
        _req_gpu_f32(w, "weight")
        n, k = w.shape
        if x.shape[-1] != k or k % 8 or n % 8:
            raise RuntimeError("pointnet_refine_amd.linear_out16: in/out features must be multiples of 8")
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = w.contiguous()
        rows = x2.shape[0]
        y = torch.empty((rows, n), dtype=torch.bfloat16, device=x.device)
        ws = _ws(x.device, L.lib().prh_linear_bf16_workspace_bytes(rows, k, n, 0))
        L.check(L.lib().prh_linear_forward_out16(_p(x2), k, _p(w), _p(b), _p(y), rows, k, n, _p(ws), ws.numel(),
                                                 x.device.index, _stream(x.device)), "prh_linear_forward_out16")
        ctx.save_for_backward(x2, w)
        ctx.has_bias, ctx.xshape = b is not None, x.shape
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        n, k = w.shape
        rows, dev = x2.shape[0], x2.device
        dy2 = dy.reshape(rows, n)
        if dy2.dtype != torch.bfloat16 or not dy2.is_contiguous():
            dy2 = dy2.to(torch.bfloat16).contiguous()
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_dx else None
        dw, tw = _grad_buf(ctx.w_sink, w.shape, dev, need_dw)
        db, tb = _grad_buf(ctx.b_sink, (n,), dev, need_db)
        ws = _ws(dev, L.lib().prh_linear_bf16_workspace_bytes(rows, k, n, 1))
        L.check(L.lib().prh_linear_backward_dy16(_p(x2), k, _p(w), _p(dy2), _p(dx), _p(dw), _p(db), rows, k, n, _p(ws),
                                                 ws.numel(), dev.index, _stream(dev)), "prh_linear_backward_dy16")
        return (dx.reshape(ctx.xshape) if need_dx else None), _grad_ret(dw, tw), _grad_ret(db, tb)


def linear_out16(x, w, b=None):
    return LinearOut16Fn.apply(x, w, b)


class PosHiddenFn(torch.autograd.Function):
    """h = relu(xyz W0^T + b0) for 3-wide points, one elementwise HIP pass (first layer of the
    positional-encoding MLP, src/model.py:64-75).  xyz may be a (..., 3) view of wider rows
    (context[:, :, :3]); it is read in place.  Point gradients (the decoder's query positions)
    for hidden <= 256."""

    @staticmethod
    def forward(ctx, xyz, w0, b0):
        ctx.w_sink, ctx.b_sink = _sink_view(w0), _sink_view(b0)
        _req_gpu_f32(xyz, "input")
        _req_gpu_f32(w0, "weight")
        hdim = w0.shape[0]
        if xyz.shape[-1] != 3 or w0.shape[1] != 3:
            raise RuntimeError(f"pos_hidden: expected 3-wide points and a (H, 3) weight, got {tuple(xyz.shape)} and {tuple(w0.shape)}")
        rows = xyz.numel() // 3
        # rows must be evenly strided: (B, N, 3) view of contiguous (B, N, C) rows, or contiguous
        ld = xyz.stride(-2) if xyz.dim() >= 2 else 3
        ok = xyz.stride(-1) == 1 and ld >= 3
        for d in range(xyz.dim() - 2):
            ok = ok and xyz.stride(d) == xyz.stride(d + 1) * xyz.shape[d + 1]
        if not ok:
            xyz = xyz.contiguous()
            ld = 3
        w0 = w0.contiguous()
        h = torch.empty((*xyz.shape[:-1], hdim), dtype=torch.float32, device=xyz.device)
        L.check(L.lib().prh_pos_hidden_forward(_p(xyz), ld, _p(w0), _p(b0), _p(h), rows, hdim,
                                               xyz.device.index, _stream(xyz.device)), "prh_pos_hidden_forward")
        ctx.save_for_backward(xyz, h, w0)
        ctx.ld, ctx.has_bias = ld, b0 is not None
        return h

    @staticmethod
    def backward(ctx, dh):
        xyz, h, w0 = ctx.saved_tensors
        hdim = h.shape[-1]
        rows = h.numel() // hdim
        dev = h.device
        dh = dh.contiguous()
        need_dx = ctx.needs_input_grad[0]
        if need_dx and hdim > 256:
            raise RuntimeError("pos_hidden: gradient with respect to the points is implemented for hidden <= 256")
        need_dw, need_db = ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty((*xyz.shape[:-1], 3), dtype=torch.float32, device=dev) if need_dx else None
        dw, tw = _grad_buf(ctx.w_sink, (hdim, 3), dev, need_dw)
        db, tb = _grad_buf(ctx.b_sink, (hdim,), dev, need_db)
        if need_dx or need_dw or need_db:
            ws = _ws(dev, L.lib().prh_pos_hidden_backward_workspace_bytes(rows, hdim))
            L.check(L.lib().prh_pos_hidden_backward(_p(xyz), ctx.ld, _p(h), _p(dh), _p(w0), _p(dx), _p(dw), _p(db),
                                                    rows, hdim, _p(ws), ws.numel(), dev.index, _stream(dev)),
                    "prh_pos_hidden_backward")
        return dx, _grad_ret(dw, tw), _grad_ret(db, tb)


class LinearSmallFn(torch.autograd.Function):
    """nn.Linear with <= 4 outputs as one HBM pass (regression heads' Linear(128, 3),
    src/model.py:162-166)."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.w_sink, ctx.b_sink = _sink_view(w), _sink_view(b)
        _req_gpu_f32(x, "input")
        _req_gpu_f32(w, "weight")
        n, k = w.shape
        if x.shape[-1] != k:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied ({x.shape} and {k}x{n})")
        x2 = x.reshape(-1, k)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        w = w.contiguous()
        rows = x2.shape[0]
        y = torch.empty((rows, n), dtype=torch.float32, device=x.device)
        L.check(L.lib().prh_linear_small_forward(_p(x2), _p(w), _p(b), _p(y), rows, k, n, x.device.index,
                                                 _stream(x.device)), "prh_linear_small_forward")
        ctx.save_for_backward(x2, w)
        ctx.has_bias, ctx.xshape = b is not None, x.shape
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        n, k = w.shape
        rows = x2.shape[0]
        dev = x2.device
        dy2 = dy.reshape(rows, n).contiguous()
        need_dx, need_dw, need_db = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_dx else None
        dw, tw = _grad_buf(ctx.w_sink, w.shape, dev, need_dw)
        db, tb = _grad_buf(ctx.b_sink, (n,), dev, need_db)
        ws = _ws(dev, L.lib().prh_linear_small_backward_workspace_bytes(rows, k, n))
        L.check(L.lib().prh_linear_small_backward(_p(x2), _p(w), _p(dy2), _p(dx), _p(dw), _p(db), rows, k, n,
                                                  _p(ws), ws.numel(), dev.index, _stream(dev)),
                "prh_linear_small_backward")
        return (dx.reshape(ctx.xshape) if need_dx else None), _grad_ret(dw, tw), _grad_ret(db, tb)


def linear_small(x, w, b=None):
    return LinearSmallFn.apply(x, w, b)


def linear_small_supported(k, n):
    lpr = k // 4
    return 1 <= n <= 4 and k % 4 == 0 and 1 <= lpr <= 64 and (lpr & (lpr - 1)) == 0


def pos_hidden(xyz, w0, b0=None):
    return PosHiddenFn.apply(xyz, w0, b0)


def pos_hidden_supported(hidden):
    return 4 <= hidden <= 1024 and (hidden & (hidden - 1)) == 0


# ------------------------------------------------------------------------------------------
# Shared-MLP stack (Conv1d k=1 + BatchNorm1d [+ ReLU]) x L    -- point_mlp
# ------------------------------------------------------------------------------------------
class MlpStackFn(torch.autograd.Function):
    """x (P,cin) -> y (P,cout_last).  params = [w,b,gamma,beta]*L (w as (cout,cin)),
    buffers = [running_mean, running_var, num_batches_tracked]*L (updated in train mode)."""

    @staticmethod
    def forward(ctx, x, relu_last, training, momentum, eps, buffers, *params):
        _req_gpu_f32(x, "input")
        nl = len(params) // 4
        dev = x.device
        x = x.contiguous()
        P = x.shape[0]
        ws_ = [params[4 * l].reshape(params[4 * l].shape[0], -1).contiguous() for l in range(nl)]
        if x.shape[1] != ws_[0].shape[1]:
            raise RuntimeError(f"expected input to have {ws_[0].shape[1]} channels, but got {x.shape[1]} channels instead")
        if training and P < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {tuple(x.shape)}")
        layers = (L.BnLayer * nl)()
        for l in range(nl):
            _req_gpu_f32(ws_[l], "weight")
            layers[l] = _bn_layer(ws_[l], params[4 * l + 1], params[4 * l + 2], params[4 * l + 3],
                                  buffers[3 * l], buffers[3 * l + 1], buffers[3 * l + 2])
        ctot = sum(w.shape[0] for w in ws_)
        z_cat = torch.empty((P, ctot), dtype=torch.float32, device=dev)
        y = torch.empty((P, ws_[-1].shape[0]), dtype=torch.float32, device=dev)
        coef = torch.empty((4, ctot), dtype=torch.float32, device=dev)
        nb = L.lib().prh_mlp_stack_workspace_bytes(P, nl, layers)
        ws = _ws(dev, nb)
        L.check(L.lib().prh_mlp_stack_forward(layers, nl, int(relu_last), _p(x), P, int(training),
                                              float(momentum), float(eps), _p(z_cat), _p(y),
                                              _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]),
                                              _p(ws), ws.numel(), dev.index, _stream(dev)),
                "prh_mlp_stack_forward")
        ctx.save_for_backward(x, z_cat, coef, *ws_, *[params[4 * l + 2] for l in range(nl)])
        ctx.nl, ctx.relu_last, ctx.training = nl, int(relu_last), int(training)
        ctx.wshapes = [params[4 * l].shape for l in range(nl)]
        ctx.sinks = [_sink_view(t) for t in params]
        return y

    @staticmethod
    def backward(ctx, dy):
        saved = ctx.saved_tensors
        nl = ctx.nl
        x, z_cat, coef = saved[0], saved[1], saved[2]
        ws_ = saved[3:3 + nl]
        gammas = saved[3 + nl:3 + 2 * nl]
        dev = x.device
        P = x.shape[0]
        dy = dy.contiguous()
        layers = (L.BnLayer * nl)()
        grads = (L.BnLayerGrad * nl)()
        outs: List[Optional[torch.Tensor]] = []
        for l in range(nl):
            w = ws_[l]
            layers[l] = L.BnLayer(_p(w), _p(gammas[l]), _p(gammas[l]), _p(gammas[l]), _p(gammas[l]),
                                  _p(gammas[l]), None, w.shape[1], w.shape[0])
            co = w.shape[0]
            bufs = [_grad_buf(ctx.sinks[4 * l + i], w.shape if i == 0 else (co,), dev) for i in range(4)]
            grads[l] = L.BnLayerGrad(*[_p(b[0]) for b in bufs])
            outs += bufs
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        nb = L.lib().prh_mlp_stack_workspace_bytes(P, nl, layers)
        ws = _ws(dev, nb)
        L.check(L.lib().prh_mlp_stack_backward(layers, nl, ctx.relu_last, _p(x), P, ctx.training,
                                               _p(dy), _p(z_cat), _p(coef[0]), _p(coef[1]),
                                               _p(coef[2]), _p(coef[3]), grads, _p(dx), _p(ws),
                                               ws.numel(), dev.index, _stream(dev)),
                "prh_mlp_stack_backward")
        rets = []
        for i, (buf, tok) in enumerate(outs):
            g = _grad_ret(buf, tok)
            rets.append(g.reshape(ctx.wshapes[i // 4]) if (g is not None and i % 4 == 0) else g)
        return (dx, None, None, None, None, None, *rets)


def mlp_stack(x, layers: Sequence[Sequence[torch.Tensor]], buffers: Sequence[torch.Tensor],
              relu_last: bool, training: bool, momentum: float, eps: float):
    flat = [t for ly in layers for t in ly]
    return MlpStackFn.apply(x, relu_last, training, momentum, eps, list(buffers), *flat)


# ------------------------------------------------------------------------------------------
# MultiScalePointNetEncoder
# ------------------------------------------------------------------------------------------
ENC_PARAM_ORDER = (
    [f"conv{k}.{n}" for k in range(1, 6) for n in ("weight", "bias")]
    + [f"bn{k}.{n}" for k in range(1, 6) for n in ("weight", "bias")]
    + ["fusion.0.weight", "fusion.0.bias", "fusion.1.weight", "fusion.1.bias",
       "intensity_gate.0.weight", "intensity_gate.0.bias", "intensity_gate.2.weight",
       "intensity_gate.2.bias"])


def _enc_params_struct(params, buffers, C_in):
    """params in ENC_PARAM_ORDER (conv weights already 2-D contiguous); buffers =
    [rm, rv, nbt] * 6 (bn1..5, fusion.1) or None (backward: statistics not touched)."""
    prm = L.EncoderParams()
    prm.in_channel = C_in
    out_dim = params[8].shape[0]
    prm.out_dim = out_dim
    for k in range(5):
        w, b = params[2 * k], params[2 * k + 1]
        g, beta = params[10 + 2 * k], params[11 + 2 * k]
        if buffers is not None:
            rm, rv, nbt = buffers[3 * k], buffers[3 * k + 1], buffers[3 * k + 2]
        else:
            rm, rv, nbt = g, g, None
        prm.conv[k] = _bn_layer(w, b, g, beta, rm, rv, nbt)
    if buffers is not None:
        rm, rv, nbt = buffers[15], buffers[16], buffers[17]
    else:
        rm, rv, nbt = params[22], params[22], None
    prm.fusion = _bn_layer(params[20], params[21], params[22], params[23], rm, rv, nbt)
    prm.gate_w1, prm.gate_b1 = _p(params[24]), _p(params[25])
    prm.gate_w2, prm.gate_b2 = _p(params[26]), _p(params[27])
    return prm


class EncoderFn(torch.autograd.Function):
    """MultiScalePointNetEncoder.forward (src/model.py:39-62) on point-major input.
    x (B,N,C) -> (gfeat (B,2*out) or None, fused (B,N,out))."""

    @staticmethod
    def forward(ctx, x, want_global, training, momentum, eps, buffers, *params):
        _req_gpu_f32(x, "input")
        dev = x.device
        x = x.contiguous()
        B, N, Cin = x.shape
        p2 = list(params)
        for i in (0, 2, 4, 6, 8, 20, 24, 26):    # Conv1d weights (cout,cin,1) -> (cout,cin)
            p2[i] = params[i].reshape(params[i].shape[0], -1).contiguous()
        for t in p2:
            _req_gpu_f32(t, "parameter")
        if Cin != p2[0].shape[1]:
            raise RuntimeError(f"Given groups=1, weight of size {list(params[0].shape)}, expected input"
                               f"[{B}, {Cin}, {N}] to have {p2[0].shape[1]} channels, but got {Cin} channels instead")
        if training and B * N < 2:
            raise ValueError(f"Expected more than 1 value per channel when training, got input size {(B, Cin, N)}")
        if training:
            invalidate_fused_images()          # the running statistics are about to change
        out_dim = p2[8].shape[0]
        cat = 64 + 128 + 256 + 512 + out_dim
        P = B * N
        need_bwd = any(ctx.needs_input_grad)
        ctx.bf16 = L.lib().prh_get_gemm_mode() == 4
        if ctx.bf16:
            return EncoderFn._forward_bf16(ctx, x, want_global, training, momentum, eps, buffers, params, p2, need_bwd)
        z_cat = torch.empty((P, cat), dtype=torch.float32, device=dev)
        z_fus = torch.empty((P, out_dim), dtype=torch.float32, device=dev)
        gate = torch.empty((P, out_dim), dtype=torch.float32, device=dev) if need_bwd else None
        coef = torch.empty((5, cat + out_dim), dtype=torch.float32, device=dev)   # row 4: operand maxima
        fused = torch.empty((B, N, out_dim), dtype=torch.float32, device=dev)
        gfeat = torch.empty((B, 2 * out_dim), dtype=torch.float32, device=dev) if want_global else None
        argmax = torch.empty((B, out_dim), dtype=torch.int32, device=dev) if (want_global and need_bwd) else None
        prm = _enc_params_struct(p2, buffers, Cin)
        gemm_mode = L.lib().prh_get_gemm_mode()
        sv = L.EncoderSaved(_p(z_cat), _p(z_fus), _p(gate), _p(coef[0]), _p(coef[1]), _p(coef[2]),
                            _p(coef[3]), _p(argmax), _p(coef[4]) if (training and need_bwd) else None)
        nb = L.lib().prh_encoder_workspace_bytes(B, N, Cin, out_dim, 0)
        ws = _ws(dev, nb)
        L.check(L.lib().prh_encoder_forward(C.byref(prm), _p(x), B, N, int(training), float(momentum),
                                            float(eps), _p(fused), _p(gfeat), C.byref(sv), _p(ws),
                                            ws.numel(), dev.index, _stream(dev)),
                "prh_encoder_forward")
        if need_bwd:
            ctx.save_for_backward(x, z_cat, z_fus, gate, coef, argmax if argmax is not None else coef, *p2)
            ctx.has_argmax = argmax is not None
            ctx.training = int(training)
            ctx.gemm_mode = gemm_mode
            ctx.pshapes = [t.shape for t in params]
            ctx.sinks = [_sink_view(t) for t in params]
            ctx.consumed = False
        # bound of max(fused) from the fusion statistics (training, split-fp16 cores), else None
        ctx_amax = coef[4, 6:7] if (training and need_bwd and gemm_mode == 3) else None
        EncoderFn.last_fused_amax = ctx_amax
        if want_global:
            return gfeat, fused
        return None, fused

    @staticmethod
    def _forward_bf16(ctx, x, want_global, training, momentum, eps, buffers, params, p2, need_bwd):
        """Mode 4 (BASELINE config 3): activations kept for the backward, and `fused` itself, in bf16."""
        dev = x.device
        B, N, Cin = x.shape
        out_dim = p2[8].shape[0]
        cat = 64 + 128 + 256 + 512 + out_dim
        P = B * N
        bf = torch.bfloat16
        z_cat = torch.empty((P, cat), dtype=bf, device=dev)
        z_fus = torch.empty((P, out_dim), dtype=bf, device=dev)
        gate = torch.empty((P, out_dim), dtype=bf, device=dev) if need_bwd else None
        coef = torch.empty((4, cat + out_dim), dtype=torch.float32, device=dev)
        fused = torch.empty((B, N, out_dim), dtype=bf, device=dev)
        gfeat = torch.empty((B, 2 * out_dim), dtype=torch.float32, device=dev) if want_global else None
        argmax = torch.empty((B, out_dim), dtype=torch.int32, device=dev) if (want_global and need_bwd) else None
        prm = _enc_params_struct(p2, buffers, Cin)
        sv = L.EncoderSavedBf16(_p(z_cat), _p(z_fus), _p(gate), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]),
                                _p(argmax))
        nb = L.lib().prh_encoder_bf16_workspace_bytes(B, N, Cin, out_dim, 0)
        ws = _ws(dev, nb)
        L.check(L.lib().prh_encoder_forward_bf16(C.byref(prm), _p(x), B, N, int(training), float(momentum), float(eps),
                                                 _p(fused), _p(gfeat), C.byref(sv), _p(ws), ws.numel(), dev.index,
                                                 _stream(dev)), "prh_encoder_forward_bf16")
        if need_bwd:
            ctx.save_for_backward(x, z_cat, z_fus, gate, coef, argmax if argmax is not None else coef, *p2)
            ctx.has_argmax = argmax is not None
            ctx.training = int(training)
            ctx.pshapes = [t.shape for t in params]
            ctx.sinks = [_sink_view(t) for t in params]
            ctx.consumed = False
        EncoderFn.last_fused_amax = None
        return (gfeat if want_global else None), fused

    @staticmethod
    def _backward_bf16(ctx, d_gfeat, d_fused):
        saved = ctx.saved_tensors
        x, z_cat, z_fus, gate, coef, argmax = saved[:6]
        if ENCODER_RETAIN_GRAPH:
            gate = gate.clone()
        p2 = list(saved[6:])
        dev = x.device
        B, N, Cin = x.shape
        out_dim = p2[8].shape[0]
        if d_fused is not None:
            d_fused = d_fused.to(torch.bfloat16).contiguous()
        if d_gfeat is not None:
            if not ctx.has_argmax:
                raise RuntimeError("encoder backward: gradient for global_feat but no argmax saved")
            d_gfeat = d_gfeat.float().contiguous()
        prm = _enc_params_struct(p2, None, Cin)
        sv = L.EncoderSavedBf16(_p(z_cat), _p(z_fus), _p(gate), _p(coef[0]), _p(coef[1]), _p(coef[2]), _p(coef[3]),
                                _p(argmax) if ctx.has_argmax else None)
        gb = [_grad_buf(ctx.sinks[i], t.shape, dev) for i, t in enumerate(p2)]
        g = [b_[0] for b_ in gb]
        gr = L.EncoderGrads()
        for k in range(5):
            gr.conv[k] = L.BnLayerGrad(_p(g[2 * k]), _p(g[2 * k + 1]), _p(g[10 + 2 * k]), _p(g[11 + 2 * k]))
        gr.fusion = L.BnLayerGrad(_p(g[20]), _p(g[21]), _p(g[22]), _p(g[23]))
        gr.d_gate_w1, gr.d_gate_b1, gr.d_gate_w2, gr.d_gate_b2 = _p(g[24]), _p(g[25]), _p(g[26]), _p(g[27])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        nb = L.lib().prh_encoder_bf16_workspace_bytes(B, N, Cin, out_dim, 1)
        ws = _ws(dev, nb)
        L.check(L.lib().prh_encoder_backward_bf16(C.byref(prm), _p(x), B, N, ctx.training, _p(d_fused), _p(d_gfeat),
                                                  C.byref(sv), C.byref(gr), _p(dx), _p(ws), ws.numel(), dev.index,
                                                  _stream(dev)), "prh_encoder_backward_bf16")
        grads = [_grad_ret(*b_) for b_ in gb]
        grads = [gi.reshape(s) if gi is not None else None for gi, s in zip(grads, ctx.pshapes)]
        return (dx, None, None, None, None, None, *grads)

    @staticmethod
    def backward(ctx, d_gfeat, d_fused):
        if ctx.consumed:
            raise RuntimeError("pointnet_refine_amd encoder: backward through the same graph a second "
                               "time is not supported by default (the saved gate buffer, 4 KB per point, is "
                               "overwritten by its gradient); call ops.allow_encoder_retain_graph(True) before "
                               "the first backward to work on a copy instead")
        ctx.consumed = not ENCODER_RETAIN_GRAPH
        if ctx.bf16:
            return EncoderFn._backward_bf16(ctx, d_gfeat, d_fused)
        saved = ctx.saved_tensors
        x, z_cat, z_fus, gate, coef, argmax = saved[:6]
        if ENCODER_RETAIN_GRAPH:
            gate = gate.clone()          # the only saved buffer the backward writes (dG in place)
        p2 = list(saved[6:])
        dev = x.device
        B, N, Cin = x.shape
        out_dim = p2[8].shape[0]
        # the incoming d_fused (LineRefineNet: context_proj's dgrad output, which nobody else holds) becomes the
        # fusion layer's gradient scratch instead of 4 KB per point of workspace - unless it is a view, the
        # pooled gradient is there as well, or a second backward was asked for
        scratch = (d_fused is not None and d_gfeat is None and d_fused.is_contiguous() and not ENCODER_RETAIN_GRAPH
                   and d_fused.dtype == torch.float32
                   and (d_fused._base is None or d_fused._base.numel() == d_fused.numel()))     # not a slice of something larger
        if d_fused is not None:
            d_fused = d_fused.contiguous()
        if d_gfeat is not None:
            if not ctx.has_argmax:
                raise RuntimeError("encoder backward: gradient for global_feat but no argmax saved")
            d_gfeat = d_gfeat.contiguous()
        prm = _enc_params_struct(p2, None, Cin)
        same_mode = ctx.training and ctx.gemm_mode == L.lib().prh_get_gemm_mode()
        sv = L.EncoderSaved(_p(z_cat), _p(z_fus), _p(gate), _p(coef[0]), _p(coef[1]), _p(coef[2]),
                            _p(coef[3]), _p(argmax) if ctx.has_argmax else None,
                            _p(coef[4]) if same_mode else None)
        gb = [_grad_buf(ctx.sinks[i], t.shape, dev) for i, t in enumerate(p2)]
        g = [b_[0] for b_ in gb]
        gr = L.EncoderGrads()
        for k in range(5):
            gr.conv[k] = L.BnLayerGrad(_p(g[2 * k]), _p(g[2 * k + 1]), _p(g[10 + 2 * k]), _p(g[11 + 2 * k]))
        gr.fusion = L.BnLayerGrad(_p(g[20]), _p(g[21]), _p(g[22]), _p(g[23]))
        gr.d_gate_w1, gr.d_gate_b1, gr.d_gate_w2, gr.d_gate_b2 = _p(g[24]), _p(g[25]), _p(g[26]), _p(g[27])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        nb = L.lib().prh_encoder_workspace_bytes(B, N, Cin, out_dim, 2 if scratch else 1)
        ws = _ws(dev, nb)
        L.check(L.lib().prh_encoder_backward(C.byref(prm), _p(x), B, N, ctx.training, _p(d_fused),
                                             _p(d_gfeat), int(scratch), C.byref(sv), C.byref(gr), _p(dx), _p(ws),
                                             ws.numel(), dev.index, _stream(dev)),
                "prh_encoder_backward")
        grads = [_grad_ret(*b_) for b_ in gb]
        grads = [gi.reshape(s) if gi is not None else None for gi, s in zip(grads, ctx.pshapes)]
        return (dx, None, None, None, None, None, *grads)


ENCODER_RETAIN_GRAPH = False


def allow_encoder_retain_graph(flag: bool = True):
    """loss.backward(retain_graph=True) followed by a second backward through the same encoder call: the
    reference allows it (stock autograd keeps every saved tensor intact).  Here the saved gate buffer is
    overwritten by its gradient to save 17 GB at B=4096; with this switch on, the backward works on a copy."""
    global ENCODER_RETAIN_GRAPH
    ENCODER_RETAIN_GRAPH = bool(flag)


def encoder(x_pm, params, buffers, want_global: bool, training: bool, momentum: float, eps: float):
    return EncoderFn.apply(x_pm, want_global, training, momentum, eps, list(buffers), *params)


def encoder_with_amax(x_pm, params, buffers, want_global, training, momentum, eps):
    """encoder() plus a 1-element device tensor bounding max(fused) from above (None when the
    forward did not produce one): lets the Linear that consumes `fused` skip its read pass."""
    EncoderFn.last_fused_amax = None
    gfeat, fused = EncoderFn.apply(x_pm, want_global, training, momentum, eps, list(buffers), *params)
    amax, EncoderFn.last_fused_amax = EncoderFn.last_fused_amax, None
    return gfeat, fused, amax


# ------------------------------------------------------------------------------------------
# Fused eval-mode encoder (+ context_proj): one kernel, BatchNorm folded (csrc/prh_fused.hpp)
# ------------------------------------------------------------------------------------------
_FUSED_IMAGES = {}
_FUSED_SAT = {}           # device index -> int32[1] counter the fused kernel increments when it clamps an activation


def invalidate_fused_images():
    """Drop the cached weight images of encoder_eval_fused.  Not needed for correctness - the cache key
    carries a device-side fingerprint of every tensor the image is made of, so `p.data.copy_()`, EMA
    updates, broadcasts and raw-pointer writers are all seen - but it frees the images' memory."""
    _FUSED_IMAGES.clear()


def _tensor_key(t):
    return (t.data_ptr(), t._version, tuple(t.shape))


def _fingerprint(tensors):
    """L2 norms of the tensors, computed on the device (one multi-tensor launch) and read back in one
    copy: part of the image cache key.  Version counters miss writes through `.data` and through raw
    pointers; the values themselves do not."""
    ts = [t.detach() for t in tensors if t is not None and t.is_floating_point()]
    return tuple(torch.stack(torch._foreach_norm(ts)).tolist())


class FusedSaturation(RuntimeError):
    """The fused eval kernel clamped an activation at the fp16 maximum: its result was discarded."""


def fused_saturation(device, reset=True) -> int:
    """Number of activation groups the fused eval kernel has clamped at 65504 on `device` since the last
    reset (synchronises).  The kernel carries activations between layers as fp16 planes; post-BatchNorm
    values of a trained checkpoint are O(1), anything that reaches the clamp is flagged here."""
    dev = torch.device(device)
    t = _FUSED_SAT.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if t is None:
        return 0
    n = int(t.item())
    if n and reset:
        t.zero_()
    return n


def encoder_eval_fused_supported(params, proj_w=None):
    """The fused kernel is built for the reference's widths (src/model.py:10-37,147)."""
    widths = [params[2 * k].shape[0] for k in range(5)]
    ok = widths == [64, 128, 256, 512, 1024] and params[20].shape[0] == 1024 and params[24].shape[0] == 64
    ok = ok and 4 <= params[0].shape[1] <= 64
    if proj_w is not None:
        ok = ok and tuple(proj_w.shape) == (256, 1024)
    return ok


def encoder_eval_fused(x_pm, params, buffers, eps, proj_w=None, proj_b=None, want_fused=False, want_global=False,
                       precision="fp32", check=True):
    """Eval-mode MultiScalePointNetEncoder (+ context_proj) in ONE kernel with BatchNorm folded into
    the weights (src/model.py:39-62 in eval mode, :194): x (B,N,C) -> (memory (B,N,256) or None,
    fused (B,N,1024) or None, global_feat (B,2048) or None).  precision "fp32": two fp16 planes,
    three products, fp32-level error; "fp16": one fp16 plane (BASELINE config 5).  Inference only
    (no autograd).  The folded / split weight image is prepared once per set of weights and cached
    (keyed on the tensors' addresses, version counters and a device-side fingerprint of their values).
    check=True: raises FusedSaturation when an activation exceeded the fp16 range inside the kernel (one
    4-byte read back); check=False leaves that to a later ops.fused_saturation(device) of the caller."""
    _req_gpu_f32(x_pm, "input")
    if precision not in ("fp32", "fp16"):
        raise ValueError("encoder_eval_fused: precision must be 'fp32' or 'fp16'")
    planes = 2 if precision == "fp32" else 1
    dev = x_pm.device
    x = x_pm.contiguous()
    B, N, Cin = x.shape
    p2 = list(params)
    for i in (0, 2, 4, 6, 8, 20, 24, 26):    # Conv1d weights (cout,cin,1) -> (cout,cin)
        p2[i] = params[i].reshape(params[i].shape[0], -1)
    if Cin != p2[0].shape[1]:
        raise RuntimeError(f"Given groups=1, weight of size {list(params[0].shape)}, expected input"
                           f"[{B}, {Cin}, {N}] to have {p2[0].shape[1]} channels, but got {Cin} channels instead")
    if not encoder_eval_fused_supported(p2, proj_w):
        raise RuntimeError("encoder_eval_fused: built for widths 64/128/256/512/1024, gate hidden 64, context_proj 256x1024")
    lib = L.lib()
    key = (dev.index, planes, float(eps), tuple(_tensor_key(t) for t in params),
           tuple(_tensor_key(t) for t in buffers if t.is_floating_point()),
           None if proj_w is None else (_tensor_key(proj_w), None if proj_b is None else _tensor_key(proj_b)),
           _fingerprint(list(params) + list(buffers) + [proj_w, proj_b]))
    slot = (dev.index, planes, id(params[0]), proj_w is not None)
    ent = _FUSED_IMAGES.get(slot)
    if ent is None or ent[0] != key:
        p3 = [t.contiguous() for t in p2]
        prm = _enc_params_struct(p3, buffers, Cin)
        nb = lib.prh_encoder_fused_image_bytes(planes, Cin)
        img = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
        off = (-img.data_ptr()) % 256
        pw = proj_w.contiguous() if proj_w is not None else None
        L.check(lib.prh_encoder_fused_prepare(C.byref(prm), float(eps), _p(pw), _p(proj_b), planes,
                                              C.c_void_p(img.data_ptr() + off), nb, dev.index, _stream(dev)),
                "prh_encoder_fused_prepare")
        ent = (key, img, off)
        _FUSED_IMAGES[slot] = ent
    _, img, off = ent
    memory = torch.empty((B, N, 256), dtype=torch.float32, device=dev) if proj_w is not None else None
    fused = torch.empty((B, N, 1024), dtype=torch.float32, device=dev) if want_fused else None
    gfeat = torch.empty((B, 2048), dtype=torch.float32, device=dev) if want_global else None
    ws = _ws(dev, lib.prh_encoder_fused_workspace_bytes(B, N, planes)) if want_global else None
    sat = _FUSED_SAT.get(dev.index)
    if sat is None:
        sat = _FUSED_SAT[dev.index] = torch.zeros(1, dtype=torch.int32, device=dev)
    L.check(lib.prh_encoder_fused_forward(C.c_void_p(img.data_ptr() + off), planes, Cin, int(proj_w is not None), _p(x),
                                          B, N, _p(memory), _p(fused), _p(gfeat), _p(sat), _p(ws),
                                          ws.numel() if ws is not None else 0, dev.index, _stream(dev)),
            "prh_encoder_fused_forward")
    if check:
        n = fused_saturation(dev)
        if n:
            raise FusedSaturation(f"encoder_eval_fused: {n} activation groups exceeded the fp16 range (65504) inside the "
                                  "fused kernel; use the per-layer kernels (inference_precision=None) for these weights")
    return memory, fused, gfeat


# ------------------------------------------------------------------------------------------
# Fused cross-attention core (between in-projections and out-projection)
# ------------------------------------------------------------------------------------------
def _rows_view(t: torch.Tensor, name: str):
    """(B, L, C) tensor whose rows are contiguous and equally spaced (possibly a column block
    of a wider buffer) -> (data_ptr-carrying tensor, leading dimension)."""
    if t.dim() != 3 or t.stride(2) != 1 or t.stride(0) != t.shape[1] * t.stride(1):
        t = t.contiguous()
    return t, t.stride(1)


# gradient tensor address -> (1-element bound of its largest magnitude, shape): filled by the
# producer of a gradient that knows the bound (the attention backward), consumed once by
# LinearFn.backward
_DY_AMAX = {}


KV_GRAD_IN_PLACE = True     # False: the K / V projection gradients get buffers of their own (A/B, debugging)


class GradArena:
    """Gradient buffers of the batched K/V projections, filled block by block by the
    attention backward of each decoder layer (no per-layer dK/dV tensors, no concatenation)."""

    def __init__(self):
        self.dk = None
        self.dv = None
        self.written = set()
        self.part = None          # [n_blocks][waves][2]: largest |dV|, |dK| each attention backward stored


class KVTokenFn(torch.autograd.Function):
    """Ties the arena to autograd: returns a scalar token every attention call takes as an
    input; its backward runs after ALL those calls' backwards and hands the arena buffers on
    as the gradients of k_all / v_all."""

    @staticmethod
    def forward(ctx, k_all, v_all, arena, block):
        ctx.arena, ctx.block = arena, block
        ctx.shape, ctx.dev, ctx.dtype = tuple(k_all.shape), k_all.device, k_all.dtype
        return torch.zeros((), dtype=torch.float32, device=k_all.device)

    @staticmethod
    def backward(ctx, dtoken):
        a = ctx.arena
        nblk = ctx.shape[-1] // ctx.block
        if a.dk is None:
            a.dk = torch.zeros(ctx.shape, dtype=ctx.dtype, device=ctx.dev)
            a.dv = torch.zeros(ctx.shape, dtype=ctx.dtype, device=ctx.dev)
        else:
            for i in range(nblk):
                if i not in a.written:
                    a.dk[..., i * ctx.block:(i + 1) * ctx.block].zero_()
                    a.dv[..., i * ctx.block:(i + 1) * ctx.block].zero_()
        dk, dv = a.dk, a.dv
        if a.part is not None:
            # bounds of max|dV|, max|dK| for the K/V projections' backward GEMMs: picked up by
            # LinearFn.backward through the gradient tensors' addresses (one reduction of the
            # per-wave maxima instead of a read pass over each gradient buffer)
            mx = a.part.amax(dim=(0, 1))
            _DY_AMAX.clear()      # at most this call's two hints are ever live (popped by their consumers)
            _DY_AMAX[dv.data_ptr()] = (mx[0:1], tuple(dv.shape))
            _DY_AMAX[dk.data_ptr()] = (mx[1:2], tuple(dk.shape))
        a.dk = a.dv = a.part = None
        a.written = set()
        return dk, dv, None, None


class AttentionFn(torch.autograd.Function):
    """o = dropout(softmax(q k^T / sqrt(32))) v per head of 32 channels; q (B,M,C), k/v (B,N,C).
    With (token, arena, block) the key/value operands are column block `block` of wide
    projection buffers and their gradients go straight into the arena."""

    @staticmethod
    def forward(ctx, q, k, v, heads, dropout_p, seed, token=None, arena=None, block=0):
        _req_gpu_f32(q, "query")
        kv16 = k.dtype == torch.bfloat16
        if kv16:
            if v.dtype != torch.bfloat16 or not (k.is_cuda and v.is_cuda) or arena is None:
                raise RuntimeError("pointnet_refine_amd.attention: bf16 K/V come as the wide projection buffers (arena mode)")
        else:
            _req_gpu_f32(k, "key")
            _req_gpu_f32(v, "value")
        B, M, Cq = q.shape
        N = k.shape[1]
        if Cq != heads * 32 or heads % 4:
            raise RuntimeError("pointnet_refine_amd.attention: expects heads of 32 channels, heads % 4 == 0")
        q = q.contiguous()
        if arena is not None:
            if not (k.is_contiguous() and v.is_contiguous() and k.shape == v.shape and k.shape[2] % Cq == 0):
                raise RuntimeError("pointnet_refine_amd.attention: arena mode needs contiguous (B,N,n*C) K/V")
            ldk = ldv = k.shape[2]
            koff = voff = block * Cq
        else:
            if k.shape[2] != Cq or v.shape[2] != Cq:
                raise RuntimeError("pointnet_refine_amd.attention: key/value width must equal the query width")
            k, ldk = _rows_view(k, "key")
            v, ldv = _rows_view(v, "value")
            koff = voff = 0
        o = torch.empty_like(q)
        lse = torch.empty((B, heads, M), dtype=torch.float32, device=q.device)
        scale = 1.0 / (32.0 ** 0.5)
        es = k.element_size()
        kp = C.c_void_p(k.data_ptr() + es * koff)
        vp = C.c_void_p(v.data_ptr() + es * voff)
        fwd = L.lib().prh_attn_forward_kv16 if kv16 else L.lib().prh_attn_forward
        L.check(fwd(_p(q), Cq, kp, ldk, vp, ldv, _p(o), Cq, _p(lse), B, M, N,
                    heads, scale, float(dropout_p), int(seed) & 0xFFFFFFFF,
                    q.device.index, _stream(q.device)), "prh_attn_forward")
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.kv16 = kv16
        ctx.cfg = (heads, float(dropout_p), int(seed) & 0xFFFFFFFF, scale, ldk, ldv, koff, voff)
        ctx.arena, ctx.block = arena, block
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        heads, dropout_p, seed, scale, ldk, ldv, koff, voff = ctx.cfg
        B, M, Cq = q.shape
        N = k.shape[1]
        do = do.contiguous()
        dq = torch.empty_like(q)
        a = ctx.arena
        if a is not None:
            if a.dk is None:
                if KV_GRAD_IN_PLACE and M <= 32 and not ENCODER_RETAIN_GRAPH:
                    # dK / dV of a layer's column block overwrite that block of the projection buffers: a wave
                    # reads a (32 keys x 32 channels) K / V tile exactly once - the next tile is already in
                    # registers when it stores the gradient tile - nothing reads the projections after the
                    # attention backward (their Linears saved the inputs, not the outputs), and with at most 32
                    # queries there is no second query tile that would read K / V again.  Saves 12 KB per context
                    # point of a decoder micro-batch (25.8 GB at 2,048 segments x 1,024 points).
                    a.dk, a.dv = k, v
                else:
                    a.dk = torch.empty_like(k)
                    a.dv = torch.empty_like(v)
            dk, dv, lddk = a.dk, a.dv, k.shape[2]
            es = k.element_size()
            dkp = C.c_void_p(dk.data_ptr() + es * koff)
            dvp = C.c_void_p(dv.data_ptr() + es * voff)
            a.written.add(ctx.block)
            nblk, waves = k.shape[2] // Cq, B * (heads // 4) * 4
            partp = None
            if not ctx.kv16:          # operand maxima for the split-fp16 backward GEMMs (not used in the bf16 mode)
                if a.part is None:
                    a.part = torch.zeros((nblk, waves, 2), dtype=torch.float32, device=q.device)
                partp = C.c_void_p(a.part.data_ptr() + 4 * ctx.block * waves * 2)
        else:
            dk = torch.empty((B, N, Cq), dtype=torch.float32, device=q.device)
            dv = torch.empty((B, N, Cq), dtype=torch.float32, device=q.device)
            lddk, dkp, dvp = Cq, _p(dk), _p(dv)
            partp = None
        es = k.element_size()
        kp = C.c_void_p(k.data_ptr() + es * koff)
        vp = C.c_void_p(v.data_ptr() + es * voff)
        if ctx.kv16:
            L.check(L.lib().prh_attn_backward_kv16(_p(q), Cq, kp, ldk, vp, ldv, _p(o), Cq, _p(lse), _p(do), Cq,
                                                   _p(dq), Cq, dkp, lddk, dvp, lddk, B, M, N, heads, scale,
                                                   dropout_p, seed, q.device.index, _stream(q.device)),
                    "prh_attn_backward_kv16")
        else:
            L.check(L.lib().prh_attn_backward_ex(_p(q), Cq, kp, ldk, vp, ldv, _p(o), Cq, _p(lse), _p(do), Cq,
                                                 _p(dq), Cq, dkp, lddk, dvp, lddk, B, M, N, heads, scale,
                                                 dropout_p, seed, partp, q.device.index, _stream(q.device)),
                    "prh_attn_backward")
        if a is not None:
            return dq, None, None, None, None, None, torch.zeros((), device=q.device), None, None
        return dq, dk, dv, None, None, None, None, None, None


def attention(q, k, v, heads: int, dropout_p: float = 0.0, seed: int = 0):
    return AttentionFn.apply(q, k, v, heads, dropout_p, seed)


class SelfAttentionPackedFn(torch.autograd.Function):
    """Self-attention whose queries and keys are the two halves of ONE projection output qk (B, M, 2C) - the packed
    in-projection of nn.MultiheadAttention applied to q = k = tgt + pos (src/model.py:104-110).  The kernels take
    leading dimensions, so the halves are read in place (no .contiguous() copy of the query half) and dq / dk are
    written as the two halves of ONE gradient buffer: autograd's two slice-backward fills and their sum (3 stock
    launches and 3 passes over (B, M, 2C) per layer) do not exist."""

    @staticmethod
    def forward(ctx, qk, v, heads, dropout_p, seed):
        _req_gpu_f32(qk, "packed query / key")
        _req_gpu_f32(v, "value")
        B, M, C2 = qk.shape
        Cq = C2 // 2
        if C2 != 2 * heads * 32 or heads % 4 or tuple(v.shape) != (B, M, Cq):
            raise RuntimeError("pointnet_refine_amd.attention_self_packed: expects qk (B, M, 2C), v (B, M, C), heads of 32 channels")
        qk = qk.contiguous()
        v, ldv = _rows_view(v, "value")
        o = torch.empty((B, M, Cq), dtype=torch.float32, device=qk.device)
        lse = torch.empty((B, heads, M), dtype=torch.float32, device=qk.device)
        scale = 1.0 / (32.0 ** 0.5)
        kp = C.c_void_p(qk.data_ptr() + 4 * Cq)
        L.check(L.lib().prh_attn_forward(_p(qk), C2, kp, C2, _p(v), ldv, _p(o), Cq, _p(lse), B, M, M, heads, scale,
                                         float(dropout_p), int(seed) & 0xFFFFFFFF, qk.device.index, _stream(qk.device)),
                "prh_attn_forward")
        ctx.save_for_backward(qk, v, o, lse)
        ctx.cfg = (heads, float(dropout_p), int(seed) & 0xFFFFFFFF, scale, ldv)
        return o

    @staticmethod
    def backward(ctx, do):
        qk, v, o, lse = ctx.saved_tensors
        heads, dropout_p, seed, scale, ldv = ctx.cfg
        B, M, C2 = qk.shape
        Cq = C2 // 2
        do = do.contiguous()
        dqk = torch.empty_like(qk)
        dv = torch.empty((B, M, Cq), dtype=torch.float32, device=qk.device)
        kp = C.c_void_p(qk.data_ptr() + 4 * Cq)
        dkp = C.c_void_p(dqk.data_ptr() + 4 * Cq)
        L.check(L.lib().prh_attn_backward_ex(_p(qk), C2, kp, C2, _p(v), ldv, _p(o), Cq, _p(lse), _p(do), Cq,
                                             _p(dqk), C2, dkp, C2, _p(dv), Cq, B, M, M, heads, scale,
                                             dropout_p, seed, None, qk.device.index, _stream(qk.device)),
                "prh_attn_backward")
        return dqk, dv, None, None, None


def attention_self_packed(qk, v, heads: int, dropout_p: float = 0.0, seed: int = 0):
    return SelfAttentionPackedFn.apply(qk, v, heads, dropout_p, seed)


def kv_token(k_all, v_all, block: int):
    """(token, arena) for attention_block(): call once per batched K/V projection pair."""
    arena = GradArena()
    return KVTokenFn.apply(k_all, v_all, arena, block), arena


def attention_block(q, k_all, v_all, token, arena, block_index: int, heads: int,
                    dropout_p: float = 0.0, seed: int = 0):
    """Attention against column block `block_index` of the wide projections k_all / v_all
    ((B,N,n*C), contiguous); their gradients are assembled in the arena (see kv_token)."""
    return AttentionFn.apply(q, k_all.detach(), v_all.detach(), heads, dropout_p, seed, token, arena,
                             block_index)


def cast_perm_bf16(x):
    """(.., 256) fp32 rows -> the bf16 row image `attention_folded` reads (channels permuted inside
    groups of 16, csrc/prh_attnfold.hpp); returned as an opaque (rows, 256) bfloat16 tensor."""
    _req_gpu_f32(x, "memory rows")
    if x.shape[-1] != 256:
        raise RuntimeError("cast_perm_bf16: rows of 256 channels expected")
    x2 = x.reshape(-1, 256)
    if x2.stride(-1) != 1 or x2.stride(0) % 4:
        x2 = x2.contiguous()
    out = torch.empty((x2.shape[0], 256), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().prh_cast_perm_bf16(_p(x2), x2.stride(0), _p(out), x2.shape[0], x.device.index, _stream(x.device)),
            "prh_cast_perm_bf16")
    return out


def posmem_images(xyz, memory, w0, b0, w2, b2):
    """The two row images `attention_folded` reads, in one pass from their sources (inference):
    x16 = bf16(memory + pos_emb(xyz)), y16 = bf16(memory), pos_emb = Linear(3,256) + ReLU +
    Linear(256,256) (src/model.py:64-75).  xyz (B,N,3) may be a view of wider rows (context[:, :, :3]);
    memory (B,N,256).  Returns two (B*N, 256) bfloat16 tensors (opaque channel order)."""
    _req_gpu_f32(xyz, "points")
    _req_gpu_f32(memory, "memory")
    if xyz.shape[-1] != 3 or memory.shape[-1] != 256 or tuple(w0.shape) != (256, 3) or tuple(w2.shape) != (256, 256):
        raise RuntimeError("posmem_images: 3-wide points, 256-wide memory, Linear(3,256) and Linear(256,256) expected")
    rows = memory.numel() // 256
    ld = xyz.stride(-2) if xyz.dim() >= 2 else 3
    ok = xyz.stride(-1) == 1 and ld >= 3 and xyz.numel() // 3 == rows
    for d in range(xyz.dim() - 2):
        ok = ok and xyz.stride(d) == xyz.stride(d + 1) * xyz.shape[d + 1]
    if not ok:
        xyz, ld = xyz.contiguous(), 3
    m2 = memory.reshape(rows, 256)
    if not m2.is_contiguous():
        m2 = m2.contiguous()
    x16 = torch.empty((rows, 256), dtype=torch.bfloat16, device=memory.device)
    y16 = torch.empty_like(x16)
    L.check(L.lib().prh_posmem_images(_p(xyz), ld, _p(w0.contiguous()), _p(b0), _p(w2.contiguous()), _p(b2), _p(m2), 256,
                                      rows, _p(x16), _p(y16), memory.device.index, _stream(memory.device)),
            "prh_posmem_images")
    return x16, y16


def attention_folded(q, x16, y16, wk, wv, bv, heads: int):
    """Inference-only cross-attention with the key / value projections folded in
    (src/model.py:119-128 in eval mode): q (B,M,256) projected queries, x16 / y16 = cast_perm_bf16 of
    memory + pos and of memory ((B*N, 256)), wk / wv (256,256) and bv (256) the key / value rows of the
    layer's packed in_proj parameters.  Returns the (B,M,256) attention output before out_proj.  No
    autograd; 8 heads of 32 channels, M <= 32, bf16 products (BASELINE config 5)."""
    _req_gpu_f32(q, "queries")
    B, M, Cq = q.shape
    if Cq != 256 or heads != 8 or M > 32:
        raise RuntimeError("attention_folded: 8 heads of 32 channels and at most 32 queries")
    if torch.is_grad_enabled() and (q.requires_grad or wk.requires_grad):
        raise RuntimeError("attention_folded is inference-only: call it under torch.no_grad()")
    N = x16.shape[0] // B
    if x16.shape != y16.shape or x16.shape[0] != B * N or x16.dtype != torch.bfloat16:
        raise RuntimeError("attention_folded: x16 / y16 must be cast_perm_bf16 images of (B*N, 256) rows")
    q2 = q.reshape(B * M, 256)
    if not q2.is_contiguous():
        q2 = q2.contiguous()
    wk, wv, bv = wk.contiguous(), wv.contiguous(), bv.contiguous()
    o = torch.empty((B * M, 256), dtype=torch.float32, device=q.device)
    L.check(L.lib().prh_attn_fold_forward(_p(q2), 256, _p(x16), _p(y16), _p(wk), wk.stride(0), _p(wv), wv.stride(0),
                                          _p(bv), _p(o), 256, B, M, N, heads, 1.0 / math.sqrt(32.0), q.device.index,
                                          _stream(q.device)), "prh_attn_fold_forward")
    return o.view(B, M, 256)


def attention_keep_mask(B, H, M, N, dropout_p, seed, device="cpu"):
    """The dropout keep-mask the kernels use (counter-based hash of seed, b*H+h, query, key),
    re-created with integer tensor ops; (B,H,M,N) bool.  For tests and reproducibility."""
    m32 = 0xFFFFFFFF
    bh = torch.arange(B * H, dtype=torch.int64, device=device).view(B, H, 1, 1)
    qq = torch.arange(M, dtype=torch.int64, device=device).view(1, 1, M, 1)
    kk = torch.arange(N, dtype=torch.int64, device=device).view(1, 1, 1, N)
    x = (int(seed) & m32) ^ ((bh * 0xC2B2AE3D) & m32) ^ ((qq * 0x9E3779B1) & m32) ^ ((kk * 0x85EBCA77) & m32)
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & m32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & m32
    x = x ^ (x >> 16)
    return x >= int(float(dropout_p) * 4294967296.0)


class DeepSupervisionL1Fn(torch.autograd.Function):
    """sum_{l,e} |pred[l,e] - target[e]| / denom with its gradient produced in the same pass
    (row f3; train.py:63-68, train_dist.py:180-186).  pred (L, ...), target (...)."""

    @staticmethod
    def forward(ctx, pred, target, denom, geometry=None, points=0.0):
        _req_gpu_f32(pred, "pred")
        _req_gpu_f32(target, "target")
        if pred.shape[1:] != target.shape:
            raise RuntimeError(f"deep_supervision_l1: pred {tuple(pred.shape)} vs target {tuple(target.shape)}")
        pred, target = pred.contiguous(), target.contiguous()
        dev = pred.device
        need = ctx.needs_input_grad[0]
        loss = torch.empty((), dtype=torch.float32, device=dev)
        d_pred = torch.empty_like(pred) if need else None
        nb = L.lib().prh_l1_loss_workspace_bytes()
        ws = _ws(dev, nb)
        L.check(L.lib().prh_l1_loss(_p(pred), _p(target), pred.shape[0], target.numel(), float(denom), 0, _p(loss),
                                    _p(d_pred), _p(geometry), float(points), _p(ws), ws.numel(), dev.index,
                                    _stream(dev)), "prh_l1_loss")
        if need:
            ctx.save_for_backward(d_pred)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_pred,) = ctx.saved_tensors
        return d_pred * g, None, None, None, None


def deep_supervision_l1(pred, target, denom=None, geometry=None, points=None):
    """(1/L) sum_l mean|pred_l - target| on the HIP path; denom defaults to pred.numel().
    geometry: optional float32 CUDA tensor of 2 elements that ACCUMULATES the step metrics of
    train_dist.py:190-203 - [0] += mean point-to-point error of the noisy line (|target|),
    [1] += that of the last layer's prediction - over `points` points (default: this call's)."""
    if geometry is not None:
        _req_gpu_f32(geometry, "geometry")
        points = float(target.numel() // 3 if points is None else points)
        # the C entry overwrites on accumulate=0: add this call's values to the caller's tensor
        g = torch.empty(2, dtype=torch.float32, device=pred.device)
        loss = DeepSupervisionL1Fn.apply(pred, target, float(pred.numel() if denom is None else denom), g, points)
        geometry += g
        return loss
    return DeepSupervisionL1Fn.apply(pred, target, float(pred.numel() if denom is None else denom))


class AddDropoutLayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + dropout(r)) over the last dimension (256), one HIP pass each way
    (row f1; src/model.py:117,128,133).  The dropout mask is a counter hash of (seed, row,
    channel), regenerated in the backward."""

    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps, p, seed):
        for t, name in ((x, "x"), (r, "r"), (gamma, "weight"), (beta, "bias")):
            _req_gpu_f32(t, name)
        if x.shape != r.shape or x.shape[-1] != 256 or gamma.numel() != 256:
            raise RuntimeError(f"add_dropout_layernorm: shapes {tuple(x.shape)}, {tuple(r.shape)} (last dim must be 256)")
        x2, r2 = x.contiguous().view(-1, 256), r.contiguous().view(-1, 256)
        rows, dev = x2.shape[0], x.device
        y = torch.empty_like(x2)
        need = any(ctx.needs_input_grad)
        mean = torch.empty(rows, dtype=torch.float32, device=dev) if need else None
        rstd = torch.empty(rows, dtype=torch.float32, device=dev) if need else None
        L.check(L.lib().prh_add_dropout_layernorm_forward(_p(x2), _p(r2), _p(gamma), _p(beta), rows, 256, float(eps),
                                                          float(p), int(seed), _p(y), _p(mean), _p(rstd), dev.index,
                                                          _stream(dev)), "prh_add_dropout_layernorm_forward")
        if need:
            ctx.save_for_backward(x2, r2, gamma, mean, rstd)
            ctx.p, ctx.seed, ctx.shape = float(p), int(seed), x.shape
            ctx.g_sink, ctx.b_sink = _sink_view(gamma), _sink_view(beta)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, r2, gamma, mean, rstd = ctx.saved_tensors
        dev, rows = x2.device, x2.shape[0]
        dy2 = dy.contiguous().view(-1, 256)
        dx, dr = torch.empty_like(x2), torch.empty_like(x2)
        (dg, tg), (db, tb) = _grad_buf(ctx.g_sink, gamma.shape, dev), _grad_buf(ctx.b_sink, gamma.shape, dev)
        nb = L.lib().prh_add_dropout_layernorm_workspace_bytes()
        ws = _ws(dev, nb)
        L.check(L.lib().prh_add_dropout_layernorm_backward(_p(dy2), _p(x2), _p(r2), _p(gamma), _p(mean), _p(rstd), rows,
                                                           256, ctx.p, ctx.seed, _p(dx), _p(dr), _p(dg), _p(db), _p(ws),
                                                           ws.numel(), dev.index, _stream(dev)),
                "prh_add_dropout_layernorm_backward")
        return dx.view(ctx.shape), dr.view(ctx.shape), _grad_ret(dg, tg), _grad_ret(db, tb), None, None, None


def add_dropout_layernorm(x, r, norm: torch.nn.LayerNorm, p: float):
    """norm(x + dropout(r, p)); p = 0 in eval mode.  A fresh seed per call from torch's CPU generator."""
    seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item()) if p > 0.0 else 0
    return AddDropoutLayerNormFn.apply(x, r, norm.weight, norm.bias, norm.eps, p, seed)


def layernorm_keep_mask(rows, channels, dropout_p, seed, device="cpu"):
    """The keep-mask of add_dropout_layernorm re-created with integer tensor ops (tests)."""
    m32 = 0xFFFFFFFF
    rr = torch.arange(rows, dtype=torch.int64, device=device).view(-1, 1)
    cc = torch.arange(channels, dtype=torch.int64, device=device).view(1, -1)
    x = (int(seed) & m32) ^ ((rr * 0x9E3779B1) & m32) ^ ((cc * 0x85EBCA77) & m32)
    x = x ^ (x >> 16)
    x = (x * 0x85EBCA6B) & m32
    x = x ^ (x >> 13)
    x = (x * 0xC2B2AE35) & m32
    x = x ^ (x >> 16)
    return x >= int(float(dropout_p) * 4294967296.0)
