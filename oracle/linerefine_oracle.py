"""CPU restatement of the LineRefineNet hot path -- the parity ORACLE.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product package
``pointnet_refine_amd`` never does (its ops raise when the HIP library is missing).

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
(``/root/reference/src/model.py``) in the build container, checks every function
below against it (eval forward, train-mode forward, all gradients, running-stat
updates) and writes the golden vectors under ``tests/golden``; the CPU test suite
re-checks this file against those vectors without the reference present.

The restatement is plain tensor math in POINT-MAJOR layout (rows = points, columns =
channels), i.e. the layout the HIP kernels use, rather than the reference's
``nn.Conv1d`` channel-major modules:

* a 1x1 ``Conv1d`` over ``(B,C,N)`` is ``X[B*N, Cin] @ W[Cout, Cin]^T + b``
* ``BatchNorm1d`` is written out (batch mean / biased variance over all B*N rows
  in train mode, running statistics in eval mode, eps 1e-5, momentum 0.1,
  unbiased variance into ``running_var``)
* ``nn.MultiheadAttention`` is written out (packed ``in_proj`` split [q;k;v],
  8 heads of 32 contiguous channels, scale 1/sqrt(32), softmax over keys)

Gradients come from torch autograd over this tensor math.

Every function cites the reference lines it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LN_EPS = 1e-5
NHEAD = 8


def _w2d(w: torch.Tensor) -> torch.Tensor:
    """Conv1d k=1 weight (Cout,Cin,1) -> (Cout,Cin)."""
    return w.reshape(w.shape[0], w.shape[1])


def batch_norm_rows(x, p, pre, training, new_stats: Optional[dict]):
    """BatchNorm1d over rows of x (P,C).  src/model.py:15-19,25 (nn.BatchNorm1d).
    train: batch mean / biased var; running_var gets the UNBIASED var;
    eval: running statistics."""
    g, b = p[pre + ".weight"], p[pre + ".bias"]
    if training:
        mean = x.mean(dim=0)
        var = x.var(dim=0, unbiased=False)
        if new_stats is not None:
            n = x.shape[0]
            with torch.no_grad():
                unb = var * (n / max(n - 1, 1))
                new_stats[pre + ".running_mean"] = (
                    (1 - BN_MOMENTUM) * p[pre + ".running_mean"] + BN_MOMENTUM * mean)
                new_stats[pre + ".running_var"] = (
                    (1 - BN_MOMENTUM) * p[pre + ".running_var"] + BN_MOMENTUM * unb)
                new_stats[pre + ".num_batches_tracked"] = p[pre + ".num_batches_tracked"] + 1
    else:
        mean, var = p[pre + ".running_mean"], p[pre + ".running_var"]
    return (x - mean) * torch.rsqrt(var + BN_EPS) * g + b


def encoder_forward(p: Dict[str, torch.Tensor], ctx_pm: torch.Tensor, prefix: str = "",
                    training: bool = False, new_stats: Optional[dict] = None):
    """MultiScalePointNetEncoder.forward, src/model.py:39-62, on point-major
    input ctx_pm (B,N,C).  Returns (global_feat (B,2*out), fused_pm (B,N,out));
    the reference returns fused as (B,out,N) = fused_pm.transpose(1,2)."""
    B, N, C = ctx_pm.shape
    x = ctx_pm.reshape(B * N, C)
    intensity = x[:, 3:4]                                             # :42
    feats = []
    h = x
    for k in range(1, 6):                                             # :43-47
        z = h @ _w2d(p[f"{prefix}conv{k}.weight"]).t() + p[f"{prefix}conv{k}.bias"]
        h = torch.relu(batch_norm_rows(z, p, f"{prefix}bn{k}", training, new_stats))
        feats.append(h)
    cat = torch.cat(feats, dim=1)                                     # :50
    zf = cat @ _w2d(p[f"{prefix}fusion.0.weight"]).t() + p[f"{prefix}fusion.0.bias"]
    fused = torch.relu(batch_norm_rows(zf, p, f"{prefix}fusion.1", training, new_stats))  # :51
    u = torch.relu(intensity @ _w2d(p[f"{prefix}intensity_gate.0.weight"]).t()
                   + p[f"{prefix}intensity_gate.0.bias"])
    gate = torch.sigmoid(u @ _w2d(p[f"{prefix}intensity_gate.2.weight"]).t()
                         + p[f"{prefix}intensity_gate.2.bias"])       # :54
    fused = fused * (0.5 + 0.5 * gate)                                # :55
    fused = fused.reshape(B, N, -1)
    max_pool = fused.max(dim=1)[0]                                    # :58
    avg_pool = fused.mean(dim=1)                                      # :59
    return torch.cat([max_pool, avg_pool], dim=1), fused              # :60-62


def shared_mlp3_forward(p, line_pm, prefix="point_mlp.", training=False, new_stats=None):
    """LineRefineNet.point_mlp, src/model.py:150-159,200-201: 3->64->128->256,
    BN after each conv, ReLU after the first two only.  line_pm (B,M,3) ->
    (B,M,256)."""
    B, M, C = line_pm.shape
    h = line_pm.reshape(B * M, C)
    for conv, bn, relu in (("0", "1", True), ("3", "4", True), ("6", "7", False)):
        z = h @ _w2d(p[f"{prefix}{conv}.weight"]).t() + p[f"{prefix}{conv}.bias"]
        h = batch_norm_rows(z, p, f"{prefix}{bn}", training, new_stats)
        if relu:
            h = torch.relu(h)
    return h.reshape(B, M, -1)


def pos_emb(p, xyz, prefix="pos_emb.mlp."):
    """PositionalEncoding.forward, src/model.py:64-75."""
    h = torch.relu(xyz @ p[prefix + "0.weight"].t() + p[prefix + "0.bias"])
    return h @ p[prefix + "2.weight"].t() + p[prefix + "2.bias"]


def layer_norm(x, p, pre):
    mu = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mu) * torch.rsqrt(var + LN_EPS) * p[pre + ".weight"] + p[pre + ".bias"]


def mha(p, pre, q_in, k_in, v_in):
    """nn.MultiheadAttention(256, 8, batch_first=True) forward with dropout off,
    src/model.py:84-85,113-114,126.  Packed in_proj split as [q;k;v]."""
    d = q_in.shape[-1]
    hd = d // NHEAD
    W, bias = p[pre + ".in_proj_weight"], p[pre + ".in_proj_bias"]
    q = q_in @ W[0:d].t() + bias[0:d]
    k = k_in @ W[d:2 * d].t() + bias[d:2 * d]
    v = v_in @ W[2 * d:3 * d].t() + bias[2 * d:3 * d]
    B, L, _ = q.shape
    S = k.shape[1]
    q = q.reshape(B, L, NHEAD, hd).transpose(1, 2)
    k = k.reshape(B, S, NHEAD, hd).transpose(1, 2)
    v = v.reshape(B, S, NHEAD, hd).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, L, d)
    return o @ p[pre + ".out_proj.weight"].t() + p[pre + ".out_proj.bias"]


def decoder_layer(p, pre, tgt, memory, query_pos, pos):
    """DetrTransformerDecoderLayer.forward with dropout off, src/model.py:103-135."""
    q = tgt + query_pos
    tgt = layer_norm(tgt + mha(p, pre + ".self_attn", q, q, tgt), p, pre + ".norm1")     # :113-117
    q = tgt + query_pos
    k = memory + pos
    tgt = layer_norm(tgt + mha(p, pre + ".cross_attn", q, k, memory), p, pre + ".norm2")  # :123-128
    ff = torch.relu(F.linear(tgt, p[pre + ".linear1.weight"], p[pre + ".linear1.bias"]))
    ff = F.linear(ff, p[pre + ".linear2.weight"], p[pre + ".linear2.bias"])
    return layer_norm(tgt + ff, p, pre + ".norm3")                                        # :131-133


def reg_head(p, pre, tgt):
    """reg_branches[i], src/model.py:172-179: Linear(256,128)-ReLU-Linear(128,3)."""
    h = torch.relu(tgt @ p[pre + ".0.weight"].t() + p[pre + ".0.bias"])
    return h @ p[pre + ".2.weight"].t() + p[pre + ".2.bias"]


def linerefine_forward(p, context, noisy_line, training=False, new_stats=None,
                       return_intermediates=False):
    """LineRefineNet.forward with dropout off, src/model.py:181-234.
    context (B,N,4), noisy_line (B,M,3) -> (6,B,M,3)."""
    _, fused = encoder_forward(p, context, "context_encoder.", training, new_stats)  # :192-193
    memory = fused @ p["context_proj.weight"].t() + p["context_proj.bias"]          # :194
    pos_mem = pos_emb(p, context[:, :, :3])                                          # :197
    tgt = shared_mlp3_forward(p, noisy_line, "point_mlp.", training, new_stats)     # :200-201
    cur = noisy_line.clone()                                                         # :204
    outs = []
    for i in range(6):                                                               # :209
        pos_tgt = pos_emb(p, cur)                                                    # :212
        tgt = decoder_layer(p, f"decoder_layers.{i}", tgt, memory, pos_tgt, pos_mem)  # :217
        cur = cur + reg_head(p, f"reg_branches.{i}", tgt)                            # :220,227
        outs.append(cur - noisy_line)                                                # :230-231
    out = torch.stack(outs)                                                          # :234
    if return_intermediates:
        return out, {"memory": memory, "fused": fused}
    return out


def deep_supervision_l1(pred_stack, target):
    """loss of train.py:63-68 / train_dist.py:180-186: mean over layers of L1Loss."""
    return sum((pred_stack[l] - target).abs().mean() for l in range(pred_stack.shape[0])) \
        / pred_stack.shape[0]


def as_params(sd, dtype=torch.float32, requires_grad=False):
    """state_dict -> dict of leaf tensors (floating entries cast to dtype)."""
    out = {}
    for k, v in sd.items():
        if v.is_floating_point():
            t = v.detach().to(dtype).clone()
            is_buf = k.endswith("running_mean") or k.endswith("running_var")
            t.requires_grad_(requires_grad and not is_buf)
            out[k] = t
        else:
            out[k] = v.clone()
    return out
