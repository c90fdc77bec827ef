"""Procedural weights and synthetic inputs shared by the oracle, the golden-vector
generator and the tests.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the product
package ``pointnet_refine_amd``; only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg use it.

Everything here is a closed-form function of (key name, shape, seed) built on
numpy's PCG64 stream, which is stable across numpy versions and platforms, so the
fixtures under ``tests/golden`` never have to store weights or inputs.

Shapes follow the reference's ``state_dict`` (``/root/reference/src/model.py:7-37``
encoder, ``:138-179`` LineRefineNet); the synthetic input distribution follows
SURVEY.md section 8(d).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

D_MODEL = 256
NUM_DECODER_LAYERS = 6
FFN_DIM = 1024


def _rng(key: str, seed: int = 0) -> np.random.Generator:
    return np.random.default_rng([zlib.crc32(key.encode()), seed])


def encoder_manifest(in_channel: int = 4, out_dim: int = 1024, prefix: str = ""):
    """(key, shape, kind) for MultiScalePointNetEncoder (src/model.py:7-37)."""
    m = []
    chans = [in_channel, 64, 128, 256, 512, out_dim]
    for k in range(1, 6):
        m.append((f"{prefix}conv{k}.weight", (chans[k], chans[k - 1], 1), "w"))
        m.append((f"{prefix}conv{k}.bias", (chans[k],), "b"))
    for k in range(1, 6):
        m += _bn_manifest(f"{prefix}bn{k}", chans[k])
    cat = 64 + 128 + 256 + 512 + out_dim
    m.append((f"{prefix}fusion.0.weight", (out_dim, cat, 1), "w"))
    m.append((f"{prefix}fusion.0.bias", (out_dim,), "b"))
    m += _bn_manifest(f"{prefix}fusion.1", out_dim)
    m.append((f"{prefix}intensity_gate.0.weight", (64, 1, 1), "w"))
    m.append((f"{prefix}intensity_gate.0.bias", (64,), "b"))
    m.append((f"{prefix}intensity_gate.2.weight", (out_dim, 64, 1), "w"))
    m.append((f"{prefix}intensity_gate.2.bias", (out_dim,), "b"))
    return m


def _bn_manifest(pre, c):
    return [
        (f"{pre}.weight", (c,), "gamma"),
        (f"{pre}.bias", (c,), "beta"),
        (f"{pre}.running_mean", (c,), "rmean"),
        (f"{pre}.running_var", (c,), "rvar"),
        (f"{pre}.num_batches_tracked", (), "nbt"),
    ]


def _lin(pre, out_f, in_f):
    return [(f"{pre}.weight", (out_f, in_f), "w"), (f"{pre}.bias", (out_f,), "b")]


def _ln(pre, c):
    return [(f"{pre}.weight", (c,), "gamma"), (f"{pre}.bias", (c,), "beta")]


def linerefine_manifest(feature_dim: int = 1024):
    """(key, shape, kind) for LineRefineNet in reference state_dict order
    (src/model.py:138-179); 205 entries for feature_dim=1024."""
    d = D_MODEL
    m = encoder_manifest(4, feature_dim, "context_encoder.")
    m += _lin("context_proj", d, feature_dim)
    m.append(("point_mlp.0.weight", (64, 3, 1), "w"))
    m.append(("point_mlp.0.bias", (64,), "b"))
    m += _bn_manifest("point_mlp.1", 64)
    m.append(("point_mlp.3.weight", (128, 64, 1), "w"))
    m.append(("point_mlp.3.bias", (128,), "b"))
    m += _bn_manifest("point_mlp.4", 128)
    m.append(("point_mlp.6.weight", (d, 128, 1), "w"))
    m.append(("point_mlp.6.bias", (d,), "b"))
    m += _bn_manifest("point_mlp.7", d)
    m += _lin("pos_emb.mlp.0", d, 3)
    m += _lin("pos_emb.mlp.2", d, d)
    for i in range(NUM_DECODER_LAYERS):
        p = f"decoder_layers.{i}"
        for att in ("self_attn", "cross_attn"):
            m.append((f"{p}.{att}.in_proj_weight", (3 * d, d), "w"))
            m.append((f"{p}.{att}.in_proj_bias", (3 * d,), "b"))
            m += _lin(f"{p}.{att}.out_proj", d, d)
        m += _lin(f"{p}.linear1", FFN_DIM, d)
        m += _lin(f"{p}.linear2", d, FFN_DIM)
        for n in ("norm1", "norm2", "norm3"):
            m += _ln(f"{p}.{n}", d)
    for i in range(NUM_DECODER_LAYERS):
        m += _lin(f"reg_branches.{i}.0", 128, d)
        m += _lin(f"reg_branches.{i}.2", 3, 128)
    return m


def make_tensor(key: str, shape, kind: str, seed: int = 0) -> torch.Tensor:
    """Closed-form tensor for one state_dict entry."""
    r = _rng(key, seed)
    if kind == "nbt":
        return torch.tensor(int(r.integers(0, 50)), dtype=torch.int64)
    if kind == "w":
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = r.uniform(-1.0, 1.0, size=shape) * (1.7 / np.sqrt(max(fan_in, 1)))
        # the raw-intensity channel of the first conv / gate sees values up to 255
        return torch.from_numpy(a.astype(np.float32))
    if kind == "b":
        return torch.from_numpy(r.uniform(-0.2, 0.2, size=shape).astype(np.float32))
    if kind == "gamma":
        return torch.from_numpy(r.uniform(0.5, 1.5, size=shape).astype(np.float32))
    if kind == "beta":
        return torch.from_numpy(r.uniform(-0.3, 0.3, size=shape).astype(np.float32))
    if kind == "rmean":
        return torch.from_numpy(r.uniform(-0.1, 0.1, size=shape).astype(np.float32))
    if kind == "rvar":
        return torch.from_numpy(r.uniform(0.5, 1.5, size=shape).astype(np.float32))
    raise ValueError(kind)


def make_state_dict(manifest, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    return OrderedDict((k, make_tensor(k, s, kind, seed)) for k, s, kind in manifest)


def tame_first_layers(sd, prefix="context_encoder."):
    """Scale the weights that multiply raw intensity (0..255) so that eval-mode
    running statistics (O(1)) give non-saturated activations, as a trained
    checkpoint would.  In-place; returns sd."""
    for k in (f"{prefix}conv1.weight",):
        if k in sd and sd[k].shape[1] >= 4:
            sd[k][:, 3] *= 0.02
            sd[k][:, 0] *= 0.05
    k = f"{prefix}intensity_gate.0.weight"
    if k in sd:
        sd[k] *= 0.05
    return sd


def linerefine_state_dict(seed: int = 0, feature_dim: int = 1024):
    return tame_first_layers(make_state_dict(linerefine_manifest(feature_dim), seed))


def encoder_state_dict(in_channel=4, out_dim=1024, seed: int = 0):
    return tame_first_layers(make_state_dict(encoder_manifest(in_channel, out_dim), seed), "")


# --------------------------------------------------------------------------------------
# synthetic inputs, SURVEY.md section 8(d)
# --------------------------------------------------------------------------------------

def synth_batch(B: int, N: int, C: int = 4, M: int = 32, seed: int = 1234):
    """context (B,N,C), noisy_line (B,M,3), target_offset (B,M,3) as float32 torch CPU
    tensors.  x ~ U(-25,25), y ~ N(0,.5), z ~ N(0,.1), centred per segment
    (src/dataset.py:232-233); channel 3 = raw intensity clamp(round(Exp(12)),0,255)
    (src/dataset.py:65-68,234); channels >=4 (C=6 encoder variant) ~ N(0,1) normals.
    noisy_line: 32 points along x with yaw<=2deg, lateral shift U(+-.4), jitter N(0,.05)
    (tools/augment_train_data.py:23-48,71-75), centred on its centroid."""
    r = np.random.default_rng([seed, B, N, C, M])
    ctx = np.empty((B, N, C), np.float32)
    ctx[..., 0] = r.uniform(-25, 25, (B, N))
    ctx[..., 1] = r.normal(0, 0.5, (B, N))
    ctx[..., 2] = r.normal(0, 0.1, (B, N))
    ctx[..., :3] -= ctx[..., :3].mean(axis=1, keepdims=True)
    if C > 3:
        ctx[..., 3] = np.clip(np.round(r.exponential(12.0, (B, N))), 0, 255)
    if C > 4:
        ctx[..., 4:] = r.normal(0, 1, (B, N, C - 4))
    xs = np.linspace(-25, 25, M, dtype=np.float64)
    gt = np.zeros((B, M, 3))
    gt[..., 0] = xs
    gt[..., 1] = r.normal(0, 0.05, (B, 1)) + 0.002 * xs * r.normal(0, 1, (B, 1))
    yaw = np.deg2rad(r.uniform(-2, 2, (B, 1)))
    noisy = gt.copy()
    noisy[..., 0] = gt[..., 0] * np.cos(yaw) - gt[..., 1] * np.sin(yaw)
    noisy[..., 1] = gt[..., 0] * np.sin(yaw) + gt[..., 1] * np.cos(yaw)
    noisy[..., 0] += r.uniform(-0.4, 0.4, (B, 1))
    noisy[..., 1] += r.uniform(-0.4, 0.4, (B, 1))
    noisy[..., 2] += r.uniform(-0.1, 0.1, (B, 1))
    noisy += r.normal(0, 0.05, (B, M, 3))
    cen = noisy.mean(axis=1, keepdims=True)
    target = (gt - noisy).astype(np.float32)
    noisy = (noisy - cen).astype(np.float32)
    return torch.from_numpy(ctx), torch.from_numpy(noisy), torch.from_numpy(target)
