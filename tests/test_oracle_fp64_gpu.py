"""Per-tensor gradient parity against an fp64 run of the oracle.

The golden G2 fixture stores fp32 gradients of the reference (norms + 64-entry heads), and two valid
fp32 evaluations differ by isolated ReLU / arg-max flips - which is why its gates are 5e-3 / 2e-2.
Against the oracle run in fp64 there is no such noise on the reference side, so every tensor that
receives a gradient is held to 2e-3 rel-L2 here:

* on the G2 inputs (B=8, N=256), oracle in fp64 on the host (measured worst tensor 1.1e-3, point_mlp.0.weight);
* at BASELINE config 2's size (B=512, N=1024, train mode): `memory`, the batch statistics / running
  statistics of all nine BatchNorm layers, `out`, the loss and every parameter gradient.  At that size
  the oracle's tensor math (oracle/linerefine_oracle.py, unchanged) is executed in fp64 by stock PyTorch
  on the device - rocBLAS / eager kernels, none of this repo's - because 13 TFLOP of fp64 do not fit a
  CPU test; it is still the restatement checking the HIP path, not the other way round (measured worst
  tensor 4.8e-4 at config 2's size, 2.6e-4 at the N=2048 shape)."""
import re

import pytest
import torch

from conftest import maxdiff, rel_l2
from oracle import linerefine_oracle as O
from oracle import procedural as P

pytestmark = pytest.mark.gpu

GRAD_GATE = 2e-3


def _pre_bn_bias(k):      # the true gradient of a bias in front of a train-mode BatchNorm is exactly zero
    return bool(re.search(r"(conv\d\.bias|fusion\.0\.bias|point_mlp\.[036]\.bias)$", k))


def _hip_model(sd):
    from pointnet_refine_amd.model import LineRefineNet
    m = LineRefineNet()
    m.load_state_dict(sd, strict=True)
    m = m.cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    return m


def _oracle_fp64(sd, ctx, noisy, target, device):
    sd64 = {k: v.to(device) for k, v in sd.items()}
    p = O.as_params(sd64, dtype=torch.float64, requires_grad=True)
    stats = {}
    out, inter = O.linerefine_forward(p, ctx.to(device).double(), noisy.to(device).double(), training=True,
                                      new_stats=stats, return_intermediates=True)
    loss = O.deep_supervision_l1(out, target.to(device).double())
    loss.backward()
    return p, out.detach(), float(loss), inter["memory"].detach(), stats


def _check(m, hip_out, hip_loss, p, out64, loss64, stats):
    assert maxdiff(hip_out, out64) < 1e-4
    assert abs(hip_loss - loss64) < 1e-5
    worst, worst_k = 0.0, ""
    for k, v in m.named_parameters():
        if _pre_bn_bias(k):
            continue
        r = rel_l2(p[k].grad, v.grad)
        if r > worst:
            worst, worst_k = r, k
    print(f"worst per-tensor gradient rel-L2 vs the fp64 oracle: {worst:.3e} ({worst_k})")
    assert worst < GRAD_GATE, (worst_k, worst)
    msd = m.state_dict()
    for k, ref in stats.items():
        if "running" in k:
            assert maxdiff(msd[k], ref) <= 1e-5 * float(ref.abs().max()) + 1e-6, k
        else:
            assert int(msd[k]) == int(ref), k


def test_g2_inputs_every_gradient_vs_fp64_oracle():
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(8, 256, 4, 32, seed=1234)
    m = _hip_model(sd)
    out = m(ctx.cuda(), noisy.cuda())
    from pointnet_refine_amd import ops
    loss = ops.deep_supervision_l1(out, target.cuda())
    loss.backward()
    p, out64, loss64, _, stats = _oracle_fp64(sd, ctx, noisy, target, "cpu")
    _check(m, out.detach(), float(loss), p, out64, loss64, stats)


def test_config2_train_step_vs_fp64_oracle_at_size():
    """B=512, N=1024, C=4 train mode (BASELINE config 2) against the oracle in fp64."""
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(512, 1024, 4, 32, seed=2)
    m = _hip_model(sd)
    cg, ng = ctx.cuda(), noisy.cuda()
    memory = m.encode_context(cg)
    out = m.decode(cg, ng, memory, m.encode_line(ng))
    from pointnet_refine_amd import ops
    loss = ops.deep_supervision_l1(out, target.cuda())
    loss.backward()
    torch.cuda.synchronize()
    p, out64, loss64, mem64, stats = _oracle_fp64(sd, ctx, noisy, target, "cuda")
    assert maxdiff(memory.detach().reshape(mem64.shape), mem64) < 1e-4
    _check(m, out.detach(), float(loss), p, out64, loss64, stats)


def test_config4_rank_share_points_2048_vs_fp64_oracle():
    """BASELINE config 4's per-rank shape is B=512, N=2048; the same number of points as config 2 fits this
    check as B=256, N=2048 (N = 2048 keys per cross-attention, 64 key tiles per head): train-mode step
    against the oracle in fp64, every gradient tensor within 2e-3."""
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(256, 2048, 4, 32, seed=4)
    m = _hip_model(sd)
    cg, ng = ctx.cuda(), noisy.cuda()
    memory = m.encode_context(cg)
    out = m.decode(cg, ng, memory, m.encode_line(ng))
    from pointnet_refine_amd import ops
    loss = ops.deep_supervision_l1(out, target.cuda())
    loss.backward()
    torch.cuda.synchronize()
    p, out64, loss64, mem64, stats = _oracle_fp64(sd, ctx, noisy, target, "cuda")
    assert maxdiff(memory.detach().reshape(mem64.shape), mem64) < 1e-4
    _check(m, out.detach(), float(loss), p, out64, loss64, stats)
