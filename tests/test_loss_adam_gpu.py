"""Row f3: the fused deep-supervision L1 loss (+ gradient) and the flat-buffer Adam step against
torch's own nn.L1Loss / torch.optim.Adam - the operators the reference's training loops call
(train.py:40,63-72, train_dist.py:150,180-189)."""
import pytest
import torch

from conftest import maxdiff, rel_l2

pytestmark = pytest.mark.gpu


def test_l1_deep_supervision_matches_torch():
    from pointnet_refine_amd import ops
    torch.manual_seed(0)
    for shape in ((6, 5, 32, 3), (6, 2048, 32, 3), (1, 7, 1, 3)):
        pred = torch.randn(*shape, device="cuda", requires_grad=True)
        target = torch.randn(*shape[1:], device="cuda")
        with torch.no_grad():
            pred[0, 0, 0, 0] = target[0, 0, 0]                 # an exact zero difference: sign(0) = 0
        loss = ops.deep_supervision_l1(pred, target)
        (loss * 3.0).backward()
        p2 = pred.detach().clone().requires_grad_(True)
        crit = torch.nn.L1Loss()
        ref = sum(crit(p2[l], target) for l in range(shape[0])) / shape[0]      # train_dist.py:180-186
        (ref * 3.0).backward()
        assert abs(float(loss.detach()) - float(ref.detach())) <= 2e-6 * max(1.0, abs(float(ref.detach())))
        assert maxdiff(pred.grad, p2.grad) <= 1e-9 + 1e-6 * float(p2.grad.abs().max())
        assert float(pred.grad[0, 0, 0, 0]) == 0.0
    # micro-batched use: chunk losses with the full-batch denominator add up
    pred = torch.randn(6, 64, 32, 3, device="cuda")
    target = torch.randn(64, 32, 3, device="cuda")
    whole = ops.deep_supervision_l1(pred, target)
    parts = sum(ops.deep_supervision_l1(pred[:, s:s + 16].contiguous(), target[s:s + 16].contiguous(), pred.numel())
                for s in range(0, 64, 16))
    assert abs(float(whole) - float(parts)) < 1e-6
    with pytest.raises(RuntimeError):
        ops.deep_supervision_l1(pred.cpu(), target.cpu())
    # geometry metrics of train_dist.py:190-203 on the same pass, accumulated over chunks
    noisy = torch.randn(64, 32, 3, device="cuda")
    geo = torch.zeros(2, device="cuda")
    for s in range(0, 64, 16):
        ops.deep_supervision_l1(pred[:, s:s + 16].contiguous(), target[s:s + 16].contiguous(), pred.numel(), geo, 64 * 32)
    line_gt, line_pred = noisy + target, noisy + pred[-1]
    init_err = (noisy - line_gt).norm(dim=-1).mean()
    refine_err = (line_pred - line_gt).norm(dim=-1).mean()
    assert abs(float(geo[0]) - float(init_err)) < 2e-6 * float(init_err)
    assert abs(float(geo[1]) - float(refine_err)) < 2e-6 * float(refine_err)


def test_flat_adam_matches_torch_adam():
    from pointnet_refine_amd.train_step import FlatAdam, FlatGrads
    torch.manual_seed(1)
    shapes = [(64, 4, 1), (64,), (1024, 1984, 1), (3,), (256, 1024), (7, 5)]
    pa = [torch.nn.Parameter(torch.randn(*s, device="cuda") * 0.1) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    fg = FlatGrads(pa)
    opt_a = FlatAdam(fg, lr=1e-3)
    opt_b = torch.optim.Adam(pb, lr=1e-3)                     # the reference's optimiser
    for step in range(5):
        gs = [torch.randn_like(p) * (10.0 ** (step - 2)) for p in pa]
        opt_a.zero_grad()
        for p, q, g in zip(pa, pb, gs):
            p.grad.copy_(g)
            q.grad = g.clone()
        opt_a.step()
        opt_b.step()
        for p, q in zip(pa, pb):
            assert maxdiff(p, q) <= 2e-7 + 2e-6 * float(q.abs().max()), step
    assert all(p.data.data_ptr() >= opt_a.flat.data_ptr() for p in pa)        # parameters live in the flat buffer


def test_train_step_with_fused_loss_and_adam_matches_torch_path():
    """TrainStep(model) [fused L1 + FlatAdam] against TrainStep(model, torch Adam, torch loss): same
    weights after two steps of the full model (dropout off)."""
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    torch.manual_seed(3)
    a = LineRefineNet().cuda().train()
    b = LineRefineNet().cuda().train()
    b.load_state_dict(a.state_dict())
    for m in (a, b):
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0
    ctx, noisy, target = synthetic_batch(16, 256, torch.device("cuda", 0))
    torch_l1 = lambda out, tgt, denom: (out - tgt.unsqueeze(0)).abs().sum() / denom
    sa = TrainStep(a, decoder_chunk=8)
    sb = TrainStep(b, torch.optim.Adam(b.parameters(), lr=1e-3), decoder_chunk=8, loss_fn=torch_l1)
    la, lb = sa(ctx, noisy, target), sb(ctx, noisy, target)
    assert abs(float(la) - float(lb)) < 1e-5
    # Adam's first step moves every element by lr * sign(g): an element whose gradient sits at the
    # fp32 noise floor (analytically zero for the biases in front of a BatchNorm, which are
    # skipped) can take either sign on either path, so the check is on the share of elements that
    # agree, per tensor, plus equal losses on the second step
    import re
    worst = 0.0
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        if re.search(r"(conv\d\.bias|fusion\.0\.bias|point_mlp\.[036]\.bias)$", k):
            continue
        worst = max(worst, float(((p - q).abs() > 1e-4).float().mean()))
    assert worst < 0.03, worst
    la, lb = sa(ctx, noisy, target), sb(ctx, noisy, target)
    assert abs(float(la) - float(lb)) < 1e-4


def test_training_trajectory_split_fp16_vs_exact_fp32_cores():
    """Ten optimiser steps of the full model (dropout off) from the same start on the exact fp32
    MFMA cores, the split-bf16 cores and the default split-fp16 cores.  Adam's normalised steps
    amplify fp32-level noise, so fp32-accurate runs drift apart slowly (chaotically); a
    systematic error of a core family would show in the first steps already."""
    from pointnet_refine_amd import _lib
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    lib = _lib.lib()
    ctx, noisy, target = synthetic_batch(64, 1024, torch.device("cuda", 0))
    old = lib.prh_get_gemm_mode()
    curves = {}
    try:
        for mode in (0, 1, 3):
            lib.prh_set_gemm_mode(mode)
            torch.manual_seed(11)
            m = LineRefineNet().cuda().train()
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.0
            step = TrainStep(m, decoder_chunk=32)
            curves[mode] = [float(step(ctx, noisy, target)) for _ in range(10)]
    finally:
        lib.prh_set_gemm_mode(old)
    for mode, name in ((0, "exact fp32"), (1, "split-bf16"), (3, "split-fp16")):
        print(f"loss, {name} cores:", " ".join(f"{v:.5f}" for v in curves[mode]))
    a, b, c = curves[0], curves[1], curves[3]
    assert a[-1] < 0.9 * a[0]                                          # it trains
    assert abs(a[0] - c[0]) < 2e-5 * a[0] and abs(a[0] - b[0]) < 2e-5 * a[0]     # same first loss
    drift_bf16 = [abs(x - y) / x for x, y in zip(a, b)]
    drift_fp16 = [abs(x - y) / x for x, y in zip(a, c)]
    print("relative drift from the exact cores, split-bf16:", " ".join(f"{v:.1e}" for v in drift_bf16))
    print("relative drift from the exact cores, split-fp16:", " ".join(f"{v:.1e}" for v in drift_fp16))
    # the first steps agree tightly; later the trajectories separate chaotically (which of the two
    # split families ends up closer to the exact run changes with any reordering of fp32 sums),
    # but they stay on the same loss curve
    assert max(drift_fp16[:3]) < 5e-4 and max(drift_bf16[:3]) < 5e-4
    assert max(drift_fp16) < 5e-2 and max(drift_bf16) < 5e-2


def test_graph_captured_step_matches_eager_step():
    """TrainStep(graph=True): forward + loss + backward captured once, replayed per step.  With
    dropout off the replays must reproduce the eager steps; with dropout on, consecutive replays
    must draw different masks (device-side seed counter) yet stay finite and train."""
    from pointnet_refine_amd import _lib
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    ctx, noisy, target = synthetic_batch(8, 256, torch.device("cuda", 0))
    try:
        losses, buffers = {}, {}
        for graph in (False, True):
            torch.manual_seed(21)
            m = LineRefineNet().cuda().train()
            for mod in m.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.0
            step = TrainStep(m, decoder_chunk=4, graph=graph)
            losses[graph] = [float(step(ctx, noisy, target)) for _ in range(4)]
            buffers[graph] = {k: v.detach().clone() for k, v in m.named_buffers()}
        for a, b in zip(losses[False], losses[True]):
            assert abs(a - b) < 2e-4 * max(a, 1e-3), (losses[False], losses[True])
        # checkpoints of graph and eager runs agree: the capture's warm-up passes must not leak
        # into the BatchNorm running statistics or num_batches_tracked
        for k, v in buffers[False].items():
            w = buffers[True][k]
            if v.is_floating_point():
                assert float((v - w).abs().max()) <= 1e-3 * float(v.abs().max()) + 1e-5, k
            else:
                assert int(v) == int(w) == 4, (k, int(v), int(w))
        # dropout on: the same weights and inputs give different losses on consecutive replays
        torch.manual_seed(22)
        m = LineRefineNet().cuda().train()
        step = TrainStep(m, torch.optim.SGD(m.parameters(), lr=0.0), decoder_chunk=4, graph=True)
        l3 = [float(step(ctx, noisy, target)) for _ in range(3)]
        assert all(torch.isfinite(torch.tensor(l3))) and len({round(v, 7) for v in l3}) == 3, l3
    finally:
        _lib.lib().prh_set_dropout_seed_source(None)


@pytest.mark.parametrize("chunk", [None, 8])
def test_gradient_sinks_match_autograd_accumulation(chunk):
    """TrainStep hands the flat gradient buffer to the library's backward kernels (ops gradient
    sinks: the first contribution of a step overwrites a parameter's region, later ones are
    added, the Functions return None).  Same gradients as plain autograd accumulation into
    zeroed .grad tensors, with dropout on and the decoder micro-batched or not."""
    from pointnet_refine_amd import ops
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    torch.manual_seed(11)
    a = LineRefineNet().cuda().train()
    b = LineRefineNet().cuda().train()
    b.load_state_dict(a.state_dict())
    ctx, noisy, target = synthetic_batch(16, 256, torch.device("cuda", 0))
    sa = TrainStep(a, decoder_chunk=chunk)
    torch.manual_seed(5)                     # dropout seeds are drawn from torch's CPU generator
    sa.grads.zero()
    la = sa.forward_backward(ctx, noisy, target)
    assert ops.LAST_SINK_WRITES >= len(sa.grads.params) - 8, ops.LAST_SINK_WRITES   # nearly every parameter went direct
    assert not ops._SINKS_ACTIVE and not ops._SINK_WRITTEN       # the bookkeeping belongs to the scope
    # plain autograd on the twin: no TrainStep, no sinks; poisoned .grad would show an overwrite
    torch.manual_seed(5)
    sb = TrainStep(b, decoder_chunk=chunk)
    ops.clear_grad_sinks(sb.grads.params)    # registered by the constructor: take them away again
    sb.grads.zero()
    lb = sb.forward_backward(ctx, noisy, target)
    assert ops.LAST_SINK_WRITES == 0
    assert abs(float(la) - float(lb)) < 1e-6
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert p.grad is not None and q.grad is not None, k
        scale = float(q.grad.abs().max()) + 1e-12
        assert maxdiff(p.grad, q.grad) <= 2e-6 * scale + 1e-9, (k, maxdiff(p.grad, q.grad), scale)
    # a second step through the sinks starts from a zeroed buffer again: no carry-over
    torch.manual_seed(5)
    sa.grads.zero()
    sa.forward_backward(ctx, noisy, target)
    for (k, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        scale = float(q.grad.abs().max()) + 1e-12
        assert maxdiff(p.grad, q.grad) <= 2e-6 * scale + 1e-9, k
    sa.close()
    assert not any(id(p) in ops._GRAD_SINKS for p in sa.grads.params)


def test_gradient_accumulation_and_step_ownership():
    """(a) forward_backward(accumulate=True) ADDS to the flat buffer - two micro-batches
    then one optimiser step see the sum of both gradients, for sinked and autograd-handled parameters
    alike; (b) a second TrainStep on the same model takes the sinks over, and dropping the FIRST one
    afterwards must not remove the second one's registration (its gradients keep going direct)."""
    import gc
    from pointnet_refine_amd import ops
    from pointnet_refine_amd.model import LineRefineNet
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    dev = torch.device("cuda", 0)
    torch.manual_seed(2)
    m = LineRefineNet().cuda().train()
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, torch.nn.MultiheadAttention):
            mod.dropout = 0.0
    b1 = synthetic_batch(8, 256, dev, seed=1)
    b2 = synthetic_batch(8, 256, dev, seed=2)
    st = TrainStep(m, torch.optim.SGD(m.parameters(), lr=0.0), decoder_chunk=4)
    st.grads.zero(); st.forward_backward(*b1); g1 = st.grads.flat.clone()
    st.grads.zero(); st.forward_backward(*b2); g2 = st.grads.flat.clone()
    st.grads.zero()
    st.forward_backward(*b1)
    st.forward_backward(*b2, accumulate=True)
    want = g1 + g2
    assert float((st.grads.flat - want).abs().max()) <= 2e-6 * float(want.abs().max())
    st.forward_backward(*b2)                                   # default: the step starts over
    assert float((st.grads.flat - g2).abs().max()) <= 2e-6 * float(g2.abs().max())
    # (b) ownership
    st2 = TrainStep(m, torch.optim.SGD(m.parameters(), lr=0.0), decoder_chunk=4)
    del st
    gc.collect()
    assert all(id(p) in ops._GRAD_SINKS for p in st2.grads.params)
    st2.grads.zero()
    st2.forward_backward(*b1)
    assert ops.LAST_SINK_WRITES >= len(st2.grads.params) - 8
    assert float((st2.grads.flat - g1).abs().max()) <= 2e-6 * float(g1.abs().max())
    st2.close()
    assert not any(id(p) in ops._GRAD_SINKS for p in st2.grads.params)
    st2.grads.zero()
    st2.forward_backward(*b1)                                  # a closed step re-registers when used again
    assert ops.LAST_SINK_WRITES >= len(st2.grads.params) - 8
    st2.close()
