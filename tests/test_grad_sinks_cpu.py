"""Host logic of the gradient sinks (pointnet_refine_amd/ops.py): how a training loop's flat gradient
buffer is resolved for parameters and for row blocks of packed parameters, first-contribution /
later-contribution bookkeeping, and the K / V row concatenation whose backward scatters straight
into the sinks.  Pure torch on CPU tensors - no kernel is launched (the kernels that write through
the sinks are covered by tests/test_loss_adam_gpu.py::test_gradient_sinks_match_autograd_accumulation)."""
import torch

from pointnet_refine_amd import ops


def _params():
    torch.manual_seed(0)
    w = torch.nn.Parameter(torch.randn(6, 4))          # a packed [3 x 2 rows, 4] parameter
    b = torch.nn.Parameter(torch.randn(6))
    flat = torch.zeros(w.numel() + b.numel())
    gw, gb = flat[:24].view(6, 4), flat[24:]
    return w, b, flat, gw, gb


def test_sink_view_resolves_parameters_and_row_blocks_only_while_active():
    w, b, flat, gw, gb = _params()
    ops.register_grad_sinks([(w, gw), (b, gb)])
    try:
        assert ops._sink_view(w) is None                     # not inside a step
        with ops.sinks_active():
            assert ops._sink_view(w).data_ptr() == gw.data_ptr()
            blk = w[2:4]                                     # rows 2..3 of the packed parameter
            sv = ops._sink_view(blk)
            assert sv.shape == blk.shape and sv.data_ptr() == gw[2:4].data_ptr()
            assert ops._sink_view(b[4:]).data_ptr() == gb[4:].data_ptr()
            assert ops._sink_view(w.t()) is None             # not a contiguous block: autograd handles it
            assert ops._sink_view(torch.nn.Parameter(torch.zeros(2))) is None     # not registered
            with torch.no_grad():
                frozen = w.detach()
            assert ops._sink_view(frozen) is None            # nothing to accumulate for it
        assert ops._sink_view(w) is None
    finally:
        ops.clear_grad_sinks([w, b])
    assert not ops._GRAD_SINKS


def test_first_contribution_overwrites_later_ones_add():
    w, b, flat, gw, gb = _params()
    ops.register_grad_sinks([(w, gw), (b, gb)])
    try:
        with ops.sinks_active():
            sink = ops._sink_view(b)
            buf, tok = ops._grad_buf(sink, (6,), b.device)
            assert buf.data_ptr() == gb.data_ptr() and tok is ops._DIRECT     # the kernel would write the sink itself
            buf.copy_(torch.arange(6.0))
            assert ops._grad_ret(buf, tok) is None
            buf2, tok2 = ops._grad_buf(sink, (6,), b.device)                  # second contribution of the step
            assert buf2.data_ptr() != gb.data_ptr() and tok2.data_ptr() == gb.data_ptr()
            buf2.fill_(1.0)
            assert ops._grad_ret(buf2, tok2) is None
            assert torch.equal(gb, torch.arange(6.0) + 1.0)
            # a shape that does not match the sink falls back to a plain tensor handed to autograd
            buf3, tok3 = ops._grad_buf(sink, (5,), b.device)
            assert tok3 is None and ops._grad_ret(buf3, tok3) is buf3
        with ops.sinks_active(new_step=True):                                  # next step: overwrite again
            buf, tok = ops._grad_buf(ops._sink_view(b), (6,), b.device)
            assert tok is ops._DIRECT
        # outside a step nothing is redirected
        buf, tok = ops._grad_buf(None, (6,), b.device)
        assert tok is None
    finally:
        ops.clear_grad_sinks([w, b])


def test_cat_rows_scatters_into_sinks_and_matches_autograd_without_them():
    w, b, flat, gw, gb = _params()
    w2 = torch.nn.Parameter(torch.randn(6, 4))
    g2 = torch.zeros(6, 4)
    up = torch.randn(4, 4)
    # reference: plain autograd
    ref = torch.cat([w[2:4], w2[2:4]])
    (ref * up).sum().backward()
    want_w, want_w2 = w.grad.clone(), w2.grad.clone()
    w.grad = None
    w2.grad = None
    ops.register_grad_sinks([(w, gw), (w2, g2)])
    try:
        with ops.sinks_active():
            out = ops.cat_rows([w[2:4], w2[2:4]])
            assert torch.equal(out, ref.detach())
            (out * up).sum().backward()
            # a second use in the same step is ADDED
            out = ops.cat_rows([w[2:4], w2[2:4]])
            (out * up).sum().backward()
        assert w.grad is None and w2.grad is None            # nothing went through autograd's accumulation
        assert torch.allclose(gw, 2 * want_w) and torch.allclose(g2, 2 * want_w2)
        assert float(gw[:2].abs().sum()) == 0.0 and float(gw[4:].abs().sum()) == 0.0
    finally:
        ops.clear_grad_sinks([w, w2])
    # without sinks the Function is a plain concatenation for autograd
    out = ops.cat_rows([w[2:4], w2[2:4]])
    (out * up).sum().backward()
    assert torch.allclose(w.grad, want_w) and torch.allclose(w2.grad, want_w2)
