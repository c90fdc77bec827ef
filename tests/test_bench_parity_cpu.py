"""Host logic of bench.py's parity gate (grade_parity): every tensor that receives a gradient is held to a
per-tensor bound - fp32-accurate modes at the benchmarked state, reduced precision at the initial weights plus
the Adam-update error at the trained state - and the raw figures are always reported.  No GPU."""
import copy

import bench


def _rec(per, out_abs=5e-6, out_rel=2e-6):
    return {"out_max_abs_vs_exact_fp32": out_abs, "out_rel_l2_vs_exact_fp32": out_rel, "worst_grad_rel_l2": 0.0,
            "worst_grad": "", "all_grads_rel_l2": 1e-6, "loss": 0.2, "loss_exact_fp32": 0.2,
            "per_tensor": copy.deepcopy(per), "fwd_bwd_ms": 1.0, "exact_fp32_fwd_bwd_ms": 2.0}


def test_fp32_accurate_modes_gate_every_live_tensor():
    per = [("a.weight", 1.0, 5e-4, None), ("b.weight", 1e-3, 1e-6, None), ("conv1.bias", 1e-12, 1e-12, None)]
    p = bench.grade_parity(_rec(per), 3)
    assert p["ok"] and p["grad_tensors_checked"] == 2 and p["grad_tensors_over_bound"] == []     # the zero-gradient bias is not "live"
    per[1] = ("b.weight", 1e-3, 5e-6, None)                                                      # 5e-3 rel-L2 on a small tensor
    p = bench.grade_parity(_rec(per), 3)
    assert not p["ok"] and p["grad_tensors_over_bound"][0][0] == "b.weight"
    p = bench.grade_parity(_rec(per[:1], out_abs=2e-4), 0)                                       # outputs beyond 1e-4
    assert not p["ok"]


def test_reduced_precision_gates_initial_state_and_adam_update():
    trained = [("enc.weight", 3e-4, 3.3e-4, 0.012), ("dec.weight", 1.0, 4e-3, 0.02)]             # enc: rel-L2 1.1, update error small
    init = [("enc.weight", 5e-2, 9e-4, None), ("dec.weight", 1.0, 5e-3, None)]
    p = bench.grade_parity(_rec(trained, out_rel=6e-3), 4, _rec(init, out_rel=4e-3))
    assert p["ok"]
    assert p["grad_tensors_over_bound"][0][0] == "enc.weight"            # reported, not hidden
    assert p["at_init"]["worst_grad_rel_l2"] < 0.1 and p["adam_update_err_rms_lr_worst"] == 0.02
    bad_init = [("enc.weight", 5e-2, 1e-2, None), ("dec.weight", 1.0, 5e-3, None)]               # 0.2 at the initial weights
    assert not bench.grade_parity(_rec(trained, out_rel=6e-3), 4, _rec(bad_init))["ok"]
    bad_upd = [("enc.weight", 3e-4, 3.3e-4, 0.3), ("dec.weight", 1.0, 4e-3, 0.02)]                # 0.3 learning rates of update error
    assert not bench.grade_parity(_rec(bad_upd, out_rel=6e-3), 4, _rec(init))["ok"]
    assert not bench.grade_parity(_rec(trained, out_rel=8e-2), 4, _rec(init))["ok"]              # outputs beyond 5e-2 rel-L2
    # without either extra record the trained-state rel-L2 itself is the gate
    assert not bench.grade_parity(_rec([(n, a, b, None) for n, a, b, _ in trained], out_rel=6e-3), 4)["ok"]
