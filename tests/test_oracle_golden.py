"""The oracle (oracle/linerefine_oracle.py) against the golden vectors that
oracle/make_golden.py produced by running the REFERENCE (src/model.py) on CPU.
Runs without the reference and without a GPU.

Tolerances: forward outputs max-abs (fp32 noise floor of the reference itself is
~1e-6 on single layers and ~2e-5 through the 6 decoder layers with these weights);
gradients by relative L2 per parameter, because one ReLU pre-activation inside fp32
noise of zero flips a mask and moves isolated entries by O(1e-3) (see make_golden.py).
"""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import maxdiff, rel_l2
from oracle import linerefine_oracle as O
from oracle import procedural as P

HEAD = 64


def _pre_bn_bias(k):
    return bool(re.search(r"(conv\d\.bias|fusion\.0\.bias|point_mlp\.[036]\.bias)$", k))


def test_g6_manifest(golden_dir):
    man = json.load(open(os.path.join(golden_dir, "g6_state_dict_manifest.json")))
    ours = [[k, list(s), "int64" if kind == "nbt" else "float32"] for k, s, kind in P.linerefine_manifest()]
    assert len(ours) == 205 and ours == man["entries"]
    n = sum(int(np.prod(s)) for k, s, kind in P.linerefine_manifest()
            if kind not in ("nbt", "rmean", "rvar"))
    assert n == man["n_params"] == 9695954


def test_g1_eval_forward(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_eval_forward.npz"))
    sd = P.linerefine_state_dict(0)
    ctx, noisy, _ = P.synth_batch(8, 256, 4, 32, seed=1234)
    assert abs(float(ctx.double().sum()) - float(g["ctx_sum"])) < 1e-6
    assert abs(float(noisy.double().sum()) - float(g["noisy_sum"])) < 1e-6
    p = O.as_params(sd)
    with torch.no_grad():
        out, inter = O.linerefine_forward(p, ctx, noisy, return_intermediates=True)
        gfeat, fused = O.encoder_forward(p, ctx, "context_encoder.")
    assert out.shape == (6, 8, 32, 3)
    assert maxdiff(out, g["out"]) < 5e-5
    assert maxdiff(gfeat, g["global_feat"]) < 1e-5
    assert maxdiff(inter["memory"][:, ::16, ::8], g["memory_sub"]) < 1e-5
    assert maxdiff(fused[:, ::16, ::8], g["fused_sub"]) < 1e-5


def test_g2_train_fwd_bwd(golden_dir):
    g = np.load(os.path.join(golden_dir, "g2_train_fwd_bwd.npz"))
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(8, 256, 4, 32, seed=1234)
    p = O.as_params(sd, requires_grad=True)
    ctx.requires_grad_(True)
    noisy.requires_grad_(True)
    ns = {}
    out = O.linerefine_forward(p, ctx, noisy, training=True, new_stats=ns)
    loss = O.deep_supervision_l1(out, target)
    loss.backward()
    assert maxdiff(out, g["out"]) < 1e-4
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    assert rel_l2(g["dctx"], ctx.grad) < 5e-3
    assert rel_l2(g["dnoisy"], noisy.grad) < 5e-3
    rels = []
    for k, nrm in zip(g["grad_keys"], g["grad_norms"]):
        k = str(k)
        if _pre_bn_bias(k):
            continue
        gr = p[k].grad.reshape(-1).double()
        assert abs(float(gr.norm()) - nrm) <= 5e-3 * nrm + 1e-12, k
        rels.append(rel_l2(g["gh::" + k], gr[:HEAD]))
    assert max(rels) < 2e-2 and float(np.median(rels)) < 1e-3
    for k, v in ns.items():
        ref = torch.from_numpy(g["st::" + k])
        assert maxdiff(ref, v) <= 1e-5 * float(ref.double().abs().max()) + 1e-7, k


@pytest.mark.parametrize("name,C,B,N", [("g3_encoder_c4_train", 4, 4, 192), ("g4_encoder_c6_train", 6, 3, 160)])
def test_g3_g4_encoder(golden_dir, name, C, B, N):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    sd = P.encoder_state_dict(C, 1024, seed=3)
    sd["fusion.1.weight"][5] = 0.0
    sd["fusion.1.bias"][5] = -1.0
    ctx, _, _ = P.synth_batch(B, N, C, 32, seed=77)
    r = np.random.default_rng(5)
    up_g = torch.from_numpy(r.normal(0, 1, (B, 2048)).astype(np.float32))
    up_f = torch.from_numpy(r.normal(0, 1, (B, N, 1024)).astype(np.float32))
    for mode in ("train", "eval"):
        p = O.as_params(sd, requires_grad=True)
        x = ctx.clone().requires_grad_(True)
        ns = {}
        gf, fu = O.encoder_forward(p, x, "", mode == "train", ns)
        ((gf * up_g).sum() + (fu * up_f).sum()).backward()
        assert maxdiff(gf, g[f"{mode}::gfeat"]) < 2e-5
        assert maxdiff(fu[:, ::8, ::16], g[f"{mode}::fused_sub"]) < 2e-5
        assert rel_l2(g[f"{mode}::dx"], x.grad) < 5e-3
        for k, nrm in zip(g[f"{mode}::grad_keys"], g[f"{mode}::grad_norms"]):
            k = str(k)
            if mode == "train" and _pre_bn_bias(k):
                continue
            gr = p[k].grad.reshape(-1).double()
            assert abs(float(gr.norm()) - nrm) <= 5e-3 * nrm + 1e-9, (mode, k)
        if mode == "train":
            # dead channel 5 ties at every point: the whole max-pool gradient goes to
            # the FIRST point; gamma=0 so only beta sees it.
            for k, v in ns.items():
                ref = torch.from_numpy(g["st::" + k])
                assert maxdiff(ref, v) <= 1e-5 * float(ref.double().abs().max()) + 1e-7, k


def test_g5_point_mlp_c3(golden_dir):
    g = np.load(os.path.join(golden_dir, "g5_point_mlp_c3.npz"))
    sd = P.linerefine_state_dict(0)
    r = np.random.default_rng(9)
    x = torch.from_numpy(r.normal(0, 1.5, (4, 1024, 3)).astype(np.float32))
    p = O.as_params(sd)
    for mode in ("train", "eval"):
        with torch.no_grad():
            y = O.shared_mlp3_forward(p, x, "point_mlp.", mode == "train", {})
        assert maxdiff(y[:, ::16, ::2], g[mode]) < 2e-5
