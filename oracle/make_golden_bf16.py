"""What is the REFERENCE's own error when it runs in bf16?  Runs only in the build container:
imports the reference's src/model.py, loads the procedural weights, and compares its fp32
results with the same module under torch.autocast(bfloat16) - the standard way to train the
reference "in bf16" (BASELINE config 3) - on the G1 (eval) and G2 (train-mode, dropout off)
inputs.  The measured errors are the yardstick for this repo's bf16 mode (tests/test_bf16_gpu.py):
SURVEY 8(d)'s 5e-2 figure came from one such measurement in eval mode.  Writes
tests/golden/g9_bf16_autocast.npz (error figures + the autocast outputs; data only).
Usage: PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden_bf16.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import procedural as P  # noqa: E402
from oracle.make_golden import import_reference, zero_dropout, maxdiff, rel_l2, is_pre_bn_bias  # noqa: E402


def main():
    LineRefineNet, Encoder = import_reference()
    out = {}
    # ---- encoder alone, G3 set-up (tests/test_encoder_gpu.py): both returns, random upstream gradients
    B, N, Cc = 4, 192, 4
    esd = P.encoder_state_dict(Cc, 1024, seed=3)
    esd["fusion.1.weight"][5] = 0.0
    esd["fusion.1.bias"][5] = -1.0
    ectx, _, _ = P.synth_batch(B, N, Cc, 32, seed=77)
    r = np.random.default_rng(5)
    up_g = torch.from_numpy(r.normal(0, 1, (B, 2048)).astype(np.float32))
    up_f = torch.from_numpy(r.normal(0, 1, (B, N, 1024)).astype(np.float32))

    def enc(train, autocast):
        m = Encoder(in_channel=Cc, out_dim=1024)
        m.load_state_dict(esd, strict=True)
        m.train(train)
        x = ectx.clone().requires_grad_(True)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            gf, fu = m(x.transpose(2, 1))
        ((gf.float() * up_g).sum() + (fu.float().transpose(2, 1) * up_f).sum()).backward()
        return gf.float().detach(), fu.float().detach(), x.grad.detach(), {k: p.grad.detach().float() for k, p in m.named_parameters()}

    for train in (False, True):
        a, b = enc(train, False), enc(train, True)
        tag = "enc_train" if train else "enc_eval"
        heads = {k: rel_l2(a[3][k].reshape(-1)[:64], b[3][k].reshape(-1)[:64]) for k in a[3]
                 if not (train and is_pre_bn_bias(k)) and float(a[3][k].norm()) > 0}
        out[tag + "_fused_rel_l2"] = rel_l2(a[1], b[1])
        out[tag + "_gfeat_rel_l2"] = rel_l2(a[0], b[0])
        out[tag + "_dx_rel_l2"] = rel_l2(a[2], b[2])
        out[tag + "_grad_head_rel_l2_median"] = float(np.median(list(heads.values())))
        out[tag + "_grad_head_rel_l2_worst"] = max(heads.values())
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(8, 256, 4, 32, seed=1234)

    def run(train, autocast):
        m = LineRefineNet()
        m.load_state_dict(sd, strict=True)
        zero_dropout(m)
        m.train(train)
        nl = noisy.clone().requires_grad_(train)
        with torch.autocast("cpu", dtype=torch.bfloat16, enabled=autocast):
            o = m(ctx, nl)
            if train:
                loss = sum(torch.nn.functional.l1_loss(o[l].float(), target) for l in range(6)) / 6
        if not train:
            return o.float().detach(), None, None
        loss.backward()
        grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}
        return o.float().detach(), float(loss), grads

    with torch.no_grad():
        e32, _, _ = run(False, False)
        e16, _, _ = run(False, True)
    t32, l32, g32 = run(True, False)
    t16, l16, g16 = run(True, True)
    rels = {k: rel_l2(g32[k], g16[k]) for k in g32 if not is_pre_bn_bias(k) and float(g32[k].norm()) > 0}
    worst = max(rels, key=rels.get)
    # the same figure on the first 64 entries of each tensor - what the G2 fixture keeps of the
    # reference's gradients, hence what a test of another bf16 implementation can compare with
    heads = {k: rel_l2(g32[k].reshape(-1)[:64], g16[k].reshape(-1)[:64]) for k in rels}
    hworst = max(heads, key=heads.get)
    out["train_grad_head_rel_l2_median"] = float(np.median(list(heads.values())))
    out["train_grad_head_rel_l2_worst"] = heads[hworst]
    out["train_grad_head_rel_l2_p90"] = float(np.quantile(list(heads.values()), 0.9))
    out["eval_out_maxabs"] = maxdiff(e32, e16)
    out["eval_out_rel_l2"] = rel_l2(e32, e16)
    out["train_out_maxabs"] = maxdiff(t32, t16)
    out["train_out_rel_l2"] = rel_l2(t32, t16)
    out["train_loss_rel"] = abs(l16 - l32) / l32
    out["train_grad_rel_l2_median"] = float(np.median(list(rels.values())))
    out["train_grad_rel_l2_worst"] = rels[worst]
    for k, v in out.items():
        print(f"{k:28s} {v:.4e}")
    print("worst gradient tensor:", worst, "| on the 64-entry heads:", hworst)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g9_bf16_autocast.npz"),
                        eval_out=e16.numpy(), train_out=t16.numpy(), worst_grad=np.array(worst),
                        **{k: np.float64(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
