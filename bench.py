#!/usr/bin/env python
"""Benchmark of the LineRefineNet hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no RANK in the environment this process only LAUNCHES: it starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
bench.py <same flags>` as a child (run_dist_train.sh:16 of the reference does the same with
torchrun) and never touches the GPU itself; under an external torchrun (RANK set) it is a
rank.  One process per GPU, RCCL ("nccl" backend) over xGMI.

One STEP = one training pass over one synthetic batch per GPU, exactly what
train_dist.py:173-189 does per iteration: buffer broadcast, zero_grad, forward,
deep-supervision L1 loss, backward, gradient all-reduce (mean), Adam step.  Inputs are
generated on the device before the timed region.  Weak scaling: every rank processes --batch
segments.

Rank 0 prints ONE JSON line (contract in the task brief) with these extra objects:
  roofline      the dominant kernel of the timed region, timed live with HIP events on its
                launch stream (library profiler), against the peak of the MFMA products it
                issues; roofline.step = whole-step algorithmic FLOPs over the step time
  parity        one more step at the benchmark size, on the same inputs, weights and dropout
                seeds, with the exact-fp32 MFMA cores (PRH_GEMM=fp32) next to one with the
                benchmarked cores: max |out| difference and worst per-tensor gradient rel-L2
  cpu_baseline  the oracle (CPU restatement of the reference, kind "port") timed on the host
                cores on a bounded sample of the same workload (rank 0, N=1 only): full
                fwd+bwd (= value), encoder-only fwd+bwd, eval forward
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense v_mfma_f32_{32x32x16,16x16x32}_{bf16,f16}
HBM_PEAK_GBS = 8000.0
GEMM_MODES = {"fp32": 0, "split": 1, "bf16op": 2, "split16": 3, "bf16": 4}
DEFAULT_GEMM = "split16"
PMC_TRAFFIC_FILES = {3: "r03m_pmc_traffic_B4096_split16.json", 4: "r03m_pmc_traffic_B4096_bf16.json"}   # gemm mode -> committed PMC pass
# MFMA products issued per algorithmic MAC and what the peak is quoted on, per gemm mode
MODE_INFO = {
    0: (1, FP32_MFMA_PEAK_TFLOPS, "157.3 TF fp32 MFMA (v_mfma_f32_32x32x2_f32)",
        "f32"),
    1: (6, BF16_MFMA_PEAK_TFLOPS, "2500 TF dense bf16 MFMA / 6 products per fp32-accurate MAC (3-plane bf16 split)",
        "f32 (large GEMMs: 3xbf16-split MFMA, fp32 accumulate)"),
    2: (1, BF16_MFMA_PEAK_TFLOPS, "2500 TF dense bf16 MFMA",
        "bf16 MFMA operands, fp32 accumulate and storage (reduced precision)"),
    3: (3, BF16_MFMA_PEAK_TFLOPS, "2500 TF dense fp16 MFMA / 3 products per fp32-accurate MAC (2 scaled fp16 planes)",
        "f32 (large GEMMs: 2xfp16-split MFMA, 3 products, fp32 accumulate)"),
    4: (1, BF16_MFMA_PEAK_TFLOPS, "2500 TF dense bf16 MFMA (v_mfma_f32_16x16x32_bf16)",
        "bf16 (encoder activations and their gradients stored in bf16, bf16 MFMA operands in every GEMM; "
        "fp32 accumulate, statistics, master weights and optimiser; attention core on split-fp16 products)"),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=4096, help="segments per GPU per step")
    ap.add_argument("--points", type=int, default=1024, help="context points per segment")
    ap.add_argument("--decoder-chunk", type=int, default=4096,
                    help="segments per decoder micro-batch (bounds the decoder's activation memory; results are "
                         "identical to the unchunked step).  4096 = one pass at the headline batch: 208 GB peak, "
                         "2 %% faster than two micro-batches of 2048 (176 GB)")
    ap.add_argument("--gemm", choices=sorted(GEMM_MODES), default=DEFAULT_GEMM,
                    help="GEMM cores for the large GEMMs. split16 = two scaled fp16 planes, 3 MFMA "
                         "products, fp32-level error (default); split = three bf16 planes, 6 products; "
                         "fp32 = exact fp32 MFMA everywhere; bf16 = BASELINE config 3: bf16 operands AND "
                         "bf16 activation storage; bf16op = bf16 operands on fp32 storage (round-1 toggle)")
    ap.add_argument("--graph", action="store_true",
                    help="capture forward+loss+backward in a HIP graph (small, launch-bound batches)")
    ap.add_argument("--torch-adam", action="store_true", help="torch.optim.Adam instead of the fused flat-buffer Adam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the exact-fp32 parity step after the timed region")
    ap.add_argument("--no-workloads", action="store_true",
                    help="skip the `workloads` object (the other BASELINE.json configs, run after the timed region)")
    ap.add_argument("--kernels", type=int, default=0, help="print the K longest GEMM launches (live HIP-event times) to stderr")
    ap.add_argument("--cpu-batch", type=int, default=16,
                    help="segments in the CPU sample (16: the host's best throughput; 64 measured 2.5x slower per segment)")
    ap.add_argument("--encoder-only", action="store_true",
                    help="second workload (not the headline metric): MultiScalePointNetEncoder.forward returning "
                         "(global_feat, fused) - the north_star's fused shared-MLP + max-pool - alone: train-mode "
                         "forward+backward, or with --eval the inference forward through the single fused kernel")
    ap.add_argument("--eval", action="store_true", help="with --encoder-only: eval-mode forward only (fused kernel)")
    ap.add_argument("--stub", action="store_true",
                    help="launcher self-test: a small CPU stand-in model over the gloo backend instead of "
                         "the HIP model over RCCL (tests/test_bench_launcher_cpu.py); not a measurement")
    return ap.parse_args()


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """N > 1 without a launcher: become one.  Nothing here touches the GPU (not even
    torch.cuda.is_available()); the ranks are fresh child processes."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    log("launching", " ".join(cmd))
    return subprocess.call(cmd, env=env)


def host_cores():
    """CPU share of this process: cgroup quota if set, else affinity, capped at 16 (the
    GPU box hands one GPU 16 cores; os.cpu_count() reports the whole host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def step_flops(batch, points, line_points=32):
    """Algorithmic FLOPs of one training step per GPU (SURVEY.md 8(d) / BASELINE.md section 3):
    forward 2*MAC = 8,013,952*N + 12,482,432*M per segment, forward+backward = 3x."""
    return 3.0 * batch * (8_013_952.0 * points + 12_482_432.0 * line_points)


def cpu_baseline(points, batch):
    """Oracle (port of the reference) on the host cores, bounded sample: the three figures of
    BASELINE.md section 4 - (a) full forward+backward, (b) encoder-only forward+backward,
    (c) eval forward - 1 warm-up + best of 3 each."""
    import torch
    from oracle import linerefine_oracle as O
    from oracle import procedural as P
    torch.set_num_threads(host_cores())
    nt = torch.get_num_threads()
    sd = P.linerefine_state_dict(0)
    ctx, noisy, target = P.synth_batch(batch, points, 4, 32, seed=1234)

    def full():
        p = O.as_params(sd, requires_grad=True)
        out = O.linerefine_forward(p, ctx, noisy, training=True, new_stats={})
        O.deep_supervision_l1(out, target).backward()

    def encoder():
        p = O.as_params(sd, requires_grad=True)
        g, f = O.encoder_forward(p, ctx, "context_encoder.", True, {})
        (f.square().mean() + g.square().mean()).backward()

    def evalf():
        p = O.as_params(sd, requires_grad=False)
        with torch.no_grad():
            O.linerefine_forward(p, ctx, noisy, training=False)

    res = {}
    for name, fn in (("full_fwd_bwd", full), ("encoder_fwd_bwd", encoder), ("eval_forward", evalf)):
        best = float("inf")
        for it in range(4):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            log(f"cpu_baseline {name} iter {it}: {dt:.2f} s on {nt} threads")
            if it > 0:
                best = min(best, dt)
        res[name] = round(batch / best, 3)
    return {"value": res["full_fwd_bwd"], "unit": "segments/s", "cores": nt, "kind": "port",
            "encoder_fwd_bwd": res["encoder_fwd_bwd"], "eval_forward": res["eval_forward"],
            "sample": f"oracle/linerefine_oracle.py, B={batch}, N={points}, fp32, torch-CPU {nt} threads, "
                      f"1 warm-up + best of 3 each: value = full forward+backward (no optimizer); "
                      f"encoder_fwd_bwd = MultiScalePointNetEncoder alone; eval_forward = no_grad forward"}


def kernel_sources_sha():
    """Fingerprint of EVERY file of csrc/ (the file set _lib._SOURCES compiles); scripts/pmc_traffic.py
    records the same value next to the counters."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pointnet_refine_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hpp", ".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(dname, batch, points, mode):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (FETCH_SIZE x2
    + WRITE_SIZE, separate rocprofv3 --pmc runs of this same command, scripts/pmc_traffic.sh).
    Counters cannot be read from inside the benchmark, so this is a lookup valid for the
    configuration it was taken on (that GEMM mode, B=4096, N=1024); null otherwise."""
    import re
    fn = PMC_TRAFFIC_FILES.get(mode)
    if fn is None or not (batch == 4096 and points == 1024):
        return None, None
    path = os.path.join(ROOT, "profiles", fn)
    if not os.path.exists(path):
        return None, None
    recs = json.load(open(path))
    # the counters describe the kernels as they were when the passes ran: a file taken before the
    # last edit of the kernel sources is not used (traffic null, the file name says "stale")
    sha = recs[0].get("kernel_sources_sha") if recs else None
    if sha is not None and sha != kernel_sources_sha():
        return None, fn + " (stale: kernel sources changed since the PMC passes)"
    ma = re.match(r"attn16_(fwd|bwd)<(split|bf16)>", dname)
    m = re.match(r"gemm_(nt|tn)_(h2|b16)(tr)?<(\d),(\d)>", dname)
    if ma:
        tmpl = f"prh::attn16_{ma.group(1)}_kernel<{0 if ma.group(2) == 'split' else 1}>"
        cands = [r for r in recs if r["kernel"] == tmpl]
    elif not m:
        return None, fn
    elif m.group(2) == "b16":
        tmpl = (f"prh::gemm_tn_b16_kernel<{m.group(5)}>" if m.group(1) == "tn"
                else f"prh::gemm_nt_b16_kernel<{m.group(4)}, {m.group(5)},")
        cands = [r for r in recs if r["kernel"].startswith(tmpl)]
        if m.group(1) == "nt" and m.group(4) == "0":      # plain operand: the DMA + phase-split core <EPI, C16>
            cands += [r for r in recs if r["kernel"].startswith(f"prh::gemm_nt_b16d_kernel<{m.group(5)},")]
    else:
        if m.group(1) == "tn" and m.group(3):      # wgrad core with transposed fragment reads
            tmpl = f"prh::gemm_tn_tr_kernel<{m.group(5)}>"
        elif m.group(1) == "tn":                   # column-staged wgrad core, two fp16 planes
            tmpl = f"prh::gemm_tn_s3_kernel<{m.group(4)}, {m.group(5)}, 2>"
        else:                                      # <PRO, EPI, PP>: PP = phase-split k-loop (PRH_H2_PP)
            tmpl = f"prh::gemm_nt_h2_kernel<{m.group(4)}, {m.group(5)},"
        cands = [r for r in recs if r["kernel"] == tmpl or (tmpl.endswith(",") and r["kernel"].startswith(tmpl))]
    if not cands:
        return None, fn
    r = max(cands, key=lambda r: r["fetch_bytes_largest_launch"])      # the fusion-layer launch
    return r["fetch_bytes_largest_launch"] + (r["write_bytes_largest_launch"] or 0.0), fn


class StubNet:
    """CPU stand-in with the model's call contract, for the launcher self-test (--stub)."""

    @staticmethod
    def build():
        import torch

        class Tiny(torch.nn.Module):
            def __init__(self):
                super().__init__()
                self.l1 = torch.nn.Linear(4, 16)
                self.bn = torch.nn.BatchNorm1d(16)
                self.l2 = torch.nn.Linear(16, 3)

            def forward(self, context, noisy_line):
                h = torch.relu(self.bn(self.l1(context.reshape(-1, 4))))
                g = h.reshape(context.shape[0], -1, 16).max(dim=1)[0]
                off = self.l2(g).unsqueeze(1) + 0 * noisy_line
                return torch.stack([off + noisy_line * 0.1 * l for l in range(6)])

        return Tiny()


def parity_check(step, model, batch, lib, mode, dev, batch_bench=None, vhat=None):
    """One step at the benchmark size with the benchmarked GEMM cores and one with the exact-fp32
    MFMA cores: same inputs, same weights and BatchNorm buffers, same dropout seeds (the hash
    masks are functions of seeds drawn from torch's CPU generator).  No optimiser step.  vhat: Adam's
    bias-corrected second-moment estimate, laid out like the flat gradient buffer - when given, every
    tensor also gets the RMS of (g_bench - g_exact) / (sqrt(vhat) + 1e-8): the error of the parameter
    update this gradient would cause, in units of the learning rate."""
    import torch
    from pointnet_refine_amd import ops
    snap = {k: v.detach().clone() for k, v in model.state_dict().items()}
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    res, ms = {}, {}
    step.keep_out = True
    try:
        for tag, m in (("bench", mode), ("exact", 0)):
            lib.prh_set_gemm_mode(m)
            with torch.no_grad():
                for k, v in model.state_dict().items():
                    v.copy_(snap[k])
            args_ = batch_bench if (tag == "bench" and batch_bench is not None) else batch
            # the exact-fp32 cores run here for the first time in the process: one untimed pass takes the
            # workspace growth and the first-launch costs (7.6 s at B=4096 where the steady state is ~2 s)
            for rep in range(2 if tag == "exact" else 1):
                with torch.no_grad():
                    for k, v in model.state_dict().items():
                        v.copy_(snap[k])
                torch.manual_seed(20240217)
                step.grads.zero()
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                loss = step.forward_backward(*args_)
                torch.cuda.synchronize(dev)
                ms[tag] = (time.perf_counter() - t0) * 1e3
            res[tag] = (step.last_out.clone(), step.grads.flat.clone(), float(loss))
            step.last_out = None
            ops.release_workspaces()
            torch.cuda.empty_cache()
    finally:
        step.keep_out = False
        lib.prh_set_gemm_mode(mode)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(snap[k])
    (oa, ga, la), (ob, gb, lb) = res["bench"], res["exact"]
    out_err = float((oa - ob).abs().max())
    out_rel = float((oa.double() - ob.double()).norm() / (ob.double().norm() + 1e-30))
    worst, worst_name, off = 0.0, "", 0
    gmax = 0.0
    per = []
    for n, p, o in zip(names, step.grads.params, step.grads.offsets):
        k = p.numel()
        off = o + k
        a, b = ga[o:off].double(), gb[o:off].double()
        upd = None
        if vhat is not None:
            d = vhat[off - k:off].double().sqrt() + 1e-8
            upd = float(((a - b) / d).square().mean().sqrt())
        per.append((n, float(b.norm()), float((a - b).norm()), upd))
        gmax = max(gmax, per[-1][1])
    for n, bn, dn, _ in per:
        if bn > 1e-7 * gmax and bn > 0:          # tensors that receive a gradient at all
            r = dn / bn
            if r > worst:
                worst, worst_name = r, n
    total = float((ga.double() - gb.double()).norm() / (gb.double().norm() + 1e-30))
    return {"out_max_abs_vs_exact_fp32": out_err, "out_rel_l2_vs_exact_fp32": out_rel, "worst_grad_rel_l2": worst, "worst_grad": worst_name,
            "all_grads_rel_l2": total, "loss": la, "loss_exact_fp32": lb, "per_tensor": per,
            "fwd_bwd_ms": round(ms["bench"], 2), "exact_fp32_fwd_bwd_ms": round(ms["exact"], 2)}


GRAD_BOUND_REDUCED = 1e-1      # reduced-precision modes: per-tensor gradient rel-L2 against the exact-fp32 cores
GRAD_BOUND_FP32 = 2e-3         # fp32-accurate modes (1e-3; ReLU / arg-max flips inside fp32 noise add O(1e-3))
UPDATE_BOUND = 1e-1            # reduced precision, trained state: RMS Adam-update error per tensor, in units of lr


def grade_parity(parity, mode, at_init=None):
    """Turns parity_check's record into the gated `parity` object of the bench line: EVERY tensor
    that receives a gradient (norm above 1e-7 of the largest tensor's) is held to a per-tensor bound,
    not only the outputs.

    fp32-accurate modes: rel-L2 <= 2e-3 per tensor at the benchmarked (trained) state.
    Reduced precision (bf16 mode): a forward rounded to 8 bits moves the point at which the gradient
    is evaluated, which costs an ABSOLUTE gradient error that does not shrink with the gradient
    (profiles/r03b_bf16_grad_budget.txt: the encoder's gradients fall 200-500x over the first 25 Adam
    steps of the benchmark, their absolute error 3x) - so rel-L2 against a vanishing gradient is not a
    bounded quantity for ANY 8-bit forward.  Gated instead: (a) rel-L2 <= 1e-1 per tensor at the
    INITIAL weights (`at_init`, same size, same inputs), where every tensor carries signal; (b) at the
    trained state, the error of the Adam update the gradient causes, RMS per tensor of
    (g_bf16 - g_exact) / (sqrt(vhat) + eps) <= 0.1 learning rates; the raw rel-L2 figures of the trained
    state are reported next to it (`grad_tensors_over_bound`)."""
    def live_rel(per):
        gmax = max((t[1] for t in per), default=0.0)
        return [(t[0], t[2] / t[1]) for t in per if t[1] > 1e-7 * gmax and t[1] > 0]

    per = parity.pop("per_tensor")
    accurate = mode in (0, 1, 3)
    bound = GRAD_BOUND_FP32 if accurate else GRAD_BOUND_REDUCED
    live = live_rel(per)
    over = sorted(((n, r) for n, r in live if r > bound), key=lambda t: -t[1])
    parity["grad_tensors_checked"] = len(live)
    parity["grad_tensors_over_bound"] = [[n, float(f"{r:.3g}")] for n, r in over[:8]]
    rs = sorted(r for _, r in live)
    parity["grad_rel_l2_median"] = float(f"{rs[len(rs) // 2]:.3g}") if rs else None
    for k in ("out_max_abs_vs_exact_fp32", "out_rel_l2_vs_exact_fp32", "worst_grad_rel_l2", "all_grads_rel_l2"):
        parity[k] = float(f"{parity[k]:.4g}")
    if accurate:      # fp32-accurate cores: the north_star's 1e-4 on the outputs
        parity["gate"] = {"out_max_abs": 1e-4, "per_tensor_grad_rel_l2": bound}
        parity["ok"] = bool(parity["out_max_abs_vs_exact_fp32"] <= 1e-4 and not over)
        return parity
    # reduced precision: SURVEY 8(d)'s 5e-2 on the outputs, read as a relative figure (it was taken from
    # the reference under bf16 autocast: 1.7e-2 rel-L2 = 7.9e-2 max-abs in eval mode, 5.9e-2 / 4.0e-1 in
    # train mode on the G2 inputs - tests/golden/g9_bf16_autocast.npz), and the gradient gates above
    ok = parity["out_rel_l2_vs_exact_fp32"] <= 5e-2
    gate = {"out_rel_l2": 5e-2, "reference_under_bf16_autocast": {"train_out_rel_l2": 5.9e-2, "train_out_max_abs": 0.397}}
    upd = [(t[0], t[3]) for t in per if t[3] is not None]
    if upd:
        wn, wu = max(upd, key=lambda t: t[1])
        parity["adam_update_err_rms_lr_worst"] = float(f"{wu:.3g}")
        parity["adam_update_err_worst_tensor"] = wn
        gate["adam_update_err_rms_lr"] = UPDATE_BOUND
        ok = ok and wu <= UPDATE_BOUND
    if at_init is not None:
        li = live_rel(at_init.pop("per_tensor"))
        oi = sorted(((n, r) for n, r in li if r > bound), key=lambda t: -t[1])
        wi = max(li, key=lambda t: t[1]) if li else ("", 0.0)
        parity["at_init"] = {"grad_tensors_checked": len(li), "worst_grad_rel_l2": float(f"{wi[1]:.3g}"), "worst_grad": wi[0],
                             "grad_tensors_over_bound": [[n, float(f"{r:.3g}")] for n, r in oi[:8]],
                             "out_rel_l2_vs_exact_fp32": float(f"{at_init['out_rel_l2_vs_exact_fp32']:.3g}"),
                             "out_max_abs_vs_exact_fp32": float(f"{at_init['out_max_abs_vs_exact_fp32']:.3g}")}
        gate["at_init_per_tensor_grad_rel_l2"] = bound
        ok = ok and not oi
    elif not upd:
        gate["per_tensor_grad_rel_l2"] = bound
        ok = ok and not over
    parity["gate"] = gate
    parity["ok"] = bool(ok)
    return parity


def _timed_steps(fn, warmup, steps, dev):
    import torch
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3


def extra_workloads(dev, lib, headline_mode, budget_s=270.0):
    """The other BASELINE.json configs under the same clock as the headline, run AFTER its timed
    region and parity step (rank 0, N=1): config 2 (B=512, N=1024 step + per-tensor parity against the exact-fp32 cores),
    config 3 (bf16 training step + per-tensor parity), config 5
    (whole-scene refinement, fp16 forward), the encoder alone through the single fused kernel (the
    north_star's fused shared MLP + max-pool; both precisions), config 4's per-rank share (B=512,
    N=2048) and the reference's own per-GPU batch (B=32, N=2048; eager and HIP graph).  Each entry
    carries its own wall time; an entry that fails or no longer fits the budget says so."""
    import numpy as np
    import torch
    from pointnet_refine_amd import ops
    from pointnet_refine_amd.model import LineRefineNet, MultiScalePointNetEncoder
    from pointnet_refine_amd.synth import synthetic_batch
    from pointnet_refine_amd.train_step import TrainStep
    t_start = time.perf_counter()
    out = {}

    def clean():
        ops.release_workspaces()
        torch.cuda.empty_cache()

    def guarded(name, fn):
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": f"budget of {budget_s:.0f} s used up"}
            return
        t0 = time.perf_counter()
        try:
            out[name] = fn()
        except Exception as e:      # report, never hide
            out[name] = {"error": f"{type(e).__name__}: {str(e)[:160]}"}
        finally:
            lib.prh_set_gemm_mode(headline_mode)
            clean()
        out[name]["wall_s"] = round(time.perf_counter() - t0, 2)
        log(f"workload {name}: {out[name]}")

    def train_wl(B, N, mode, steps, warmup, graph=False, parity=False, chunk=4096):
        lib.prh_set_gemm_mode(mode)
        torch.manual_seed(0)
        model = LineRefineNet().to(dev).train()
        step = TrainStep(model, None, decoder_chunk=chunk, graph=graph)
        batch = synthetic_batch(B, N, dev, seed=1234)
        p_init = parity_check(step, model, batch, lib, mode, dev) if (parity and mode in (2, 4)) else None
        torch.cuda.reset_peak_memory_stats(dev)
        ms = _timed_steps(lambda: step(*batch), warmup, steps, dev)
        r = {"workload": f"LineRefineNet training step, B={B}, N={N}" + (", HIP graph" if graph else ""),
             "dtype": MODE_INFO[mode][3], "ms_per_step": round(ms, 3), "segments_per_s": round(B / ms * 1e3, 1),
             "steps": steps, "warmup": warmup, "max_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)}
        if parity:
            vhat = step.opt.exp_avg_sq / (1.0 - step.opt.param_groups[0]["betas"][1] ** step.opt.steps)
            r["parity"] = grade_parity(parity_check(step, model, batch, lib, mode, dev, vhat=vhat if mode in (2, 4) else None),
                                       mode, p_init)
        step.close()
        return r

    guarded("config3_bf16_train_step", lambda: train_wl(4096, 1024, 4, steps=6, warmup=2, parity=True))
    guarded("config2_B512_N1024_fp32_accurate_step", lambda: train_wl(512, 1024, 3, steps=6, warmup=2, parity=True))
    guarded("config4_per_rank_share_B512_N2048", lambda: train_wl(512, 2048, 3, steps=6, warmup=2))
    guarded("reference_batch_B32_N2048_eager", lambda: train_wl(32, 2048, 3, steps=30, warmup=5, chunk=None))
    guarded("reference_batch_B32_N2048_graph", lambda: train_wl(32, 2048, 3, steps=30, warmup=5, graph=True, chunk=None))

    def encoder_eval(prec):
        torch.manual_seed(0)
        enc = MultiScalePointNetEncoder(4, 1024).to(dev).eval()
        enc.inference_precision = prec
        B, N = 4096, 1024
        ctx, _, _ = synthetic_batch(B, N, dev, seed=1234)
        with torch.no_grad():
            ms = _timed_steps(lambda: enc.forward_pointmajor(ctx, True), 2, 5, dev)
        flops = 5_587_584.0 * N * B
        products = 1 if prec == "fp16" else 3
        peak = BF16_MFMA_PEAK_TFLOPS / products
        byts = (16.0 * N + 4096.0 * N + 8192.0) * B
        return {"workload": "MultiScalePointNetEncoder eval forward -> (global_feat, fused), ONE fused kernel, B=4096, N=1024",
                "dtype": "fp16 operands, fp32 accumulate" if prec == "fp16" else "f32 (2xfp16-split MFMA, 3 products)",
                "ms": round(ms, 3), "segments_per_s": round(B / ms * 1e3, 1),
                "mfma": {"achieved_tflops": round(flops / ms / 1e9, 1), "peak": round(peak, 1),
                         "frac": round(flops / ms / 1e9 / peak, 4)},
                "hbm": {"algorithmic_gbs": round(byts / ms / 1e6, 1), "frac_of_8TBs": round(byts / ms / 1e6 / HBM_PEAK_GBS, 4)}}

    def encoder_train(mode):
        lib.prh_set_gemm_mode(mode)
        torch.manual_seed(0)
        enc = MultiScalePointNetEncoder(4, 1024).to(dev).train()
        B, N = 4096, 1024
        ctx, _, _ = synthetic_batch(B, N, dev, seed=1234)
        up_g = torch.randn(B, 2048, device=dev)
        up_f = torch.randn(B, N, 1024, device=dev, dtype=torch.bfloat16 if mode == 4 else torch.float32)

        def step():
            for p_ in enc.parameters():
                p_.grad = None
            gf, fu = enc.forward_pointmajor(ctx, True)
            torch.autograd.backward([gf, fu], [up_g, up_f])
        torch.cuda.reset_peak_memory_stats(dev)
        ms = _timed_steps(step, 1, 3, dev)
        flops = 3.0 * 5_587_584.0 * N * B
        peak = MODE_INFO[mode][1] / MODE_INFO[mode][0]
        return {"workload": "MultiScalePointNetEncoder (shared MLP + fusion + gate + max/mean pool) train-mode forward + backward, "
                            "gradients on global_feat and fused, B=4096, N=1024",
                "dtype": MODE_INFO[mode][3], "ms": round(ms, 2), "segments_per_s": round(B / ms * 1e3, 1),
                "mfma": {"achieved_tflops": round(flops / ms / 1e9, 1), "peak": round(peak, 1), "frac": round(flops / ms / 1e9 / peak, 4)},
                "max_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1)}

    guarded("encoder_train_fwd_bwd", lambda: encoder_train(3))
    guarded("encoder_eval_fused_fp16", lambda: encoder_eval("fp16"))
    guarded("encoder_eval_fused_fp32_accurate", lambda: encoder_eval("fp32"))

    def scene():
        from pointnet_refine_amd.io import refine_scene
        P, L = 100_000, 4096
        rng = np.random.default_rng(0)
        lines = []
        for _ in range(L):
            x = np.sort(rng.uniform(-60, 60, 6))
            lines.append(np.stack([x, rng.uniform(-40, 40) + 0.2 * np.sin(x / 9), rng.normal(0, 0.02, 6)], 1))
        xyz = np.stack([rng.uniform(-60, 60, P), rng.uniform(-40, 40, P), rng.normal(0, 0.05, P)], 1)
        cloud = torch.from_numpy(np.column_stack([xyz, np.clip(rng.exponential(12, P), 0, 255)]).astype(np.float32)).to(dev)
        torch.manual_seed(0)
        model = LineRefineNet().to(dev).eval()
        res = {}
        for prec in ("layers", "fp32", "fp16"):
            refine_scene(model, cloud, lines[:2048], precision=prec)
            torch.cuda.synchronize(dev)
            best = 1e9
            for _ in range(2):
                t0 = time.perf_counter()
                refined, _ = refine_scene(model, cloud, lines, seed=1, precision=prec)
                torch.cuda.synchronize(dev)
                best = min(best, time.perf_counter() - t0)
            res[prec] = (best, refined)
        ref = res["layers"][1]
        return {"workload": "io.refine_scene: 100k-point cloud, 4096 polylines -> 4096 x (1024, 4) contexts on the GPU, "
                            "eval forward in batches of 2048, end to end incl. copy back",
                "fp16_lines_per_s": round(L / res["fp16"][0], 1), "fp16_ms": round(res["fp16"][0] * 1e3, 1),
                "fp16_max_abs_vs_per_layer_fp32": float(f"{float(np.abs(res['fp16'][1] - ref).max()):.3g}"),
                "fp32_accurate_lines_per_s": round(L / res["fp32"][0], 1),
                "fp32_accurate_max_abs_vs_per_layer_fp32": float(f"{float(np.abs(res['fp32'][1] - ref).max()):.3g}"),
                "per_layer_fp32_lines_per_s": round(L / res["layers"][0], 1)}

    guarded("config5_whole_scene", scene)
    out["total_wall_s"] = round(time.perf_counter() - t_start, 1)
    return out


def run(args):
    import ctypes as C
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ      # torchrun / torch.distributed.run
    if launched:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if args.stub else "nccl")     # "nccl" = RCCL on ROCm
    if args.stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    from pointnet_refine_amd.train_step import TrainStep
    lib = None
    mode = GEMM_MODES[args.gemm]
    B, N = args.batch, args.points
    torch.manual_seed(0)
    if args.stub:
        from pointnet_refine_amd.synth import synthetic_batch
        model = StubNet.build().train()
        opt = torch.optim.Adam(model.parameters(), lr=1e-3)
        step = TrainStep(model, opt, decoder_chunk=None, world_size=world,
                         loss_fn=lambda out, target, denom=None: (out - target.unsqueeze(0)).abs().sum()
                         / (out.numel() if denom is None else denom))
        ctx, noisy, target = synthetic_batch(B, N, dev, seed=1234 + rank)
    else:
        from pointnet_refine_amd import _lib
        from pointnet_refine_amd.model import LineRefineNet
        from pointnet_refine_amd.synth import synthetic_batch
        lib = _lib.lib()
        _lib.check(lib.prh_set_gemm_mode(mode), "prh_set_gemm_mode")
        model = LineRefineNet().to(dev).train()
        # optimizer=None: the library's flat-buffer Adam (same update as torch.optim.Adam(lr=1e-3),
        # tests/test_loss_adam_gpu.py), one launch per step; the loss is the fused HIP L1 either way
        opt = torch.optim.Adam(model.parameters(), lr=1e-3) if args.torch_adam else None
        step = TrainStep(model, opt, decoder_chunk=args.decoder_chunk, world_size=world, graph=args.graph)
        ctx, noisy, target = synthetic_batch(B, N, dev, seed=1234 + rank)
    if launched:
        for p in model.parameters():                 # same start on every rank (DDP does this)
            dist.broadcast(p.data, 0)

    def sync():
        if launched:
            if args.stub:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        if not args.stub:
            torch.cuda.synchronize(dev)

    def mem_gib():
        return 0.0 if args.stub else torch.cuda.max_memory_allocated(dev) / 2**30

    parity_init = None
    if lib is not None and mode in (2, 4) and not args.no_parity and not args.graph:
        # reduced precision: per-tensor gradient parity at the INITIAL weights (see grade_parity)
        try:
            parity_init = parity_check(step, model, (ctx, noisy, target), lib, mode, dev)
        except torch.cuda.OutOfMemoryError:
            parity_init = None
    for i in range(args.warmup):
        loss = step(ctx, noisy, target)
        if rank == 0:
            if not args.stub:
                torch.cuda.synchronize(dev)
            log(f"warmup {i} done, loss {float(loss):.5f}, mem {mem_gib():.1f} GiB")
    sync()
    # live per-launch HIP events: room for every GEMM launch of the timed region (about 500 per
    # step and decoder chunk pair); if it still fills up only whole steps are counted
    cap = 0
    marks = []
    if lib is not None:
        chunks = max(1, -(-B // max(1, args.decoder_chunk)))
        cap = min(262144, (args.steps + 1) * (400 + 300 * chunks))
        lib.prh_profile_enable(cap)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step(ctx, noisy, target)
        if lib is not None:
            marks.append(lib.prh_profile_count())
    sync()
    dt = time.perf_counter() - t0
    if rank == 0:
        log(f"timed {args.steps} steps in {dt:.3f} s")
    if launched:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    roofline = None
    if lib is not None:
        if args.graph:
            # the library's per-launch events are not recorded inside a replayed graph: one eager
            # step after the timed region supplies the per-kernel figures of the roofline object
            from pointnet_refine_amd import ops
            lib.prh_profile_reset()
            step.use_graph = False
            step.close()                 # the graph's pool and an eager step do not fit together at B=4096
            ops.release_workspaces()
            torch.cuda.empty_cache()
            step(ctx, noisy, target)
            sync()
            marks = [lib.prh_profile_count()]
        # whole steps whose launches were all recorded
        covered = [m for m in marks if m < cap]
        n_steps_prof = len(covered)
        n_rec = covered[-1] if covered else 0
        agg = {}
        name = C.create_string_buffer(64)
        ms, fl, by = C.c_float(), C.c_double(), C.c_double()
        for i in range(n_rec):
            lib.prh_profile_read(i, name, 64, C.byref(ms), C.byref(fl), C.byref(by))
            a = agg.setdefault(name.value.decode(), [0, 0.0, fl.value, by.value])
            a[0] += 1
            a[1] += ms.value
        lib.prh_profile_enable(0)
        hip_ms = sum(a[1] for a in agg.values())
        if rank == 0 and args.kernels > 0:
            for k, (c, t, f, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.kernels]:
                print(f"[bench] {k:44s} x{c:4d} {t / c:8.3f} ms  {f / (t / c * 1e-3) / 1e12:7.1f} TF", file=sys.stderr)
        if agg:
            # dominant = the kernel (name incl. its GEMM shape) with the most time in the timed region
            dname, (cnt, tot_ms, flops, bytes_) = max(agg.items(), key=lambda kv: kv[1][1])
            avg_ms = tot_ms / cnt
            achieved = flops / (avg_ms * 1e-3) / 1e12
            # achieved = ALGORITHMIC FLOPs (2*M*N*K) / live launch time.  Peak of the core the
            # kernel runs on: the exact fp32 MFMA pipe, or the dense 16-bit MFMA peak divided by
            # the products the core issues per algorithmic MAC.
            on_fp32_core = not any(t in dname for t in ("_s3", "_b1", "_h2", "_b16", "attn16", "encoder_fused"))
            products, peak16, basis, _ = MODE_INFO[mode]
            if on_fp32_core:
                products, peak16, basis = 1, FP32_MFMA_PEAK_TFLOPS, MODE_INFO[0][2]
            elif "<split>" in dname:              # attention cores: three fp16 products whatever the GEMM mode
                products, peak16, basis = MODE_INFO[3][:3]
            elif "<bf16>" in dname:
                products, peak16, basis = MODE_INFO[4][:3]
            peak = peak16 / products
            traffic, tfile = pmc_traffic(dname, B, N, mode)
            sf = step_flops(B, N)
            s_ach = sf / (dt / args.steps) / 1e12
            mode_peak = MODE_INFO[mode][1] / MODE_INFO[mode][0]
            roofline = {"bound": "mfma", "kernel": dname, "achieved": round(achieved, 2),
                        "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                        "peak_basis": basis,
                        "issued_mfma_tflops": round(achieved * products, 1),
                        "vs_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                        "traffic": traffic,
                        "traffic_source": (f"profiles/{tfile} (rocprofv3 --pmc FETCH_SIZE x2, WRITE_SIZE)"
                                           if tfile else None),
                        "algorithmic_bytes": bytes_,
                        "avg_launch_ms": round(avg_ms, 4), "launches": cnt,
                        "launches_per_step": round(cnt / max(1, n_steps_prof), 2),
                        "algorithmic_gbs": round(bytes_ / (avg_ms * 1e-3) / 1e9, 1),
                        "hbm_frac": round(bytes_ / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                        "profiled_steps": n_steps_prof,
                        "hip_gemm_ms_per_step": round(hip_ms / max(1, n_steps_prof), 2),
                        "step": {"flops": sf, "achieved_tflops": round(s_ach, 1),
                                 "peak": round(mode_peak, 1), "frac": round(s_ach / mode_peak, 4),
                                 "basis": "whole training step: 3 x (8,013,952*N + 12,482,432*32) FLOP per "
                                          "segment (SURVEY 8d) / step time, against the peak of this GEMM mode"}}

    parity = None
    if lib is not None and not args.no_parity and not args.graph:
        try:
            vhat = None
            if mode in (2, 4) and getattr(step.opt, "exp_avg_sq", None) is not None and step.opt.steps > 0:
                vhat = step.opt.exp_avg_sq / (1.0 - step.opt.param_groups[0]["betas"][1] ** step.opt.steps)
            parity = grade_parity(parity_check(step, model, (ctx, noisy, target), lib, mode, dev, vhat=vhat), mode, parity_init)
            if rank == 0:
                log(f"parity at B={B}, N={N}: {parity}")
        except torch.cuda.OutOfMemoryError as e:       # report, never hide
            parity = {"ok": None, "error": f"parity step did not fit: {str(e)[:120]}"}
    parity_failed = bool(parity is not None and parity.get("ok") is False)
    if launched and parity is not None:
        f = torch.tensor([1.0 if parity_failed else 0.0], device=dev)
        dist.all_reduce(f, op=dist.ReduceOp.MAX)
        parity_failed = bool(f.item() > 0)

    workloads = None
    if (rank == 0 and world == 1 and lib is not None and not args.no_workloads and not args.graph
            and B == 4096 and N == 1024):
        # free the headline run's buffers first: config 3 alone needs ~140 GB
        final_loss = float(loss)
        from pointnet_refine_amd import ops as _ops
        step.close()
        del step, model, opt
        ctx = noisy = target = None
        _ops.release_workspaces()
        torch.cuda.empty_cache()
        hl_mem = mem_gib()
        workloads = extra_workloads(dev, lib, mode)
    else:
        final_loss, hl_mem = float(loss), mem_gib()

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        backend = "gloo, CPU stub" if args.stub else "RCCL gradient all-reduce"
        line = {
            "metric": ("STUB (launcher self-test, not a measurement) " if args.stub else "") +
                      "lane segments/sec (fwd+bwd) at B=4096,N=1024; HBM GB/s vs roofline",
            "value": round(world * B * args.steps / dt, 2), "unit": "segments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 (CPU stub)" if args.stub else MODE_INFO[mode][3],
            "data": "synthetic",
            "config": {"workload": ("CPU stand-in step, " if args.stub else "") +
                                   f"LineRefineNet training step (fwd + deep-supervision L1 + bwd + Adam), "
                                   f"B={B}/GPU, N={N}, M=32, C=4",
                       "global_batch": world * B, "points": N,
                       "parallelism": f"dp{world}" + (f" ({backend})" if world > 1 else ""),
                       "decoder_chunk": args.decoder_chunk},
            "loss": round(final_loss, 6),
            "max_mem_gb": round(hl_mem, 1),
        }
        if roofline is not None:
            line["roofline"] = roofline
        if parity is not None:
            line["parity"] = parity
        if workloads is not None:
            line["workloads"] = workloads
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(N, args.cpu_batch)
        emit(line)
    if launched:
        dist.destroy_process_group()
    if parity_failed:
        log("PARITY GATE FAILED at the benchmark size - see the parity object of the line above")
        return 4
    return 0


def run_encoder(args):
    """--encoder-only: the encoder API path (src/model.py:39-62) with both returns live."""
    import ctypes as C
    import torch
    from pointnet_refine_amd import _lib
    from pointnet_refine_amd.model import MultiScalePointNetEncoder
    from pointnet_refine_amd.synth import synthetic_batch
    if args.gpus != 1 or "RANK" in os.environ:
        raise SystemExit("--encoder-only is a single-GPU measurement (replicas only: no exchange on this path)")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    lib = _lib.lib()
    mode = GEMM_MODES[args.gemm]
    _lib.check(lib.prh_set_gemm_mode(mode), "prh_set_gemm_mode")
    B, N = args.batch, args.points
    torch.manual_seed(0)
    enc = MultiScalePointNetEncoder(4, 1024).to(dev)
    ctx, _, _ = synthetic_batch(B, N, dev, seed=1234)
    x_cm = ctx.transpose(2, 1)                       # the reference's (B, C, N) view; read in place, never copied
    flop_seg = 5_587_584.0 * N                       # encoder forward, 2 * 2,793,792 MAC per point
    if args.eval:
        enc.eval()
        enc.inference_precision = "fp16" if mode == 4 else "fp32"
        ctx_pm = ctx

        def step():
            with torch.no_grad():
                return enc.forward_pointmajor(ctx_pm, True)
        what = (f"eval forward, ONE fused kernel (BatchNorm folded, pooling in the epilogue), "
                f"{'one fp16 plane' if mode == 4 else 'two fp16 planes / three products, fp32-level error'}")
        flops = flop_seg * B
        bytes_seg = 16.0 * N + 4096.0 * N + 8192.0    # context in, fused fp32 + global_feat out
    else:
        enc.train()
        up_g = torch.randn(B, 2048, device=dev)
        up_f = torch.randn(B, N, 1024, device=dev, dtype=torch.bfloat16 if mode == 4 else torch.float32)

        def step():
            for p_ in enc.parameters():
                p_.grad = None
            gf, fu = enc.forward_pointmajor(ctx, True)
            torch.autograd.backward([gf, fu], [up_g, up_f])
            return gf, fu
        what = "train-mode forward + backward (batch statistics), gradients on global_feat and fused"
        flops = 3.0 * flop_seg * B
        bytes_seg = None
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize(dev)
        log(f"warmup {i} done, mem {torch.cuda.max_memory_allocated(dev) / 2**30:.1f} GiB")
    lib.prh_profile_enable(min(262144, (args.steps + 1) * 200))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    agg = {}
    name = C.create_string_buffer(64)
    ms, fl, by = C.c_float(), C.c_double(), C.c_double()
    for i in range(lib.prh_profile_count()):
        lib.prh_profile_read(i, name, 64, C.byref(ms), C.byref(fl), C.byref(by))
        a = agg.setdefault(name.value.decode(), [0, 0.0, fl.value, by.value])
        a[0] += 1
        a[1] += ms.value
    lib.prh_profile_enable(0)
    if args.kernels > 0:
        for k, (c, t, f, _) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:args.kernels]:
            print(f"[bench] {k:44s} x{c:4d} {t / c:8.3f} ms  {f / (t / c * 1e-3) / 1e12:7.1f} TF", file=sys.stderr)
    dname, (cnt, tot_ms, kflops, kbytes) = max(agg.items(), key=lambda kv: kv[1][1])
    avg_ms = tot_ms / cnt
    products, peak16, basis, dtype = MODE_INFO[mode]
    if args.eval:
        products = 1 if mode == 4 else 3
        peak16, basis = BF16_MFMA_PEAK_TFLOPS, ("2500 TF dense fp16 MFMA" + ("" if mode == 4 else " / 3 products per fp32-accurate MAC"))
        dtype = "fp16 operands, fp32 accumulate" if mode == 4 else "f32 (2xfp16-split MFMA, 3 products, fp32 accumulate)"
    elif not any(t in dname for t in ("_s3", "_b1", "_h2", "_b16")):
        products, peak16, basis = 1, FP32_MFMA_PEAK_TFLOPS, MODE_INFO[0][2]
    peak = peak16 / products
    ach = kflops / (avg_ms * 1e-3) / 1e12
    line = {"metric": "lane segments/sec, MultiScalePointNetEncoder alone (shared MLP + fusion + gate + max/mean pool); "
                      "HBM GB/s vs roofline",
            "value": round(B * args.steps / dt, 2), "unit": "segments/s", "n_gpus": 1, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"MultiScalePointNetEncoder (global_feat, fused), {what}, B={B}, N={N}, C=4",
                       "global_batch": B, "points": N, "parallelism": "dp1"},
            "max_mem_gb": round(torch.cuda.max_memory_allocated(dev) / 2**30, 1),
            "roofline": {"bound": "mfma", "kernel": dname, "achieved": round(ach, 2), "peak": round(peak, 1),
                         "unit": "TFLOP/s", "frac": round(ach / peak, 4), "peak_basis": basis,
                         "avg_launch_ms": round(avg_ms, 4), "launches": cnt, "traffic": None,
                         "algorithmic_bytes": kbytes,
                         "algorithmic_gbs": round(kbytes / (avg_ms * 1e-3) / 1e9, 1),
                         "hbm_frac": round(kbytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "step": {"flops": flops, "achieved_tflops": round(flops / (dt / args.steps) / 1e12, 1),
                                  "peak": round(peak, 1), "frac": round(flops / (dt / args.steps) / 1e12 / peak, 4)}}}
    if bytes_seg is not None:
        gbs = bytes_seg * B / (dt / args.steps) / 1e9
        line["hbm"] = {"algorithmic_bytes_per_segment": bytes_seg, "achieved_gbs": round(gbs, 1),
                       "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4),
                       "note": "compulsory traffic only (context in, fused + global_feat out): the kernel is bound "
                               "by the matrix pipe and L2 weight streaming, not by HBM (SURVEY D8)"}
    emit(line)
    return 0


_RESULT_FD = None


def quiet_stdout():
    """stdout carries ONE JSON line.  Libraries write to file descriptor 1 behind Python's back
    (this RCCL build prints a three-line banner - ROCm version, hostname, library path - when the
    first communicator is made), so fd 1 is pointed at stderr for the whole run and the result
    line goes to a saved copy of the original descriptor."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(line):
    data = (json.dumps(line) + "\n").encode()
    if _RESULT_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_RESULT_FD, data)


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ and not args.encoder_only:
        return launch_ranks(args)      # the ranks' own stdout passes through: rank 0 prints the line
    quiet_stdout()
    if args.encoder_only:
        return run_encoder(args)
    return run(args)


if __name__ == "__main__":
    sys.exit(main())
