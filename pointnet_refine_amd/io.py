"""On-disk formats and the whole-scene driver - host code.

  load_pcd_data      PCD reader with the reference's behaviour (src/dataset.py:31-76): ASCII via
                     numpy, binary as 16-byte xyzi-f32 or 14-byte xyz-f32 + u2-intensity records
                     chosen by payload size, an empty (0,4) array (plus a message) for anything
                     else - the reference's callers rely on "never raises"
  load_scene_items   the scene JSON schema: items[].position / noisy_candidates[][] /
                     context_lines[][] of {x,y,z} (src/dataset.py:172-187,
                     tools/generate_train_data.py:184-209)
  refine_scene       inference_whole_scene.py:94-146 for all lines of a scene at once: contexts
                     from the GPU builder (context.py), batched eval forward, refined line =
                     resampled noisy line + last-layer offset
"""
import json

import numpy as np
import torch

from .context import build_contexts

_XYZI_F32 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4")])
_XYZ_F32_I_U2 = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<u2")])


def load_pcd_data(pcd_path):
    """(P,4) float32 xyz + intensity (reference: src/dataset.py:31-76)."""
    try:
        with open(pcd_path, "rb") as f:
            header = []
            while True:
                raw = f.readline()
                if not raw:
                    raise ValueError("no DATA line in the PCD header")
                line = raw.strip()
                header.append(line)
                if line.startswith(b"DATA"):
                    break
            kind = header[-1].split()[1]
            if kind == b"ascii":
                return np.loadtxt(pcd_path, skiprows=len(header), dtype=np.float32)
            payload = f.read()
        n_points = int([h for h in header if h.startswith(b"POINTS")][0].split()[1])
        for dt in (_XYZI_F32, _XYZ_F32_I_U2):
            if len(payload) == n_points * dt.itemsize:
                rec = np.frombuffer(payload, dtype=dt)
                return np.column_stack((rec["x"], rec["y"], rec["z"], rec["intensity"].astype(np.float32)))
        print(f"[pointnet_refine_amd.io] {pcd_path}: {len(payload)} payload bytes match neither the 16-byte "
              f"nor the 14-byte record layout for {n_points} points; returning an empty cloud")
        return np.zeros((0, 4), dtype=np.float32)
    except Exception as e:                       # the reference reports and carries on
        print(f"[pointnet_refine_amd.io] could not read {pcd_path} ({type(e).__name__}: {e}); returning an empty cloud")
        return np.zeros((0, 4), dtype=np.float32)


def _xyz(points):
    return np.array([[p["x"], p["y"], p["z"]] for p in points], dtype=np.float64).reshape(-1, 3)


def load_scene_items(json_path):
    """List of dicts with numpy polylines: 'position' (ground truth, may be absent),
    'noisy_candidates' (list), 'context_lines' (list, empty lines dropped)."""
    with open(json_path, "r") as f:
        data = json.load(f)
    items = []
    for item in data.get("items", []):
        items.append({
            "position": _xyz(item["position"]) if "position" in item else None,
            "noisy_candidates": [_xyz(c) for c in item.get("noisy_candidates", [])],
            "context_lines": [_xyz(l) for l in item.get("context_lines", []) if len(l) > 0],
        })
    return items


@torch.no_grad()
def refine_scene(model, pcd_points, raw_lines, num_line_points=32, num_context_points=1024,
                 crop_radius=0.3, decay_scale=2.0, batch_lines=2048, seed=0, precision=None):
    """Refine every polyline of a scene (inference_whole_scene.py:94-146, NUM_CONTEXT_POINTS 1024,
    CROP_RADIUS 0.3).  pcd_points (P,4) numpy or CUDA tensor; raw_lines list of (n_i,3).
    Returns (refined (L,M,3), noisy_resampled (L,M,3)) numpy arrays in scene coordinates.

    The eval forward takes the fused encoder kernel (csrc/prh_fused.hpp: context -> memory in one
    launch).  precision: None = the model's setting (fp32-accurate by default); "fp32"; "fp16" =
    BASELINE config 5's batched reduced-precision forward - encoder on one fp16 plane, decoder
    GEMMs on one bf16 plane (GEMM mode 4) for the duration of the call; "layers" = the per-layer
    kernels (the round-1 path, kept for comparison)."""
    dev = next(model.parameters()).device
    cloud = pcd_points if torch.is_tensor(pcd_points) else torch.from_numpy(np.ascontiguousarray(pcd_points, dtype=np.float32))
    cloud = cloud.to(dev, torch.float32)
    if cloud.dim() == 2 and cloud.shape[1] > 4:
        cloud = cloud[:, :4]                      # ASCII PCDs may carry extra fields: x y z intensity come first
    if len(raw_lines) == 0:
        return np.zeros((0, num_line_points, 3)), np.zeros((0, num_line_points, 3))
    was_training = model.training
    model.eval()
    from . import ops as _ops
    enc = getattr(model, "context_encoder", None)
    old_prec = getattr(enc, "inference_precision", None)
    if precision not in (None, "fp32", "fp16", "layers"):
        raise ValueError("refine_scene: precision must be None, 'fp32', 'fp16' or 'layers'")
    import contextlib
    # fp16: decoder GEMMs on one bf16 plane for the duration of the call (process-wide setting: the
    # scope holds a lock and restores the previous mode on the way out, exception or not)
    mode_scope = _ops.gemm_mode_scope("bf16") if precision == "fp16" else contextlib.nullcontext()
    try:
        with mode_scope:
            if precision is not None and enc is not None:
                enc.inference_precision = None if precision == "layers" else precision
            ctx, noisy_c, centres, _ = build_contexts(cloud, raw_lines, num_line_points, num_context_points,
                                                      crop_radius, decay_scale, seed)
            outs = []
            for s in range(0, ctx.shape[0], batch_lines):
                outs.append(model(ctx[s:s + batch_lines], noisy_c[s:s + batch_lines])[-1])   # last layer, :139-141
            offset = torch.cat(outs)
            noisy = noisy_c + centres[:, None, :]
            return (noisy + offset).cpu().numpy(), noisy.cpu().numpy()                         # :146
    finally:
        model.train(was_training)
        if enc is not None:
            enc.inference_precision = old_prec


class SceneSampleStream:
    """Training samples straight from scene files, built on the GPU, sharded over ranks.

    The reference's ``LaneRefineDataset`` (src/dataset.py:132-253) builds ONE sample per
    ``__getitem__`` in DataLoader workers: JSON parse, KDTree crop of the whole cloud, numpy
    sampling; ``train_dist.py:129-140`` feeds it through ``DistributedSampler(shuffle=True)`` and a
    ``DataLoader(batch_size=32)``.  Here a scene's cloud goes to the device once and the contexts
    of all of THIS RANK's samples of that scene (noisy candidates of its items, :152-166) come
    from one ``prh_context_build`` call.  Same constructor arguments and the same per-sample
    tensors: ``context (N,4)`` centred on the noisy line's mean with raw intensity, ``noisy_line
    (M,3)`` centred, ``target_offset (M,3)`` = resampled ground truth - resampled noisy line
    (:206-207,241).

    Sharding and order follow ``DistributedSampler(shuffle=True)`` + ``DataLoader(batch_size)``
    (train_dist.py:129-140): every epoch ALL (item, candidate) samples of the dataset are permuted from
    ``(seed, epoch)`` - identically on every rank - the list is padded ONCE, by wrap-around, to a
    multiple of ``world_size`` (fewer than ``world_size`` duplicates per epoch, as the sampler does) and
    rank r takes entries r, r + world, ...  What differs from the reference is only the grouping: a
    rank's samples are collected per scene, so a scene's cloud is parsed and uploaded once per rank
    that drew samples from it (and not at all by the others) and its contexts come from one kernel
    call; the scenes are visited in a permuted order, a rank's samples are pooled over ``mix_scenes``
    scenes, shuffled, and cut into batches of ``batch_size``.  Every rank sees the same NUMBER of
    samples, hence the same number of batches (all full but the last), so a gradient all-reduce per
    batch cannot dead-lock.  ``batch_size=None`` keeps one batch per scene (single-rank use: the number
    of batches then depends on which scenes a rank drew from).  ``rank`` / ``world_size`` default to
    the initialised process group (else 0 / 1).  Iterating yields dicts of CUDA tensors with a leading
    sample dimension; ``len()`` is the number of samples this rank sees per epoch."""

    def __init__(self, data_root, num_line_points=32, num_context_points=2048, crop_radius=4.0,
                 decay_scale=2.0, split="train", device="cuda", seed=0, batch_size=None, shuffle=True,
                 rank=None, world_size=None, mix_scenes=4, drop_last=False):
        import os
        import torch.distributed as dist
        self.num_line_points, self.num_context_points = num_line_points, num_context_points
        self.crop_radius, self.decay_scale = crop_radius, decay_scale
        self.device, self.seed, self.split = torch.device(device), seed, split
        grouped = dist.is_available() and dist.is_initialized()
        self.rank = int(rank if rank is not None else (dist.get_rank() if grouped else 0))
        self.world = int(world_size if world_size is not None else (dist.get_world_size() if grouped else 1))
        if not 0 <= self.rank < self.world:
            raise ValueError(f"SceneSampleStream: rank {self.rank} outside world_size {self.world}")
        self.batch_size, self.shuffle = batch_size, bool(shuffle)
        self.mix_scenes, self.drop_last = max(1, int(mix_scenes)), bool(drop_last)
        self.scenes = []                       # (pcd_path, json_path, [(item_idx, noise_idx), ...])
        for name in sorted(f for f in os.listdir(data_root) if f.endswith(".json")):
            json_path = os.path.join(data_root, name)
            pcd_path = json_path.replace(".json", ".pcd")
            if not os.path.exists(pcd_path):
                continue                                             # :149-150
            with open(json_path, "r") as f:
                data = json.load(f)
            pairs = [(i, k) for i, item in enumerate(data.get("items", []))
                     if "noisy_candidates" in item and "position" in item               # :156-157
                     for k in range(len(item["noisy_candidates"]))]
            if pairs:
                self.scenes.append((pcd_path, json_path, pairs))
        self.epoch = 0

    def __len__(self):
        total = sum(len(s[2]) for s in self.scenes)
        return -(-total // self.world)           # DistributedSampler.num_samples

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _plan(self):
        """[(scene index, this rank's (item, candidate) pairs)] for the current epoch."""
        g = np.random.default_rng([int(self.seed) & 0x7FFFFFFF, self.epoch])
        flat = [(si, pr) for si, sc in enumerate(self.scenes) for pr in sc[2]]
        if not flat:
            return []
        idx = g.permutation(len(flat)) if self.shuffle else np.arange(len(flat))
        per = -(-len(flat) // self.world)
        idx = np.resize(idx, per * self.world)                     # one wrap-around pad for the whole epoch
        mine = {}
        for j in idx[self.rank::self.world]:
            si, pr = flat[int(j)]
            mine.setdefault(si, []).append(pr)
        order = g.permutation(len(self.scenes)) if self.shuffle else np.arange(len(self.scenes))
        return [(int(si), mine.get(int(si), [])) for si in order]

    def _scene_samples(self, si, pairs):
        from .context import resample_polyline
        pcd_path, json_path, _ = self.scenes[si]
        cloud = torch.from_numpy(np.atleast_2d(load_pcd_data(pcd_path))[:, :4].astype(np.float32, copy=False))
        cloud = cloud.to(self.device, torch.float32).contiguous()
        items = load_scene_items(json_path)
        raw_noisy = [items[i]["noisy_candidates"][k] for i, k in pairs]
        gt = np.stack([resample_polyline(items[i]["position"], self.num_line_points) for i, _ in pairs])
        ctx, noisy_c, centres, counts = build_contexts(
            cloud, raw_noisy, self.num_line_points, self.num_context_points, self.crop_radius, self.decay_scale,
            seed=((self.seed * 1000003 + self.epoch) * 65537 + si) * 1021 + self.rank)
        gt_c = torch.from_numpy(gt).to(self.device, torch.float32) - centres[:, None, :]
        return {"context": ctx, "noisy_line": noisy_c, "target_offset": gt_c - noisy_c, "points_in_tube": counts}

    def __iter__(self):
        plan = self._plan()
        if self.batch_size is None:
            for si, pairs in plan:
                if pairs:
                    d = self._scene_samples(si, pairs)
                    d["scene"] = self.scenes[si][0]
                    yield d
            return
        keys = ("context", "noisy_line", "target_offset", "points_in_tube")
        g = torch.Generator().manual_seed(((int(self.seed) * 7919 + self.epoch) * 31 + 17) & 0x7FFFFFFF)
        pool = None
        bs = int(self.batch_size)

        def emit(final):
            nonlocal pool
            while pool is not None and (pool[keys[0]].shape[0] >= bs or (final and pool[keys[0]].shape[0] > 0
                                                                          and not self.drop_last)):
                yield {k: pool[k][:bs] for k in keys}
                pool = {k: pool[k][bs:] for k in keys} if pool[keys[0]].shape[0] > bs else None

        for j, (si, pairs) in enumerate(plan):
            if pairs:
                d = self._scene_samples(si, pairs)
                pool = d if pool is None else {k: torch.cat([pool[k], d[k]]) for k in keys}
            if (j + 1) % self.mix_scenes == 0 and pool is not None:
                if self.shuffle:
                    perm = torch.randperm(pool[keys[0]].shape[0], generator=g).to(self.device)
                    pool = {k: pool[k][perm] for k in keys}
                yield from emit(False)
        if pool is not None and self.shuffle:
            perm = torch.randperm(pool[keys[0]].shape[0], generator=g).to(self.device)
            pool = {k: pool[k][perm] for k in keys}
        yield from emit(True)
