"""GPU context builder: the per-line crop / weight / sample / centre step of
the reference's ``LaneRefineDataset.__getitem__`` (src/dataset.py:205-253) and
``process_single_line`` (inference_whole_scene.py:94-131), batched over all polylines of a scene
on the HIP path (``prh_context_build``).  Host side: polyline resampling (tiny, numpy).

    cloud = torch.from_numpy(pcd_points).float().cuda()            # (P,4) xyz + intensity
    ctx, noisy, centres, counts = build_contexts(cloud, raw_lines, num_context_points=1024,
                                                 crop_radius=0.3, seed=epoch)
    offsets = model(ctx, noisy)                                     # (6,L,32,3)
    refined = noisy + centres[:, None, :] + offsets[-1]             # original coordinates
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

DENSE_POINTS = 200          # src/dataset.py:214: tube continuity needs ~0.25 m spacing


def resample_polyline(points, num_points=32):
    """Arc-length resampling by linear interpolation (reference: src/dataset.py:8-29)."""
    points = np.asarray(points, dtype=np.float64).reshape(-1, 3)
    if len(points) < 2:
        return np.zeros((num_points, 3))
    cum = np.concatenate(([0.0], np.cumsum(np.linalg.norm(np.diff(points, axis=0), axis=1))))
    t = np.linspace(0.0, cum[-1], num_points)
    return np.stack([np.interp(t, cum, points[:, k]) for k in range(3)], axis=1)


def resample_polylines_device(raw_lines, num_points, device):
    """resample_polyline for a list of polylines at once, on the device in float64 (padded
    cumulative arc lengths, searchsorted, linear interpolation = numpy.interp): (L,num_points,3)
    float64.  Lines with fewer than 2 points give zeros, as in the reference."""
    L = len(raw_lines)
    out = torch.zeros((L, num_points, 3), dtype=torch.float64, device=device)
    lens = [len(l) for l in raw_lines]
    keep = [i for i, n in enumerate(lens) if n >= 2]
    if not keep:
        return out
    nmax = max(lens[i] for i in keep)
    pad = np.zeros((len(keep), nmax, 3))
    cnt = np.zeros(len(keep), dtype=np.int64)
    for r, i in enumerate(keep):
        a = np.asarray(raw_lines[i], dtype=np.float64).reshape(-1, 3)
        pad[r, :len(a)] = a
        pad[r, len(a):] = a[-1]                 # repeat the end point: zero-length segments
        cnt[r] = len(a)
    pts = torch.from_numpy(pad).to(device)
    seg = (pts[:, 1:] - pts[:, :-1]).norm(dim=2)
    cum = torch.cat([torch.zeros((len(keep), 1), dtype=torch.float64, device=device), seg.cumsum(1)], 1)
    total = cum[:, -1:]
    t = total * torch.linspace(0.0, 1.0, num_points, dtype=torch.float64, device=device)[None, :]
    t[:, -1] = total[:, 0]
    # numpy.interp: index of the last knot <= t among the line's own knots
    n = torch.from_numpy(cnt).to(device)
    idx = torch.searchsorted(cum, t, right=True) - 1
    idx = torch.minimum(idx.clamp_(min=0), (n - 2)[:, None])
    x0, x1 = cum.gather(1, idx), cum.gather(1, idx + 1)
    w = ((t - x0) / (x1 - x0).clamp_min(1e-300)).clamp_(0.0, 1.0)
    w = torch.where(x1 > x0, w, torch.zeros_like(w))
    p0 = pts.gather(1, idx[:, :, None].expand(-1, -1, 3))
    p1 = pts.gather(1, (idx + 1)[:, :, None].expand(-1, -1, 3))
    out[torch.tensor(keep, device=device)] = p0 + (p1 - p0) * w[:, :, None]
    return out


def build_contexts_resampled(cloud, dense, line, num_context_points=1024, crop_radius=0.3,
                             decay_scale=2.0, seed=0, max_candidates=None, return_weights=False):
    """cloud (P,4), dense (L,D,3), line (L,M,3) float32 CUDA tensors ->
    context (L,N,4) centred on each line's mean, counts (L,) int32 [, weights (L,max_candidates)].

    The reference samples from EVERY point of a line's tube (src/dataset.py:86-130).  The kernels
    compact a tube's points into a candidate buffer of `max_candidates` slots per line; with
    max_candidates=None (default) the buffer starts at max(4*N, 8192) slots and, when the largest
    tube of the call turns out to hold more points than that (dense LiDAR at the training radius),
    the call is repeated with the buffer sized from the true counts - the draw is then identical
    to what a large enough buffer gives in the first place (same seed, same candidates in cloud
    order).  An explicit max_candidates is taken as is and a RuntimeWarning reports any tube it
    truncated."""
    for t, name in ((cloud, "cloud"), (dense, "dense"), (line, "line")):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError(f"build_contexts: {name} must be a float32 CUDA tensor (there is no CPU fallback)")
    cloud, dense, line = cloud.contiguous(), dense.contiguous(), line.contiguous()
    if cloud.dim() != 2 or cloud.shape[1] != 4:
        raise RuntimeError(f"build_contexts: cloud must be (P,4), got {tuple(cloud.shape)}")
    if dense.dim() != 3 or line.dim() != 3 or dense.shape[0] != line.shape[0] or dense.shape[2] != 3 or line.shape[2] != 3:
        raise RuntimeError("build_contexts: dense (L,D,3) and line (L,M,3) expected")
    dev = cloud.device
    n_lines, n = dense.shape[0], int(num_context_points)
    npts = cloud.shape[0]
    auto = max_candidates is None
    if auto:
        max_candidates = max(4 * n, 8192)
    max_candidates = max(int(max_candidates), n + 1)
    out = torch.empty((n_lines, n, 4), dtype=torch.float32, device=dev)
    counts = torch.empty((n_lines,), dtype=torch.int32, device=dev)
    if n_lines == 0:
        weights = torch.zeros((0, max_candidates), dtype=torch.float32, device=dev) if return_weights else None
        return (out, counts, weights) if return_weights else (out, counts)
    lib = L.lib()
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None and t.numel() > 0 else None
    while True:
        weights = torch.zeros((n_lines, max_candidates), dtype=torch.float32, device=dev) if return_weights else None
        nb = lib.prh_context_workspace_bytes(npts, n_lines, max_candidates)
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        L.check(lib.prh_context_build(p(cloud), npts, p(dense), dense.shape[1], p(line), line.shape[1], n_lines,
                                      float(crop_radius), float(decay_scale), n, max_candidates,
                                      C.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF), p(out), p(counts), p(weights),
                                      p(ws), nb, dev.index, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                "prh_context_build")
        largest = int(counts.max().item())           # one small read-back per scene
        if largest <= max_candidates:
            break
        if not auto:
            import warnings
            warnings.warn(f"build_contexts: a tube holds {largest} points but max_candidates={max_candidates}; "
                          f"its context was drawn from the first {max_candidates} points in cloud order only "
                          "(pass max_candidates=None to size the buffer from the counts)", RuntimeWarning)
            break
        max_candidates = largest                     # true counts are known now: one exact repeat
    return (out, counts, weights) if return_weights else (out, counts)


def build_contexts(cloud, raw_lines, num_line_points=32, num_context_points=1024, crop_radius=0.3,
                   decay_scale=2.0, seed=0, max_candidates=None):
    """raw_lines: sequence of (n_i,3) polylines in scene coordinates.  Returns
    context (L,N,4), noisy_line (L,M,3) centred, centres (L,3), counts (L,) - the model inputs of
    inference_whole_scene.py:98-126 for every line at once."""
    dev = cloud.device
    dense_t = resample_polylines_device(raw_lines, DENSE_POINTS, dev).float()
    line_t = resample_polylines_device(raw_lines, num_line_points, dev).float()
    ctx, counts = build_contexts_resampled(cloud, dense_t, line_t, num_context_points, crop_radius, decay_scale,
                                           seed, max_candidates)
    centres = line_t.mean(dim=1)
    return ctx, line_t - centres[:, None, :], centres, counts
