"""CPU restatement of the reference's per-line context builder - TEST
INFRASTRUCTURE ONLY: imported by tests/, never by the product path.

Follows src/dataset.py of the reference:
  arc_resample      : resample_polyline, src/dataset.py:8-29
  crop_mask         : distance crop against the polyline densified to 200 points,
                      src/dataset.py:210-222 (= inference_whole_scene.py:104-110)
  sampling_weights  : distance x intensity weights, src/dataset.py:93-122
  build_context     : the three sampling regimes + centring, src/dataset.py:86-91,124-130,229-234
Pinned by oracle/make_golden_context.py, which runs the reference's own functions on the same
inputs (crop mask via its KDTree query, weights via the sample it draws under a fixed numpy seed).
"""
import numpy as np


def arc_resample(points, num_points):
    points = np.asarray(points, dtype=np.float64)
    if len(points) < 2:                                   # :13-14
        return np.zeros((num_points, 3))
    seg = np.linalg.norm(points[1:] - points[:-1], axis=1)
    cum = np.concatenate(([0.0], np.cumsum(seg)))         # :17-18
    t = np.linspace(0.0, cum[-1], num_points)             # :22
    return np.stack([np.interp(t, cum, points[:, i]) for i in range(3)], axis=1)   # :25-27


def nearest_distance(xyz, line_pts):
    """Euclidean distance of every row of xyz to the nearest of line_pts (what KDTree.query
    returns), float64, brute force in blocks."""
    xyz = np.asarray(xyz, dtype=np.float64)
    line_pts = np.asarray(line_pts, dtype=np.float64)
    out = np.empty(len(xyz))
    for s in range(0, len(xyz), 8192):
        d = xyz[s:s + 8192, None, :] - line_pts[None, :, :]
        out[s:s + 8192] = np.sqrt((d * d).sum(-1).min(1))
    return out


def crop_mask(cloud, dense_line, radius):
    return nearest_distance(cloud[:, :3], dense_line) < radius          # :218-221 (strict <)


def sampling_weights(cands, line_pts, decay_scale):
    """Unnormalised weights of src/dataset.py:93-111 for the cropped candidates (K,4)."""
    d = nearest_distance(cands[:, :3], line_pts)                          # :94-95
    dist_w = np.exp(-d / decay_scale)                                     # :99
    inten = cands[:, 3].astype(np.float64) if cands.dtype == np.float64 else cands[:, 3]
    lo, hi = np.min(inten), np.max(inten)
    if hi > lo:
        norm = (inten - lo) / (hi - lo + 1e-6)                            # :106-107
    else:
        norm = np.ones_like(inten) * 0.5                                  # :108-109
    return dist_w * (0.5 + norm)                                          # :110,113


def build_context(cloud, dense_line, line_pts, radius, decay_scale, num_samples, rng=np.random):
    """(N,4) context block of one line: xyz centred on the line's mean, raw intensity.
    Sampling uses numpy's generator exactly as the reference does (distributional parity for the
    GPU path, exact for this function against the reference under the same seed)."""
    cands = cloud[crop_mask(cloud, dense_line, radius)] if len(cloud) and len(line_pts) else np.zeros((0, 4))
    k = len(cands)
    if k <= num_samples:
        if k == 0:
            picked = np.zeros((num_samples, 4))                          # :87-88
        else:
            picked = cands[rng.choice(k, num_samples, replace=True)]     # :90-91
    else:
        w = sampling_weights(cands, line_pts, decay_scale)
        s = w.sum()
        p = None if s < 1e-6 else (w / s) / np.sum(w / s)                 # :115-126
        picked = cands[rng.choice(k, num_samples, replace=False, p=p)]   # :129-130
    centre = np.mean(line_pts, axis=0)                                    # :231
    return np.hstack([picked[:, :3] - centre, picked[:, 3:4]]), k
