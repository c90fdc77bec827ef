"""Pins pointnet_refine_amd/io.py::load_pcd_data against the reference's loader on synthetic PCD
files of the three formats it understands (+ one it rejects) and writes the byte streams and the
reference's outputs to tests/golden/g8_pcd.npz.  Build container only.
  python oracle/make_golden_io.py"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True
from src.dataset import load_pcd_data as ref_load        # noqa: E402  (the reference)

from pointnet_refine_amd.io import load_pcd_data          # noqa: E402


def header(n, fields, sizes, types, data):
    return (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS {fields}\nSIZE {sizes}\n"
            f"TYPE {types}\nCOUNT {' '.join('1' for _ in fields.split())}\nWIDTH {n}\nHEIGHT 1\n"
            f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {data}\n").encode()


def main():
    rng = np.random.default_rng(5)
    n = 257
    xyz = rng.normal(0, 20, (n, 3)).astype(np.float32)
    inten = np.clip(np.round(rng.exponential(12, n)), 0, 255).astype(np.float32)
    files = {}
    ascii_body = "\n".join(" ".join(f"{v:.6f}" for v in (*xyz[i], inten[i])) for i in range(n)) + "\n"
    files["ascii"] = header(n, "x y z intensity", "4 4 4 4", "F F F F", "ascii") + ascii_body.encode()
    rec = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4")])
    rec["x"], rec["y"], rec["z"], rec["intensity"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], inten
    files["binary16"] = header(n, "x y z intensity", "4 4 4 4", "F F F F", "binary") + rec.tobytes()
    rec2 = np.zeros(n, dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<u2")])
    rec2["x"], rec2["y"], rec2["z"], rec2["intensity"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], inten.astype(np.uint16)
    files["binary14"] = header(n, "x y z intensity", "4 4 4 2", "F F F U", "binary") + rec2.tobytes()
    files["binary_unknown"] = header(n, "x y z", "4 4 4", "F F F", "binary") + xyz.tobytes()
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for k, blob in files.items():
            path = os.path.join(d, k + ".pcd")
            open(path, "wb").write(blob)
            want, got = ref_load(path), load_pcd_data(path)
            assert want.shape == got.shape and want.dtype == got.dtype and np.array_equal(want, got), k
            out[k + "_bytes"] = np.frombuffer(blob, dtype=np.uint8)
            out[k + "_points"] = want
            print(k, want.shape, want.dtype, "product == reference")
        missing = os.path.join(d, "missing.pcd")
        assert ref_load(missing).shape == load_pcd_data(missing).shape == (0, 4)
    path = os.path.join(ROOT, "tests", "golden", "g8_pcd.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
