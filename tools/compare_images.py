"""Compares two renders: raw accumulation dumps (`Raytracing --raw`, magic RTACCUM1), oracle
fixtures (.npz with `accum` and `params`) or PNGs, in any combination.

Raw dumps and fixtures are compared as radiance (accumulation / passes): values outside
|a - b| <= rel * |b| + abs, relative RMSE, mean ratio, NaN pixels on either side, and the same
after the reference's display transform (Reinhard, gamma 2.2: shaders/common/to_ldr.glsl:1-7,
shaders/raytracer/rect.frag:18-23).  PNGs are compared as 8-bit display values only.  If the
sizes differ, both are box-averaged to the smaller grid (a window capture of the reference is a
few pixels smaller than its render).

usage: python tools/compare_images.py A B [--rel 1e-3] [--abs 1e-4] [--max-bad F] [--max-rmse R]
exit status 1 if a limit is given and exceeded.
"""
import argparse
import json
import struct
import sys

import numpy as np


def load(path):
    """Returns (radiance [H, W, 3] float64 or None, display [H, W, 3] in 0..1)."""
    if path.endswith(".npz"):
        z = np.load(path)
        spp = int(z["params"][4])
        radiance = z["accum"][..., :3].astype(np.float64) / spp
    elif path.endswith(".png"):
        from PIL import Image
        return None, np.asarray(Image.open(path).convert("RGB"), dtype=np.float64) / 255.0
    else:
        with open(path, "rb") as f:
            if f.read(8) != b"RTACCUM1":
                raise SystemExit(f"{path}: not a raw accumulation dump")
            w, h, passes = struct.unpack("<III", f.read(12))
            data = np.frombuffer(f.read(), dtype=np.float32).reshape(h, w, 4)
        radiance = data[..., :3].astype(np.float64) / max(passes, 1)
    return radiance, to_display(radiance)


def to_display(radiance):
    c = np.nan_to_num(radiance, nan=0.0, posinf=1e30, neginf=0.0)
    c = c / (c + 1.0)
    return np.clip(c, 0.0, 1.0) ** (1.0 / 2.2)


def box_to(img, h, w):
    if img.shape[:2] == (h, w):
        return img
    ys = (np.arange(h + 1) * img.shape[0] / h).astype(int)
    xs = (np.arange(w + 1) * img.shape[1] / w).astype(int)
    out = np.empty((h, w, img.shape[2]))
    for y in range(h):
        rows = img[ys[y]:max(ys[y + 1], ys[y] + 1)]
        for x in range(w):
            out[y, x] = rows[:, xs[x]:max(xs[x + 1], xs[x] + 1)].mean(axis=(0, 1))
    return out


def compare(a, b, rel, abs_tol):
    nan = np.isnan(a).any(axis=2) | np.isnan(b).any(axis=2)
    keep = ~nan
    a, b = a[keep], b[keep]
    err = np.abs(a - b)
    return {
        "values": int(a.size),
        "nan_pixels": int(nan.sum()),
        "outside_tolerance": float((err > rel * np.abs(b) + abs_tol).mean()) if a.size else 0.0,
        "relative_rmse": float(np.sqrt((err ** 2).mean()) / (np.sqrt((b ** 2).mean()) + 1e-30))
        if a.size else 0.0,
        "max_abs_error": float(err.max()) if a.size else 0.0,
        "mean_ratio": float(a.mean() / b.mean()) if a.size and b.mean() != 0 else float("nan"),
    }


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--rel", type=float, default=1e-3)
    ap.add_argument("--abs", dest="abs_tol", type=float, default=1e-4)
    ap.add_argument("--max-bad", type=float)
    ap.add_argument("--max-rmse", type=float)
    args = ap.parse_args()
    ra, da = load(args.a)
    rb, db = load(args.b)
    h, w = min(da.shape[0], db.shape[0]), min(da.shape[1], db.shape[1])
    report = {"size": [w, h]}
    if ra is not None and rb is not None:
        report["radiance"] = compare(box_to(ra, h, w), box_to(rb, h, w), args.rel, args.abs_tol)
    report["display"] = compare(box_to(da, h, w), box_to(db, h, w), 0.0, 1.0 / 255.0)
    print(json.dumps(report, indent=1))
    judged = report.get("radiance", report["display"])
    failed = ((args.max_bad is not None and judged["outside_tolerance"] > args.max_bad) or
              (args.max_rmse is not None and judged["relative_rmse"] > args.max_rmse))
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
