"""How the HIP render of a reference scene lines up with the reference's own window screenshot
(tests/golden/ref_*_screenshot.png = outputs/*.png of the reference, the only output of the
reference itself that exists): tries the ways a W x H render can have become the slightly
smaller capture (cropped at some offset, or scaled), prints the fit of each and the per-block
and per-object statistics that tests/test_gpu_reference_screenshots.py asserts.

usage: python tools/screenshot_fit.py hyperion_rect_light|staircase_reduced [spp]
"""
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from raytracing_amd import hip, host, scenes  # noqa: E402

SHOTS = {"hyperion_rect_light": ("ref_hyperion_rect_light_screenshot.png", 1280, 720, 5),
         "staircase_reduced": ("ref_staircase_screenshot.png", 640, 800, 7)}


def display(accum, passes):
    """rect.frag:18-23: gamma_correct(tone_mapping(accum / passes))."""
    c = np.nan_to_num(accum[..., :3]) / passes
    c = c / (c + 1.0)
    return np.clip(c, 0.0, 1.0) ** (1.0 / 2.2)


def render(name, spp):
    shot, w, h, md = SHOTS[name]
    sc = scenes.load_snapshot(name)
    tile = (w // 10, h // 10)
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
    r.upload_scene(sc.desc)
    cam = sc.camera(w, h)
    done = 0
    r.clear()
    while done < spp:
        n = min(64, spp - done)
        r.upload_seeds(host.seed_schedule(done, n, r.tile_count), done)
        r.render(cam, done, n)
        done += n
    accum, passes = r.read_accum()
    r.upload_seeds(host.seed_schedule(0, 1, r.tile_count), 0)
    inst, _, _, _ = r.primary_ids(cam, 0)
    r.close()
    return display(accum, passes), inst


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "hyperion_rect_light"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    shot = np.asarray(Image.open(os.path.join(ROOT, "tests", "golden", SHOTS[name][0]))
                      .convert("RGB"), dtype=np.float32) / 255.0
    mine, inst = render(name, spp)
    sh, sw, _ = shot.shape
    h, w, _ = mine.shape
    print(f"{name}: render {w}x{h} at {spp} spp, screenshot {sw}x{sh}")
    best = None
    for dy in range(h - sh + 1):
        for dx in range(w - sw + 1):
            mse = float(((mine[dy:dy + sh, dx:dx + sw] - shot) ** 2).mean())
            if best is None or mse < best[0]:
                best = (mse, dx, dy)
    print("crop: best offset dx=%d dy=%d  rmse %.4f" % (best[1], best[2], best[0] ** 0.5))
    scaled = np.asarray(Image.fromarray((mine * 255 + 0.5).astype(np.uint8))
                        .resize((sw, sh), Image.BILINEAR), dtype=np.float32) / 255.0
    print("scale: rmse %.4f" % float(((scaled - shot) ** 2).mean()) ** 0.5)
    srgb = np.where(mine <= 0.0031308, mine * 12.92, 1.055 * mine ** (1 / 2.4) - 0.055)
    _, dx, dy = best
    crop = mine[dy:dy + sh, dx:dx + sw]
    print("crop + sRGB encode on top: rmse %.4f"
          % float(((srgb[dy:dy + sh, dx:dx + sw] - shot) ** 2).mean()) ** 0.5)
    for label, a in (("crop", crop), ("scale", scaled)):
        bh, bw = sh // 32, sw // 32
        ba = a[:bh * 32, :bw * 32].reshape(bh, 32, bw, 32, 3).mean(axis=(1, 3))
        bs = shot[:bh * 32, :bw * 32].reshape(bh, 32, bw, 32, 3).mean(axis=(1, 3))
        err = np.abs(ba - bs)
        print(f"{label}: 32x32 blocks: mean |err| {err.mean():.4f}, 95th pct "
              f"{np.percentile(err, 95):.4f}, 99th {np.percentile(err, 99):.4f}, max {err.max():.4f}; "
              f"mean render {ba.mean():.4f} shot {bs.mean():.4f}; per channel "
              f"{ba.mean(axis=(0, 1))} vs {bs.mean(axis=(0, 1))}")
    ids = inst[dy:dy + sh, dx:dx + sw]
    print("per object (instance id, pixels, render rgb, screenshot rgb):")
    for i in np.unique(ids):
        m = ids == i
        if m.sum() >= 2000:
            print("  %10d %8d  %s  %s" % (i, m.sum(), np.round(crop[m].mean(axis=0), 3),
                                          np.round(shot[m].mean(axis=0), 3)))
    # local noise of the screenshot itself: high-pass residual in 4x4 cells
    cells = shot[:sh // 4 * 4, :sw // 4 * 4].reshape(sh // 4, 4, sw // 4, 4, 3)
    print("screenshot within-4x4-cell std (median over cells): %.4f; render: %.4f"
          % (float(np.median(cells.std(axis=(1, 3)))),
             float(np.median(crop[:sh // 4 * 4, :sw // 4 * 4].reshape(sh // 4, 4, sw // 4, 4, 3)
                             .std(axis=(1, 3))))))
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    Image.fromarray((crop * 255 + 0.5).astype(np.uint8)).save(
        os.path.join(out, f"fit_{name}_render.png"))
    diff = np.clip(0.5 + 2.0 * (crop - shot), 0, 1)
    Image.fromarray((diff * 255 + 0.5).astype(np.uint8)).save(
        os.path.join(out, f"fit_{name}_diff.png"))


if __name__ == "__main__":
    main()
