"""Comparison of a render with the reference's own window screenshots (helpers for the tests).

tests/golden/ref_hyperion_rect_light_screenshot.png and ref_staircase_screenshot.png are
outputs/hyperion_rect_light.png and outputs/staircase.png of the reference repository,
unchanged: captures of its window after an unknown number of passes with unknown seeds.  They
are the only output of the reference implementation itself that exists, so they are what pins
this build's results to the reference rather than to a restatement of it.

A capture is a few pixels smaller than the render (1274x712 of 1280x720, 635x794 of 640x800):
the window's frame was cut off, so the capture is a CROP of the render at an unknown offset of
at most (W - w, H - h) pixels - not a scaled copy (a scaled copy fits ten times worse).  The
offset is found by exhaustive search.
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SHOTS = {
    # scene snapshot -> (file, render width, height, max_depth of the scene file)
    "hyperion_rect_light": ("ref_hyperion_rect_light_screenshot.png", 1280, 720, 5),
    "staircase_reduced": ("ref_staircase_screenshot.png", 640, 800, 7),
}


def load_screenshot(scene):
    from PIL import Image
    path = os.path.join(GOLDEN, SHOTS[scene][0])
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0


def display(accum, passes):
    """What the reference puts on screen, shaders/raytracer/rect.frag:18-23 with
    frame_scalar = 1 / accumulation_counter (megakernel_raytracer.cpp:748-753):
    gamma_correct(tone_mapping(accum / passes)), shaders/common/to_ldr.glsl:1-7."""
    c = np.nan_to_num(accum[..., :3].astype(np.float64)) / passes
    c = c / (c + 1.0)
    return (np.clip(c, 0.0, 1.0) ** (1.0 / 2.2)).astype(np.float32)


def register(mine, shot):
    """Offset (dx, dy) at which the capture lies in the render, by least squares."""
    h, w, _ = mine.shape
    sh, sw, _ = shot.shape
    best = None
    for dy in range(h - sh + 1):
        for dx in range(w - sw + 1):
            mse = float(((mine[dy:dy + sh, dx:dx + sw] - shot) ** 2).mean())
            if best is None or mse < best[0]:
                best = (mse, dx, dy)
    return best[1], best[2], best[0] ** 0.5


def block_means(img, size=32):
    h, w, c = img.shape
    bh, bw = h // size, w // size
    return img[:bh * size, :bw * size].reshape(bh, size, bw, size, c).mean(axis=(1, 3))


def object_colours(img, ids, min_pixels):
    """Mean colour of every object (primary-hit instance id) that covers at least min_pixels."""
    out = {}
    for i in np.unique(ids):
        m = ids == i
        if int(m.sum()) >= min_pixels:
            out[int(i)] = img[m].mean(axis=0)
    return out
