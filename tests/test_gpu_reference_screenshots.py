"""The HIP path against the reference implementation's own output: its window screenshots.

The scene files are rendered as shipped (resolution and max_depth of the JSON) at 1024 passes
through the C ABI, put through the reference's display transform (rect.frag:18-23), registered
against the capture (tests/screenshots.py) and compared at three scales:

  * means of 32 x 32 pixel blocks, every block of the picture;
  * mean colour of every object (primary-hit instance) that covers enough pixels;
  * mean colour of the whole picture, per channel.

Where the bars come from.  The capture is 8-bit: one step is 1/255 = 0.0039 in display units.
At 1024 passes a pixel of this render has a standard deviation of about 0.005 display units
(measured in flat regions; the capture's own pixels show the same), so the mean of a block of
1024 pixels carries about 0.005 / 32 = 1.6e-4 of noise from either side plus 1/255 / sqrt(12) /
32 = 3.5e-5 of rounding: block means of two correct renders differ by a few 1e-4, except in
blocks that an object's silhouette crosses, where the unknown sub-pixel jitter history of the
capture matters.  Measured on MI355X for hyperion_rect_light: mean |difference| 4e-4, 99th
percentile 2.5e-3, largest 5.3e-3; objects agree to 0.002, the whole picture to 3e-4 per
channel.  The bars below are 2-3x those figures and all below two 8-bit steps.

The staircase capture shows the brass chandelier and light socket whose mesh the reference
repository does not ship (staircase/Brass.obj), so this build renders the room without them:
the blocks they cover are masked (the capture's pixels there differ from the brass-free render
by more than 0.1), and the room is about 2 % brighter than in the capture because nothing
shadows the lamp; the bars of that scene allow for that and say so.
"""
import numpy as np
import pytest

from raytracing_amd import hip, host, scenes

import screenshots as S

pytestmark = pytest.mark.gpu

PASSES = 1024


_rendered = {}


def render_as_shipped(scene):
    if scene not in _rendered:
        _rendered[scene] = render_as_shipped_now(scene)
    return _rendered[scene]


def render_as_shipped_now(scene):
    _, w, h, md = S.SHOTS[scene]
    sc = scenes.load_snapshot(scene)
    assert (sc.info.resolution_x, sc.info.resolution_y, sc.info.max_depth) == (w, h, md)
    tile = (sc.info.tile_width, sc.info.tile_height)
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
    try:
        r.upload_scene(sc.desc)
        cam = sc.camera(w, h)
        r.clear()
        for first in range(0, PASSES, 64):
            r.upload_seeds(host.seed_schedule(first, 64, r.tile_count), first)
            r.render(cam, first, 64)
        accum, passes = r.read_accum()
        r.upload_seeds(host.seed_schedule(0, 1, r.tile_count), 0)
        ids = r.primary_ids(cam, 0)[0]
    finally:
        r.close()
    assert passes == PASSES
    return S.display(accum, passes), ids


def test_hyperion_rect_light_matches_the_reference_screenshot():
    shot = S.load_screenshot("hyperion_rect_light")
    mine, ids = render_as_shipped("hyperion_rect_light")
    dx, dy, rmse = S.register(mine, shot)
    sh, sw, _ = shot.shape
    crop, ids = mine[dy:dy + sh, dx:dx + sw], ids[dy:dy + sh, dx:dx + sw]
    # per pixel: two noisy 1024-pass renders, one of them rounded to 8 bits
    assert rmse < 0.03, rmse
    err = np.abs(S.block_means(crop) - S.block_means(shot))
    assert err.shape[:2] == (22, 39)
    assert err.mean() < 1.0e-3, err.mean()
    assert np.percentile(err, 99) < 5.0e-3, np.percentile(err, 99)
    assert err.max() < 1.0e-2, err.max()
    whole = np.abs(crop.mean(axis=(0, 1)) - shot.mean(axis=(0, 1)))
    assert (whole < 1.0e-3).all(), whole
    a, b = S.object_colours(crop, ids, 2000), S.object_colours(shot, ids, 2000)
    assert len(a) >= 20   # floor, back wall, the spheres and rings, 17 light panels
    for i in a:
        assert np.abs(a[i] - b[i]).max() < 4.0e-3, (i, a[i], b[i])


def test_staircase_matches_the_reference_screenshot():
    shot = S.load_screenshot("staircase_reduced")
    mine, ids = render_as_shipped("staircase_reduced")
    dx, dy, rmse = S.register(mine, shot)
    sh, sw, _ = shot.shape
    crop, ids = mine[dy:dy + sh, dx:dx + sw], ids[dy:dy + sh, dx:dx + sw]
    assert rmse < 0.06, rmse
    a_blocks, b_blocks = S.block_means(crop), S.block_means(shot)
    err = np.abs(a_blocks - b_blocks).max(axis=2)
    brass = err > 0.05          # the chandelier and the socket, absent from this render
    assert brass.mean() < 0.03, brass.mean()
    assert err[~brass].mean() < 0.015, err[~brass].mean()
    assert np.percentile(err[~brass], 99) < 0.04
    # brightness: no brass in front of the lamp -> at most 4 % brighter, never darker
    ratio = crop.mean(axis=(0, 1)) / shot.mean(axis=(0, 1))
    assert (ratio > 0.995).all() and (ratio < 1.04).all(), ratio
    # colour balance of the large textured surfaces (wallpaper, wood, painting): the ratio
    # between channels is independent of the brightness offset
    a, b = S.object_colours(crop, ids, 2000), S.object_colours(shot, ids, 2000)
    assert len(a) >= 8
    for i in a:
        assert np.abs(a[i] / a[i][0] - b[i] / b[i][0]).max() < 0.02, (i, a[i], b[i])
        assert np.abs(a[i] - b[i]).max() < 0.025, (i, a[i], b[i])


def test_hip_render_matches_the_converged_oracle_fixture():
    """The same 1024 passes on the same seeds, once by the CPU oracle (the committed fixture,
    tools/make_converged_fixture.py; tests/test_oracle_golden.py holds it to the reference's
    screenshot and re-renders a strip of it) and once by the HIP path: the two differ only
    where a transcendental's last place flips a path, so block means, object colours and the
    picture mean agree far below the capture's quantisation.  Measured on MI355X: block means
    differ by 1.8e-7 on average and 3.6e-5 at most, objects by 3e-7, the picture by 9e-8; the
    bars are ten times that."""
    import os
    z = np.load(os.path.join(S.GOLDEN, "oracle_hyperion_converged.npz"))
    w, h, md, passes, dx, dy = (int(v) for v in z["params"])
    assert passes == PASSES
    mine, ids = render_as_shipped("hyperion_rect_light")
    shot = S.load_screenshot("hyperion_rect_light")
    sh, sw, _ = shot.shape
    crop, ids = mine[dy:dy + sh, dx:dx + sw], ids[dy:dy + sh, dx:dx + sw]
    err = np.abs(S.block_means(crop) - z["display_blocks"])
    assert err.mean() < 2.0e-6, err.mean()
    assert err.max() < 4.0e-4, err.max()
    assert (np.abs(crop.mean(axis=(0, 1)) - z["whole_display"]) < 1.0e-6).all()
    objects = S.object_colours(crop, ids, 2000)
    assert sorted(objects) == [int(i) for i in z["object_ids"]]
    for i, theirs in zip(z["object_ids"], z["object_display"]):
        assert np.abs(objects[int(i)] - theirs).max() < 3.0e-6, (i, objects[int(i)], theirs)
