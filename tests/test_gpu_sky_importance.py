"""RT_HIP_FLAG_SKY_IMPORTANCE on the device against the oracle's implementation of the same
opt-in: single draws and densities (rt_hip_probe), then images.  Without the flag the renderer
is the reference's sampling, bit for bit (every other GPU test runs without it)."""
import numpy as np
import pytest

import oracle
from raytracing_amd import hip, host

from test_gpu_parity import ACCELS, assert_image_close

pytestmark = pytest.mark.gpu


def renderer(scene, w, h, md, tile, accel=hip.ACCEL_REBUILT, flags=hip.FLAG_SKY_IMPORTANCE):
    r = hip.HipRenderer(w, h, md, tile, accel=accel, flags=flags)
    r.upload_scene(scene.desc)
    return r


@pytest.mark.parametrize("scene_name", ["textured", "mustang"])
def test_draws_and_densities_match_the_oracle(request, scene_name):
    scene = request.getfixturevalue(scene_name)
    r = renderer(scene, 64, 64, 3, (32, 32))
    try:
        seeds = [(0x1000 + 7919 * k) & 0xFFFFFFFF for k in range(200)]
        u, f = r.probe(hip.PROBE_SKY_SAMPLE, [([s], [], None) for s in seeds])
        want = [oracle.sample_sky_importance(scene.desc, s) for s in seeds]
        want_d = np.stack([w[0] for w in want])
        assert (u[:, 0] == np.array([w[1] for w in want], dtype=np.uint32)).all()
        # sin / cos of the two angles may differ in the last place
        assert np.abs(f[:, :3] - want_d).max() < 2e-6
        pdf = oracle.eval_sky_light_many(scene.desc, f[:, :3], importance=True)[:, 3]
        # the density of the drawn direction, looked up again from the direction: an ulp of
        # atan2 / acos can move a direction that lies on a texel border into the next texel
        close = np.isclose(f[:, 3], pdf, rtol=1e-4, atol=0)
        assert close.mean() > 0.99
        rng = np.random.RandomState(2)
        dirs = rng.normal(size=(300, 3))
        dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
        _, g = r.probe(hip.PROBE_SKY, [([], list(d), None) for d in dirs])
        want = oracle.eval_sky_light_many(scene.desc, dirs, importance=True)
        assert np.allclose(g[:, :3], want[:, :3], rtol=5e-4, atol=1e-6)
        assert np.isclose(g[:, 3], want[:, 3], rtol=1e-4, atol=0).mean() > 0.99
    finally:
        r.close()
    plain = hip.HipRenderer(64, 64, 3, (32, 32))
    try:
        plain.upload_scene(scene.desc)
        with pytest.raises(RuntimeError, match="no sky distribution"):
            plain.probe(hip.PROBE_SKY_SAMPLE, [([1], [], None)])
    finally:
        plain.close()


@pytest.mark.parametrize("accel", ACCELS)
def test_image_parity_with_the_flag(textured, accel):
    info = textured.info
    w, h, tile, md, spp = info.resolution_x, info.resolution_y, (40, 30), info.max_depth, 8
    cam = textured.camera(w, h)
    seeds = host.seed_schedule(0, spp, 16)
    want, ost = oracle.render(textured.desc, cam, w, h, md, tile, seeds, sky_importance=True)
    base, _ = oracle.render(textured.desc, cam, w, h, md, tile, seeds)
    assert not np.array_equal(want, base)
    r = renderer(textured, w, h, md, tile, accel)
    try:
        r.upload_seeds(seeds)
        r.clear()
        r.render(cam, 0, spp)
        got, _ = r.read_accum()
        st = r.stats()
    finally:
        r.close()
    assert_image_close(got, want, f"sky importance accel {accel}")
    assert abs(st["closest_rays"] - ost["closest_rays"]) <= 16
    assert abs(st["shadow_rays"] + st["shadow_skipped"] - ost["shadow_rays"]) <= 16


def test_pipeline_equals_fused_kernel_with_the_flag(mustang):
    w, h, md, tile, spp = 320, 184, 7, (64, 46), 4
    cam = mustang.camera(w, h)
    images = []
    for extra in (hip.FLAG_FUSED, 0):
        r = renderer(mustang, w, h, md, tile, flags=hip.FLAG_SKY_IMPORTANCE | extra)
        try:
            r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
            r.clear()
            r.render(cam, 0, spp)
            images.append(r.read_accum()[0])
        finally:
            r.close()
    assert np.array_equal(images[0], images[1], equal_nan=True)
    want, _ = oracle.render(mustang.desc, cam, w, h, md, tile,
                            host.seed_schedule(0, spp, 20), sky_importance=True)
    assert_image_close(images[1], want, "mustang sky importance")


def test_noise_at_equal_sample_count(mustang):
    """What the opt-in buys on the reduced mustang scene at 1280x720: relative noise (RMS
    difference of two independent 32-pass renders over the mean) with and without it.  The
    emitted radiance is the Reinhard-mapped texel (at most `intensity`), so the map's dynamic
    range is compressed and the gain is modest; the test requires that the opt-in is not
    noisier, and prints both figures."""
    w, h, md, tile, spp = 1280, 720, 7, (128, 72), 32
    cam = mustang.camera(w, h)
    noise = {}
    for flags in (0, hip.FLAG_SKY_IMPORTANCE):
        r = renderer(mustang, w, h, md, tile, flags=flags)
        try:
            r.upload_seeds(host.seed_schedule(0, 2 * spp, r.tile_count))
            halves = []
            for first in (0, spp):
                r.clear()
                r.render(cam, first, spp)
                halves.append(np.nan_to_num(r.read_accum()[0][..., :3]) / spp)
        finally:
            r.close()
        a, b = halves
        noise[flags] = float(np.sqrt(((a - b) ** 2).mean()) / (0.5 * (a + b)).mean())
    print(f"relative noise at {spp} passes: reference sampling {noise[0]:.4f}, "
          f"importance sampling {noise[hip.FLAG_SKY_IMPORTANCE]:.4f}")
    assert noise[hip.FLAG_SKY_IMPORTANCE] < 1.05 * noise[0]
