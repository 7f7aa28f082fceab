"""The synthetic 4K configuration: 10 000 instances of ONE mesh (the loader de-duplicates meshes
by path, src/asset/scene.cpp:72-90, so this is true instancing: one BLAS, 10 000 TRS transforms,
src/asset/scene.cpp:221-246), three area lights and a constant sky, 3840x2160, max_depth 7.

The reference shader cannot render this scene: its TLAS phase collects candidate instances in
a 15-entry list without a bound check (shaders/common/utils.glsl:43-51) and rays here cross far
more than 15 leaf boxes.  The oracle and the HIP path walk instances as they are reached
instead, which gives the same hits as an unbounded list.

(a) At a reduced frame of the SAME scene the HIP path is compared with the oracle: primary-hit
    ids / t / barycentrics and images, both acceleration-structure modes, to the bars of
    tests/test_gpu_parity.py.
(b) At the full 3840x2160 the oracle would take minutes per pass, so size-independent
    properties are checked instead: determinism, exact pass and rectangle decomposition, the
    sample-range split used across GPUs, agreement of the two traversal orders, sane counters.
"""
import numpy as np
import pytest
import torch

import oracle
from raytracing_amd import hip, host
from raytracing_amd.distributed import new_accumulator, pass_range

from test_gpu_parity import ACCELS, assert_image_close, check_ids, make_renderer

pytestmark = pytest.mark.gpu

W, H, TILE, MD = 3840, 2160, (384, 216), 7


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_scene_is_the_stated_configuration(synthetic_4k):
    d = synthetic_4k.desc
    assert d.instance_count == 10_004          # 10 000 spheres + floor + 3 lights
    assert d.mesh_count == 3                   # sphere, floor, light quad: de-duplicated by path
    assert d.triangle_count == 15_872 + 2 + 2
    meshes = synthetic_4k.array("meshes")
    inst = synthetic_4k.array("instances")
    sphere = int(np.argmax(meshes["triangle_count"]))
    assert int(meshes["triangle_count"][sphere]) == 15_872
    assert int((inst["mesh"] == sphere).sum()) == 10_000
    assert d.light_count == 4 and d.sky_light == 3


@pytest.mark.parametrize("accel", ACCELS)
def test_primary_ids_reduced_frame(synthetic_4k, accel):
    check_ids(synthetic_4k, 480, 270, (96, 54), accel)


@pytest.mark.parametrize("accel", ACCELS)
def test_image_parity_reduced_frame(synthetic_4k, accel):
    w, h, tile, spp = 480, 270, (96, 54), 2
    cam = synthetic_4k.camera(w, h)
    seeds = host.seed_schedule(0, spp, 25)
    want, ost = oracle.render(synthetic_4k.desc, cam, w, h, MD, tile, seeds)
    r = make_renderer(synthetic_4k, w, h, MD, tile, accel)
    r.upload_seeds(seeds)
    r.clear()
    r.render(cam, 0, spp)
    got, _ = r.read_accum()
    st = r.stats()
    r.close()
    assert_image_close(got, want, f"synthetic_4k accel {accel}")
    assert st["samples"] == ost["samples"] == w * h * spp
    assert abs(st["closest_rays"] - ost["closest_rays"]) <= 16
    assert abs(st["shadow_rays"] + st["shadow_skipped"] - ost["shadow_rays"]) <= 16


@pytest.fixture(scope="module")
def renderer(synthetic_4k):
    r = hip.HipRenderer(W, H, MD, TILE, accel=hip.ACCEL_REBUILT)
    r.upload_scene(synthetic_4k.desc)
    r.upload_seeds(host.seed_schedule(0, 4, r.tile_count))
    yield r, synthetic_4k.camera(W, H)
    r.close()


def render(r, cam, first, count, rect=None):
    r.clear()
    r.render(cam, first, count, rect=rect)
    return r.read_accum()[0]


def test_full_size_is_deterministic_and_sane(renderer):
    r, cam = renderer
    r.stats(reset=True)
    a = render(r, cam, 0, 4)
    b = render(r, cam, 0, 4)
    assert same(a, b)
    assert np.isnan(a).any(axis=2).mean() < 1e-5
    assert not (np.nan_to_num(a[..., :3]) < 0).any() and (a[..., 3] == 1.0).all()
    mean = np.nan_to_num(a[..., :3]).mean() / 4
    assert 0.02 < mean < 5.0
    st = r.stats(reset=True)
    assert st["samples"] == 2 * W * H * 4
    assert st["samples"] <= st["closest_rays"] <= st["samples"] * (MD + 1)
    assert st["shadow_rays"] <= st["samples"] * MD
    assert st["closest_hits"] <= st["closest_rays"]
    # most of the frame looks at spheres or the floor
    assert st["closest_hits"] > 0.5 * st["samples"]


def test_full_size_pass_split_is_bit_exact(renderer):
    r, cam = renderer
    whole = render(r, cam, 0, 4)
    r.clear()
    r.render(cam, 0, 1)
    r.render(cam, 1, 3)
    split, passes = r.read_accum()
    assert passes == 4
    assert same(split, whole)


def test_full_size_rect_split_is_bit_exact(renderer):
    r, cam = renderer
    whole = render(r, cam, 0, 1)
    r.clear()
    for rect in [(0, 0, 1920, 1080), (1920, 0, 3840, 1080), (0, 1080, 1920, 2160),
                 (1920, 1080, 3840, 2160)]:
        r.render(cam, 0, 1, rect=rect)
    quads, _ = r.read_accum()
    assert same(quads, whole)


def test_full_size_sample_range_split(renderer):
    """What N GPUs do: disjoint pass ranges into private accumulators, then one sum."""
    r, cam = renderer
    whole = render(r, cam, 0, 4)
    stream = torch.cuda.current_stream().cuda_stream
    total = None
    for rank in range(4):
        first, count = pass_range(4, 4, rank)
        acc = new_accumulator(H, W, "cuda")
        r.render(cam, first, count, accum_dev=acc.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        total = acc if total is None else total + acc
    got = total.cpu().numpy()
    assert np.allclose(got[..., :3], whole[..., :3], rtol=2e-6, atol=1e-6, equal_nan=True)


def test_full_size_traversal_orders_agree(synthetic_4k, renderer):
    """Reference-order walk of the uploaded TLAS/BLAS against the rebuilt structure on all
    8.3 M primary rays: identical hits except at exact-distance ties."""
    r, cam = renderer
    i2, t2, d2, _ = r.primary_ids(cam, 0)
    ref = hip.HipRenderer(W, H, MD, TILE, accel=hip.ACCEL_REFERENCE)
    try:
        ref.upload_scene(synthetic_4k.desc)
        ref.upload_seeds(host.seed_schedule(0, 1, ref.tile_count))
        i1, t1, d1, _ = ref.primary_ids(cam, 0)
    finally:
        ref.close()
    assert ((i1 == hip.MISS) == (i2 == hip.MISS)).all()
    differ = (i1 != i2) | (t1 != t2)
    assert differ.mean() < 1e-4
    assert np.allclose(d1[differ], d2[differ], rtol=1e-6)
