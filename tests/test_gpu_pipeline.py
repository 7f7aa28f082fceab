"""rt_hip_render runs as a pipeline of trace / shade launches per bounce (queues in device
memory, LDS-cached traversal, a finishing kernel for the late bounces).  The single fused
kernel (RT_HIP_FLAG_FUSED, also what rt_hip_render_tile uses) computes every path with the
same device functions in the same order, so the two must agree bit for bit - images and work
counters - in both acceleration-structure modes, across batch boundaries, for deep paths, and
with the ray queues regrouped by origin cell and direction (ray binning) or not."""
import numpy as np
import pytest

from raytracing_amd import hip, host

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def run(scene, w, h, md, tile, spp, accel, flags, rect=None):
    r = hip.HipRenderer(w, h, md, tile, accel=accel, flags=flags)
    try:
        r.upload_scene(scene.desc)
        r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
        r.clear()
        r.render(scene.camera(w, h), 0, spp, rect=rect)
        img, passes = r.read_accum()
        st = r.stats()
        return img, passes, st
    finally:
        r.close()


COUNTERS = ["samples", "closest_rays", "shadow_rays", "closest_hits", "shadow_skipped",
            "nodes_visited", "tris_tested", "max_stack"]


BINNING = [hip.FLAG_NO_BIN_RAYS, hip.FLAG_BIN_RAYS]


@pytest.mark.parametrize("binning", BINNING)
@pytest.mark.parametrize("accel", [hip.ACCEL_REFERENCE, hip.ACCEL_REBUILT])
@pytest.mark.parametrize("scene_name", ["cornell", "hyperion", "staircase"])
def test_pipeline_equals_fused_kernel(request, scene_name, accel, binning):
    if accel == hip.ACCEL_REFERENCE and binning == hip.FLAG_BIN_RAYS:
        pytest.skip("ray binning belongs to the rebuilt acceleration structure's pipeline")
    scene = request.getfixturevalue(scene_name)
    w, h, md, tile, spp = 320, 184, 7, (64, 46), 5   # 184 = 23 * 8; tiles of 64 x 46
    fused = run(scene, w, h, md, tile, spp, accel, hip.FLAG_FUSED | hip.FLAG_COUNT_NODES)
    piped = run(scene, w, h, md, tile, spp, accel, hip.FLAG_COUNT_NODES | binning)
    assert fused[1] == piped[1] == spp
    assert same(fused[0], piped[0])
    for k in COUNTERS:
        assert fused[2][k] == piped[2][k], k


@pytest.mark.parametrize("binning", BINNING)
def test_pipeline_partial_blocks(cornell, binning):
    """A rectangle whose sides are not multiples of the 8x8 path blocks: the padding lanes of
    the edge blocks start no path and touch no pixel."""
    w, h, md, tile, spp = 320, 180, 5, (64, 36), 3
    rect = (5, 3, 299, 171)
    fused = run(cornell, w, h, md, tile, spp, hip.ACCEL_REBUILT, hip.FLAG_FUSED, rect)
    piped = run(cornell, w, h, md, tile, spp, hip.ACCEL_REBUILT, binning, rect)
    assert same(fused[0], piped[0])
    assert piped[2]["samples"] == (299 - 5) * (171 - 3) * spp
    assert (piped[0][:3, :, :3] == 0).all() and (piped[0][:, :5, :3] == 0).all()


def test_deep_paths_through_the_finishing_kernel(cornell):
    """max_depth 24 on a small frame (61 440 paths, below the hand-over threshold): after
    bounce 0 the finishing kernel runs the paths to their end, up to 24 more bounces."""
    w, h, tile, spp = 160, 96, (32, 24), 4
    fused = run(cornell, w, h, 24, tile, spp, hip.ACCEL_REBUILT, hip.FLAG_FUSED)
    piped = run(cornell, w, h, 24, tile, spp, hip.ACCEL_REBUILT, 0)
    assert same(fused[0], piped[0])
    for k in ["samples", "closest_rays", "shadow_rays", "closest_hits", "shadow_skipped"]:
        assert fused[2][k] == piped[2][k], k
    assert piped[2]["closest_rays"] > 3.5 * piped[2]["samples"]  # the paths really are long


def test_shallow_depths(cornell):
    """max_depth 0 and 1: fewer bounces than the pipeline has stages."""
    for md in (0, 1, 2):
        fused = run(cornell, 160, 96, md, (32, 24), 2, hip.ACCEL_REBUILT, hip.FLAG_FUSED)
        piped = run(cornell, 160, 96, md, (32, 24), 2, hip.ACCEL_REBUILT, 0)
        assert same(fused[0], piped[0]), md
        assert fused[2]["closest_rays"] == piped[2]["closest_rays"], md
        assert fused[2]["shadow_rays"] == piped[2]["shadow_rays"], md


@pytest.mark.parametrize("binning", BINNING)
def test_batches_of_passes(hyperion, binning):
    """1280x720 x 80 passes = 73.7 M paths: more than one batch of 2^26 paths.  One call must
    equal the same passes given in two calls (which also splits the batches elsewhere)."""
    w, h, md, tile, spp = 1280, 720, 3, (128, 72), 80
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT, flags=binning)
    try:
        r.upload_scene(hyperion.desc)
        r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
        cam = hyperion.camera(w, h)
        r.clear()
        r.render(cam, 0, spp)
        one, passes = r.read_accum()
        assert passes == spp
        assert r.stats(reset=True)["samples"] == w * h * spp
        r.clear()
        r.render(cam, 0, 37)
        r.render(cam, 37, 43)
        two, _ = r.read_accum()
        assert same(one, two)
    finally:
        r.close()


def test_ray_binning_is_decided_by_measurement(cornell, staircase):
    """Without a forcing flag the first large render call of a scene runs unbinned, in fixed
    chunks, and times its trace launches; the second refills; after it the bounces that fixed
    chunks won and whose rays are expensive are binned.  Whatever it decides (that depends on the clock and on who else uses the GPU,
    so the outcome itself is not asserted here: tools/scene_table.py prints it), the switch
    does not change a bit of the image."""
    w, h, md, tile, spp = 640, 800, 7, (64, 80), 16
    outcome = {}
    for name, scene in (("cornell", cornell), ("staircase", staircase)):
        r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
        try:
            r.upload_scene(scene.desc)
            r.upload_seeds(host.seed_schedule(0, 2 * spp, r.tile_count))
            cam = scene.camera(w, h)
            assert r.ray_binning() == (hip.BINNING_UNDECIDED, 0.0)
            r.clear()
            r.render(cam, 0, spp)          # measures
            r.read_accum()                 # (synchronises)
            r.render(cam, spp, spp)        # measures the other way
            auto, passes = r.read_accum()
            state, ns = r.ray_binning()
            assert passes == 2 * spp and state != hip.BINNING_UNDECIDED and ns > 0.0
            outcome[name] = (state, ns)
            # a new scene is measured anew
            r.upload_scene(scene.desc)
            assert r.ray_binning() == (hip.BINNING_UNDECIDED, 0.0)
        finally:
            r.close()
        r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT, flags=hip.FLAG_NO_BIN_RAYS)
        try:
            r.upload_scene(scene.desc)
            r.upload_seeds(host.seed_schedule(0, 2 * spp, r.tile_count))
            assert r.ray_binning() == (hip.BINNING_OFF, 0.0)
            r.clear()
            r.render(scene.camera(w, h), 0, 2 * spp)
            plain, _ = r.read_accum()
        finally:
            r.close()
        assert same(auto, plain), name
    assert all(state in (hip.BINNING_ON, hip.BINNING_OFF) for state, _ in outcome.values())


REFILL = [hip.FLAG_TRACE_CHUNKS, hip.FLAG_TRACE_REFILL]


@pytest.mark.parametrize("refill", REFILL)
@pytest.mark.parametrize("binning", BINNING)
def test_trace_launch_hands_out_rays_either_way(staircase, binning, refill):
    """The trace launch gives a wave fixed chunks of 64 rays, or gives each lane that has
    finished its ray the next one.  Either way (forced here by flag) every ray is traced
    exactly once: image and work counters equal the fused kernel's."""
    w, h, md, tile, spp = 320, 184, 7, (64, 46), 5
    fused = run(staircase, w, h, md, tile, spp, hip.ACCEL_REBUILT,
                hip.FLAG_FUSED | hip.FLAG_COUNT_NODES)
    piped = run(staircase, w, h, md, tile, spp, hip.ACCEL_REBUILT,
                hip.FLAG_COUNT_NODES | binning | refill)
    assert same(fused[0], piped[0])
    for k in COUNTERS:
        assert fused[2][k] == piped[2][k], k


def test_trace_refill_is_decided_by_measurement(hyperion):
    """The first large call of a scene is timed with fixed chunks, the second with refilled
    lanes; the faster way is kept bounce by bounce.  The outcome depends on the clock and is not
    asserted; the image must not depend on it."""
    w, h, md, tile, spp = 640, 800, 7, (64, 80), 16
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
    try:
        r.upload_scene(hyperion.desc)
        r.upload_seeds(host.seed_schedule(0, 4 * spp, r.tile_count))
        cam = hyperion.camera(w, h)
        assert r.trace_refill() == (hip.REFILL_UNDECIDED, -1.0, -1.0, 0xFFFFFFFF)
        r.clear()
        seen = []
        for call in range(4):
            r.render(cam, call * spp, spp)
            r.read_accum()
            seen.append(r.trace_refill())
        auto, passes = r.read_accum()
        assert passes == 4 * spp
        # call 0 is the base line (fixed chunks), call 1 refills; then both policies are settled
        assert seen[0][0] == hip.REFILL_UNDECIDED and seen[0][1] > 0.0 and seen[0][2] < 0.0
        state, off, on, mask = seen[1]
        assert state in (hip.REFILL_OFF, hip.REFILL_ON) and off > 0.0 and on > 0.0
        assert (mask != 0) == (state == hip.REFILL_ON)
        assert r.ray_binning()[0] != hip.BINNING_UNDECIDED
        assert seen[2] == seen[1]
        assert seen[3] == seen[2]
        r.upload_scene(hyperion.desc)   # a new scene is measured anew
        assert r.trace_refill()[0] == hip.REFILL_UNDECIDED
    finally:
        r.close()
    fused = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT, flags=hip.FLAG_FUSED)
    try:
        fused.upload_scene(hyperion.desc)
        fused.upload_seeds(host.seed_schedule(0, 4 * spp, fused.tile_count))
        fused.clear()
        fused.render(hyperion.camera(w, h), 0, 4 * spp)
        want, _ = fused.read_accum()
    finally:
        fused.close()
    assert same(auto, want)


@pytest.mark.parametrize("refill", REFILL)
def test_trace_work_counters(hyperion, refill):
    """rt_hip_trace_work: the fetch counts repeat the statistics, and the wave-step counters
    bound them - a wave step serves between 1 and 64 lanes."""
    w, h, md, tile, spp = 640, 360, 7, (64, 36), 4
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT,
                        flags=hip.FLAG_COUNT_NODES | refill)
    try:
        r.upload_scene(hyperion.desc)
        r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
        r.clear()
        r.render(hyperion.camera(w, h), 0, spp)
        st = r.stats()
        work = r.trace_work()
    finally:
        r.close()
    assert work["nodes"] == st["nodes_visited"] and work["tris"] == st["tris_tested"]
    assert work["nodes_from_lds"] <= work["nodes"]
    assert work["instance_records_from_lds"] <= work["instance_records"]
    for steps, wave_steps in (
            (work["nodes"] - work["finish_nodes"], work["node_wave_steps"]),
            (work["tris"] - work["finish_tris"], work["triangle_wave_steps"])):
        assert 0 < wave_steps <= steps <= 64 * wave_steps
    # instance entries of the finishing kernel are not told apart: only the lower bound holds
    assert 0 < work["instance_wave_steps"] <= work["instance_records"]
    assert work["node_bytes"] > 0 and work["triangle_bytes"] > 0 and work["instance_bytes"] > 0


def test_forced_trace_policies_are_not_measured(cornell):
    w, h, md, tile, spp = 640, 800, 3, (64, 80), 16
    flags = hip.FLAG_NO_BIN_RAYS | hip.FLAG_TRACE_CHUNKS
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT, flags=flags)
    try:
        r.upload_scene(cornell.desc)
        r.upload_seeds(host.seed_schedule(0, 2 * spp, r.tile_count))
        r.clear()
        for call in range(2):
            r.render(cornell.camera(w, h), call * spp, spp)
            r.read_accum()
        assert r.ray_binning() == (hip.BINNING_OFF, 0.0)
        assert r.trace_refill() == (hip.REFILL_OFF, -1.0, -1.0, 0)
    finally:
        r.close()
    for both in (hip.FLAG_BIN_RAYS | hip.FLAG_NO_BIN_RAYS,
                 hip.FLAG_TRACE_REFILL | hip.FLAG_TRACE_CHUNKS):
        with pytest.raises(RuntimeError, match="on and off together"):
            hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT, flags=both)


def test_trip_counts_and_kernel_times(hyperion):
    w, h, md, tile, spp = 640, 360, 7, (64, 36), 4
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT, flags=hip.FLAG_TIME_KERNELS)
    try:
        r.upload_scene(hyperion.desc)
        r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
        r.clear()
        r.render(hyperion.camera(w, h), 0, spp)
        st = r.stats()
        counts = [r.trip_counts(t) for t in range(md + 1)]
        assert counts[0] == (w * h * spp, w * h * spp, 0)
        live = [c[0] for c in counts]
        # paths only die; bounces handed to the finishing kernel report nothing
        handed_over = next((t for t in range(1, md + 1) if live[t] == 0), md + 1)
        assert all(live[t] >= live[t + 1] for t in range(handed_over - 1))
        assert all(c == (0, 0, 0) for c in counts[handed_over + 1:])
        assert sum(c[1] for c in counts) <= st["closest_rays"]
        assert sum(c[2] for c in counts) <= st["shadow_rays"]
        # every path of bounce t has a closest-hit ray, an occlusion ray or both
        assert all(max(c[1], c[2]) <= c[0] <= c[1] + c[2] for c in counts)
        whole_ms, calls = r.kernel_time(hip.KERNEL_RENDER)
        extend_ms, extends = r.kernel_time(hip.KERNEL_EXTEND)
        shade_ms, shades = r.kernel_time(hip.KERNEL_SHADE)
        assert calls == 1 and extends == shades and 1 <= extends <= md + 1
        assert 0.0 < extend_ms * extends + shade_ms * shades <= whole_ms
        with pytest.raises(RuntimeError):
            r.trip_counts(md + 5)
    finally:
        r.close()


def test_one_pass_larger_than_a_batch_is_refused(cornell, monkeypatch):
    """A batch holds at most 2^28 paths, or what RT_HIP_MAX_BATCH_PATHS allows: here 2^22, and
    2048 x 2064 pixels = 4.2 M paths per pass.  The pipeline must refuse the call (it used to
    truncate its lists silently); the fused kernel, which keeps no lists, still renders it, and
    so does the pipeline in two halves - or in one piece once the bound is lifted."""
    w, h, tile = 2048, 2064, (256, 258)
    monkeypatch.setenv("RT_HIP_MAX_BATCH_PATHS", str(1 << 22))
    r = hip.HipRenderer(w, h, 1, tile, accel=hip.ACCEL_REBUILT)
    try:
        r.upload_scene(cornell.desc)
        r.upload_seeds(host.seed_schedule(0, 1, r.tile_count))
        cam = cornell.camera(w, h)
        r.clear()
        with pytest.raises(RuntimeError, match="more than the 4194304"):
            r.render(cam, 0, 1)
        with pytest.raises(RuntimeError, match="exceeds one batch"):
            r.reserve(w, h, 1)
        r.render(cam, 0, 1, rect=(0, 0, w, h // 2))
        r.render(cam, 0, 1, rect=(0, h // 2, w, h))
        halves, _ = r.read_accum()
        monkeypatch.delenv("RT_HIP_MAX_BATCH_PATHS")
        r.clear()
        r.render(cam, 0, 1)
        whole, _ = r.read_accum()
    finally:
        r.close()
    assert np.array_equal(np.nan_to_num(halves), np.nan_to_num(whole))
    # (an edge-on light sample makes the shader itself produce NaN: a few pixels in 4 M)
    assert np.isnan(halves).any(axis=2).mean() < 1e-5 and np.nanmax(halves[..., :3]) > 0
    assert (np.nan_to_num(halves[::517, ::511, :3]).sum(axis=2) > 0).mean() > 0.5


def test_batches_larger_than_one_scan_pass(cornell):
    """160 passes of 1280 x 720 in one call are two batches of 73.7 M paths: more than 2^26, so
    the one-workgroup scan over the compaction's tile sums takes several tiles per thread.  The
    image must be the one that calls of 40 passes (batches of 18 M paths) give, bit for bit."""
    w, h, md, tile, spp = 1280, 720, 2, (160, 90), 160
    images = []
    for per_call in (spp, 40):
        r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
        try:
            r.upload_scene(cornell.desc)
            r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
            cam = cornell.camera(w, h)
            r.clear()
            for first in range(0, spp, per_call):
                r.render(cam, first, per_call)
            images.append(r.read_accum()[0])
        finally:
            r.close()
    assert np.array_equal(np.nan_to_num(images[0]), np.nan_to_num(images[1]))
    assert np.nanmax(images[0][..., :3]) > 0


def test_reserve_then_render(cornell):
    w, h, md, tile, spp = 320, 184, 5, (64, 46), 4
    plain = run(cornell, w, h, md, tile, spp, hip.ACCEL_REBUILT, 0)
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
    try:
        r.reserve(w, h, spp)
        r.upload_scene(cornell.desc)
        r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
        r.clear()
        r.render(cornell.camera(w, h), 0, spp)
        img, _ = r.read_accum()
        with pytest.raises(RuntimeError, match="bad size"):
            r.reserve(w + 1, h, spp)
    finally:
        r.close()
    assert same(img, plain[0])


def test_shading_kernels_are_chosen_by_what_the_scene_can_do(request, monkeypatch):
    """The pipeline's shading kernels exist once per feature set; a scene gets the smallest set
    that covers its materials and lights.  Code compiled out is code no path of the scene can
    reach, so the image equals the one of the kernels with everything in (forced through the
    environment) bit for bit - and the fused kernel's, which always carries everything
    (test_pipeline_equals_fused_kernel)."""
    expect = {"cornell": (0, 0), "hyperion": (7, 7), "staircase": (13, 13), "mustang": (31, 31),
              "teapot": (24, 24), "instanced": (23, 23), "textured": (28, 31)}
    w, h, md, tile, spp = 320, 184, 6, (64, 46), 3
    for name, want in expect.items():
        scene = request.getfixturevalue(name)
        images = []
        for forced in (None, "31"):
            if forced:
                monkeypatch.setenv("RT_HIP_SHADE_FEATURES", forced)
            else:
                monkeypatch.delenv("RT_HIP_SHADE_FEATURES", raising=False)
            r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
            try:
                r.upload_scene(scene.desc)
                got = r.shade_features()
                if name != "instanced" and name != "textured":
                    assert got == (want if not forced else (want[0], 31)), (name, got)
                else:
                    assert got[1] == (31 if forced else got[1]) and (got[0] & ~got[1]) == 0
                r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
                r.clear()
                r.render(scene.camera(w, h), 0, spp)
                images.append(r.read_accum()[0])
            finally:
                r.close()
        assert same(images[0], images[1]), name


def test_tune_settles_the_policies_before_the_first_render(hyperion):
    """rt_hip_tune makes the two measured calls itself, into a scratch image: afterwards both
    policies are settled, accumulator and pass counter are untouched, and the first render is
    the settled one - with the pixels of a render that was never tuned."""
    w, h, md, tile, spp = 640, 800, 7, (64, 80), 16
    plain = run(hyperion, w, h, md, tile, spp, hip.ACCEL_REBUILT, hip.FLAG_FUSED)
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
    try:
        r.upload_scene(hyperion.desc)
        cam = hyperion.camera(w, h)
        r.clear()
        assert r.trace_refill()[0] == hip.REFILL_UNDECIDED
        r.tune(cam, spp)                     # needs no seeds: it draws from a constant
        state, off, on, mask = r.trace_refill()
        assert state in (hip.REFILL_OFF, hip.REFILL_ON) and off > 0.0 and on > 0.0
        assert r.ray_binning()[0] != hip.BINNING_UNDECIDED
        img, passes = r.read_accum()
        assert passes == 0 and not img[..., :3].any()
        r.upload_seeds(host.seed_schedule(0, spp, r.tile_count))
        r.render(cam, 0, spp)
        assert r.trace_refill() == (state, off, on, mask)   # nothing left to measure
        img, passes = r.read_accum()
    finally:
        r.close()
    assert passes == spp and same(img, plain[0])


def test_refill_measurement_compares_calls_of_different_size_per_ray(hyperion):
    """The two measured calls may differ in size: the verdict is taken per ray, and a call that
    is not within a factor of two of the base line replaces it instead of being compared."""
    w, h, md, tile = 640, 800, 7, (64, 80)
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
    try:
        r.upload_scene(hyperion.desc)
        r.upload_seeds(host.seed_schedule(0, 128, r.tile_count))
        cam = hyperion.camera(w, h)
        r.clear()
        r.render(cam, 0, 48)           # base line: 24.6 M paths
        r.read_accum()
        assert r.trace_refill()[0] == hip.REFILL_UNDECIDED and r.trace_refill()[1] > 0.0
        r.render(cam, 48, 9)           # 4.6 M paths: too small to compare, the new base line
        r.read_accum()
        state, off_small, on, _ = r.trace_refill()
        assert state == hip.REFILL_UNDECIDED and on < 0.0
        r.render(cam, 57, 12)          # 6.1 M paths: compared with it, per ray
        r.read_accum()
        state, off, on, mask = r.trace_refill()
        assert state in (hip.REFILL_OFF, hip.REFILL_ON) and off == off_small and on > 0.0
    finally:
        r.close()


def test_two_batches_in_flight_side_by_side(hyperion):
    """A call of at least 2^24 paths runs as two (or more) batches, two at a time on two
    streams with a set of path state each; a pixel's passes are still added in pass order, so
    image and work counters equal the fused kernel's, which runs them in one launch - also
    when a second call follows at once on the same stream and reuses both sets, and with three
    batches (the third waits for the first one's set)."""
    w, h, md, tile, spp = 640, 800, 7, (64, 80), 36      # 18.4 M paths: two batches of 18 passes
    fused = run(hyperion, w, h, md, tile, 2 * spp, hip.ACCEL_REBUILT,
                hip.FLAG_FUSED | hip.FLAG_COUNT_NODES)
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT,
                        flags=hip.FLAG_COUNT_NODES | hip.FLAG_TRACE_REFILL | hip.FLAG_NO_BIN_RAYS)
    try:
        r.upload_scene(hyperion.desc)
        r.upload_seeds(host.seed_schedule(0, 2 * spp, r.tile_count))
        cam = hyperion.camera(w, h)
        r.clear()
        r.render(cam, 0, spp)          # two batches side by side
        r.render(cam, spp, spp)        # and again, without waiting
        img, passes = r.read_accum()
        st = r.stats()
        live = [r.trip_counts(t)[0] for t in range(md + 1)]
    finally:
        r.close()
    assert passes == 2 * spp and same(img, fused[0])
    for k in COUNTERS:
        assert st[k] == fused[2][k], k
    # the last call's two batches together: every path of its 36 passes was alive at bounce 0
    assert live[0] == w * h * spp and all(a >= b for a, b in zip(live, live[1:]))


def test_calls_on_different_streams_of_one_handle_are_ordered(hyperion):
    """A handle's calls share its path state and counters.  Issued on two different streams
    back to back (no host synchronisation in between), the second must wait for the first on
    the device: the image equals the one-stream render bit for bit."""
    import torch
    w, h, md, tile = 640, 360, 5, (64, 36)
    cam = hyperion.camera(w, h)
    seeds = host.seed_schedule(0, 8, 100)
    r = hip.HipRenderer(w, h, md, tile, accel=hip.ACCEL_REBUILT)
    try:
        r.upload_scene(hyperion.desc)
        r.upload_seeds(seeds)
        r.clear()
        r.render(cam, 0, 8)
        want, _ = r.read_accum()
        a, b = torch.cuda.Stream(), torch.cuda.Stream()
        r.clear(stream=a.cuda_stream)
        for k in range(8):
            r.render(cam, k, 1, stream=(a if k % 2 == 0 else b).cuda_stream)
        torch.cuda.synchronize()
        got, passes = r.read_accum()
    finally:
        r.close()
    assert passes == 8 and same(got, want)
    assert torch.cuda.current_device() == 0
