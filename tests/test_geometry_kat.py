"""Hand-built cases for the slab test and the watertight triangle test
(shaders/raytracer/ray.glsl:11-31, 47-105, 119-183), including the tie rule."""
import numpy as np
import pytest

import oracle

INF = np.float32(1.0 / 0.00000000000001)
UNIT_BOX = [0, 1, 0, 1, 0, 1]
TRI = [[0, 0, 5], [1, 0, 5], [0, 1, 5]]


def norm(v):
    v = np.asarray(v, dtype=np.float32)
    return v / np.float32(np.sqrt((v * v).sum()))


@pytest.mark.parametrize("origin,direction,t_max,expect", [
    ([0.5, 0.5, -1], [0, 0, 1], INF, True),        # straight through
    ([0.5, 0.5, -1], [0, 0, -1], INF, False),      # pointing away
    ([0.5, 0.5, -1], [0, 0, 1], 0.5, False),       # box beyond t_max
    ([0.5, 0.5, -1], [0, 0, 1], 1.0, True),        # touching at t_max is a hit
    ([0.5, 0.5, 0.5], [1, 0, 0], INF, True),       # origin inside
    ([2, 2, 2], norm([-1, -1, -1]), INF, True),    # diagonal through the far corner
    ([2, 0.5, 0.5], [0, 1, 0], INF, False),        # parallel, outside the slab
    ([-1, 1.0, 0.5], [1, 0, 0], INF, True),        # grazing along a face: touching counts
])
def test_hit_aabb(origin, direction, t_max, expect):
    assert oracle.hit_aabb(UNIT_BOX, origin, direction, 0.0, float(t_max)) is expect


def test_hit_aabb_negative_zero_direction_quirk():
    # direction.x = -0.0 is "not negative" for the sign test but 1/d is -inf, which swaps the
    # slab's near/far distances: the shader rejects a ray that starts inside the slab.
    inside = oracle.hit_aabb(UNIT_BOX, [0.5, 0.5, -1], [0.0, 0, 1], 0.0, float(INF))
    flipped = oracle.hit_aabb(UNIT_BOX, [0.5, 0.5, -1], [-0.0, 0, 1], 0.0, float(INF))
    assert inside is True and flipped is False


def test_hit_triangle_interior_and_barycentrics():
    hit, out = oracle.hit_triangle(TRI, [0.25, 0.25, 0], [0, 0, 1], 0.0, float(INF))
    assert hit
    b0, b1, b2, t = out
    assert t == np.float32(5.0)
    assert (b0, b1, b2) == (np.float32(0.5), np.float32(0.25), np.float32(0.25))
    # b_i weights vertex i: reconstruct the hit point
    p = b0 * np.float32(TRI[0]) + b1 * np.float32(TRI[1]) + b2 * np.float32(TRI[2])
    assert np.allclose(p, [0.25, 0.25, 5])


def test_hit_triangle_backface_also_hits():
    hit, out = oracle.hit_triangle(TRI, [0.25, 0.25, 10], [0, 0, -1], 0.0, float(INF))
    assert hit and out[3] == np.float32(5.0)


@pytest.mark.parametrize("origin", [[0.5, 0.0, 0], [0.0, 0.5, 0], [0.5, 0.5, 0], [0, 0, 0],
                                    [1, 0, 0], [0, 1, 0]])
def test_hit_triangle_edges_and_vertices_are_inside(origin):
    # a zero edge function is not "mixed sign": edges and vertices count as hits
    hit, _ = oracle.hit_triangle(TRI, origin, [0, 0, 1], 0.0, float(INF))
    assert hit
    assert oracle.hit_triangle_quick(TRI, origin, [0, 0, 1], 0.0, float(INF))


@pytest.mark.parametrize("origin", [[0.75, 0.75, 0], [-0.01, 0.5, 0], [0.5, -0.01, 0]])
def test_hit_triangle_outside(origin):
    hit, _ = oracle.hit_triangle(TRI, origin, [0, 0, 1], 0.0, float(INF))
    assert not hit


def test_hit_triangle_range_rule_and_ties():
    o, d = [0.25, 0.25, 0], [0, 0, 1]
    # accept 0 < t <= t_max: a hit exactly at t_max is accepted, so the later of two equal-t
    # hits replaces the earlier one
    assert oracle.hit_triangle(TRI, o, d, 0.0, 5.0)[0]
    assert not oracle.hit_triangle(TRI, o, d, 0.0, float(np.nextafter(np.float32(5), 0)))[0]
    # t = 0 (origin on the triangle) is rejected
    assert not oracle.hit_triangle(TRI, [0.25, 0.25, 5], d, 0.0, float(INF))[0]
    # behind the origin
    assert not oracle.hit_triangle(TRI, [0.25, 0.25, 6], d, 0.0, float(INF))[0]


def test_hit_triangle_degenerate_and_parallel():
    sliver = [[0, 0, 5], [1, 0, 5], [2, 0, 5]]
    assert not oracle.hit_triangle(sliver, [0.5, 0, 0], [0, 0, 1], 0.0, float(INF))[0]
    tiny = [[0, 0, 5], [1e-5, 0, 5], [0, 1e-5, 5]]  # |cross| = 1e-10 < 1e-8
    assert not oracle.hit_triangle(tiny, [1e-6, 1e-6, 0], [0, 0, 1], 0.0, float(INF))[0]
    # ray in the triangle's plane: all edge functions vanish -> det == 0
    assert not oracle.hit_triangle(TRI, [-1, 0.25, 5], [1, 0, 0], 0.0, float(INF))[0]


def test_quick_variant_agrees_with_full_test():
    rng = np.random.RandomState(11)
    agree = 0
    for _ in range(2000):
        tri = rng.uniform(-1, 1, size=(3, 3)).astype(np.float32)
        o = rng.uniform(-2, 2, size=3).astype(np.float32)
        d = norm(rng.normal(size=3))
        t_max = float(rng.uniform(0.5, 4))
        full = oracle.hit_triangle(tri, o, d, 0.0, t_max)[0]
        quick = oracle.hit_triangle_quick(tri, o, d, 0.0, t_max)
        assert full == quick
        agree += full
    assert agree > 5  # the sweep does exercise hits
