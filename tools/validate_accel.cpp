// Host-side check of the rebuilt acceleration structure (no GPU needed): builds it for scene
// snapshots and verifies that
//   * the nodes form trees (every node reached exactly once, references in range) and every
//     TLAS leaf is reached exactly once;
//   * every child box contains the boxes below it;
//   * the BLAS subtrees of the TLAS leaves that belong to one instance (an opened instance has
//     several) list every triangle of its mesh at least once and no triangle of another mesh;
//   * every triangle can be found: for sample points on it (its centroid, points next to its
//     corners and on its edges), walking down from the TLAS root through boxes that contain the
//     point reaches a leaf that lists the triangle - large triangles are cut into parts with
//     boxes of their own, so this is what guarantees that a ray hitting the triangle anywhere
//     is led to it;
//   * the reported stack bound equals what the deepest walk can have pending.
// usage: validate_accel <scene.rtsnap>...      (built by __graft_entry__.build())
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "accel_build.h"
#include "rt_host.h"

using namespace rtd;

namespace {

struct dbox {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    bool empty() const { return lo[0] > hi[0]; }
    void grow(dbox const& b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
};

[[noreturn]] void fail(std::string const& what) { throw std::runtime_error(what); }

dbox child_box_of(wide_node const& n, int c) {
    dbox b;
    b.lo[0] = n.lo_x[c], b.lo[1] = n.lo_y[c], b.lo[2] = n.lo_z[c];
    b.hi[0] = n.hi_x[c], b.hi[1] = n.hi_y[c], b.hi[2] = n.hi_z[c];
    return b;
}

bool holds(dbox const& b, double const p[3]) {
    for (int a = 0; a < 3; ++a) {
        double const slack = 2e-6 * (1.0 + std::fabs(p[a]));
        if (p[a] < b.lo[a] - slack || p[a] > b.hi[a] + slack) {
            return false;
        }
    }
    return true;
}

struct checker {
    built_accel const& acc;
    rt_scene_desc const& sc;
    std::vector<int> node_seen, leaf_seen;
    std::vector<uint32_t>* collect = nullptr;  // triangles listed below the BLAS walk in progress

    // Structure walk.  Returns the union of the child boxes of `ref` (empty for a leaf);
    // `need` = siblings a walk can have pending below.
    dbox walk(int32_t ref, bool tlas, uint32_t& need) {
        dbox all;
        need = 0;
        if (ref == kNoChild) {
            fail("walked into an unused slot");
        }
        if (ref < 0) {
            uint32_t const code = (uint32_t) ~ref;
            uint32_t const first = code >> 3, count = (code & 7u) + 1u;
            if (tlas) {
                if (count != 1 || first >= acc.instances.size()) {
                    fail("bad TLAS leaf");
                }
                ++leaf_seen[first];
            } else {
                if (first + count > acc.tris.size()) {
                    fail("bad BLAS leaf");
                }
                for (uint32_t t = first; t < first + count; ++t) {
                    if (acc.tris[t].ref_index >= sc.triangle_count) {
                        fail("leaf lists a missing triangle");
                    }
                    collect->push_back(acc.tris[t].ref_index);
                }
            }
            return all;
        }
        if ((size_t) ref >= acc.nodes.size()) {
            fail("node reference out of range");
        }
        if (node_seen[(size_t) ref]++ != 0) {
            fail("node reached twice (not a tree)");
        }
        wide_node const& n = acc.nodes[(size_t) ref];
        uint32_t kids = 0, deepest = 0;
        for (int c = 0; c < 4; ++c) {
            if (n.child[c] == kNoChild) {
                if (!(n.lo_x[c] > n.hi_x[c])) {
                    fail("unused slot without an inverted box");
                }
                continue;
            }
            ++kids;
            uint32_t child_need = 0;
            dbox const below = walk(n.child[c], tlas, child_need);
            deepest = std::max(deepest, child_need);
            dbox const mine = child_box_of(n, c);
            if (!below.empty()) {
                for (int a = 0; a < 3; ++a) {
                    if (mine.lo[a] > below.lo[a] || mine.hi[a] < below.hi[a]) {
                        char msg[256];
                        snprintf(msg, sizeof msg,
                            "node %d child %d axis %d: box [%.9g, %.9g] does not contain [%.9g, %.9g]",
                            ref, c, a, mine.lo[a], mine.hi[a], below.lo[a], below.hi[a]);
                        fail(msg);
                    }
                }
            }
            all.grow(mine);
        }
        if (kids < 2) {
            fail("inner node with fewer than two children");
        }
        need = (kids - 1u) + deepest;
        return all;
    }

    // Is triangle `id` listed in a leaf below `ref` that a walk through boxes holding `p`
    // reaches?
    bool find_in_blas(int32_t ref, double const p[3], uint32_t id) const {
        if (ref < 0) {
            uint32_t const code = (uint32_t) ~ref;
            uint32_t const first = code >> 3, count = (code & 7u) + 1u;
            for (uint32_t t = first; t < first + count; ++t) {
                if (acc.tris[t].ref_index == id) {
                    return true;
                }
            }
            return false;
        }
        wide_node const& n = acc.nodes[(size_t) ref];
        for (int c = 0; c < 4; ++c) {
            if (n.child[c] != kNoChild && holds(child_box_of(n, c), p) &&
                find_in_blas(n.child[c], p, id)) {
                return true;
            }
        }
        return false;
    }

    // The same from the TLAS: `world` must lead to a leaf of `instance` whose subtree leads
    // `object` to the triangle.
    bool find_from_tlas(int32_t ref, double const world[3], double const object[3],
        uint32_t instance, uint32_t id) const {
        if (ref < 0) {
            instance_rec const& rec = acc.instances[((uint32_t) ~ref) >> 3];
            return rec.ref_index == instance && find_in_blas(rec.blas_root, object, id);
        }
        wide_node const& n = acc.nodes[(size_t) ref];
        for (int c = 0; c < 4; ++c) {
            if (n.child[c] != kNoChild && holds(child_box_of(n, c), world) &&
                find_from_tlas(n.child[c], world, object, instance, id)) {
                return true;
            }
        }
        return false;
    }
};

void validate(char const* path) {
    rt_host_scene* scene = nullptr;
    if (rt_host_load_snapshot_file(path, &scene) != 0) {
        fail(std::string{"load: "} + rt_host_last_error());
    }
    rt_scene_desc sc{};
    rt_host_scene_desc(scene, &sc);
    built_accel const acc = build_wide_accel(sc);
    checker ck{acc, sc, std::vector<int>(acc.nodes.size(), 0),
        std::vector<int>(acc.instances.size(), 0)};
    // every distinct BLAS subtree once: the triangles it lists, its stack need
    std::map<int32_t, std::vector<uint32_t>> listed;
    uint32_t need_blas = 0;
    for (instance_rec const& rec : acc.instances) {
        if (listed.find(rec.blas_root) == listed.end()) {
            std::vector<uint32_t> triangles;
            ck.collect = &triangles;
            uint32_t need = 0;
            ck.walk(rec.blas_root, false, need);
            need_blas = std::max(need_blas, need);
            listed.emplace(rec.blas_root, std::move(triangles));
        }
    }
    uint32_t need_tlas = 0;
    ck.walk(acc.tlas_root, true, need_tlas);
    size_t reached = 0;
    for (int n : ck.node_seen) {
        reached += n == 1 ? 1 : 0;
    }
    for (int n : ck.leaf_seen) {
        if (n != 1) {
            fail("TLAS leaf not reached exactly once");
        }
    }
    // per scene instance: its leaves together list each triangle of its mesh, and only those
    std::vector<std::vector<size_t>> leaves_of(sc.instance_count);
    for (size_t r = 0; r < acc.instances.size(); ++r) {
        if (acc.instances[r].ref_index >= sc.instance_count) {
            fail("TLAS leaf refers to a missing instance");
        }
        leaves_of[acc.instances[r].ref_index].push_back(r);
    }
    // (large instance counts: every instance's leaf list is checked, the point walks below
    // are done for a spread of at most 64 instances)
    uint32_t const stride = std::max(1u, sc.instance_count / 64u);
    size_t points = 0;
    std::vector<int> seen;
    for (uint32_t i = 0; i < sc.instance_count; ++i) {
        rt_instance const& inst = sc.instances[i];
        rt_mesh const& mesh = sc.meshes[inst.mesh];
        seen.assign(mesh.triangle_count, 0);
        for (size_t r : leaves_of[i]) {
            for (uint32_t t : listed[acc.instances[r].blas_root]) {
                if (t < mesh.triangle_offset || t >= mesh.triangle_offset + mesh.triangle_count) {
                    fail("TLAS leaf lists a triangle of another mesh");
                }
                ++seen[t - mesh.triangle_offset];
            }
        }
        for (uint32_t t = 0; t < mesh.triangle_count; ++t) {
            // triangles no ray can hit are left out on purpose (accel_build.h)
            bool const left_out = !acc.lists_degenerate_triangles &&
                                  rtd::triangle_never_hit(sc.triangles[mesh.triangle_offset + t]);
            if (seen[t] < 1 && !left_out) {
                fail("triangle in no leaf of its instance");
            }
            if (seen[t] > 0 && left_out) {
                fail("a triangle no ray can hit is listed in a leaf");
            }
        }
        if (i % stride != 0) {
            continue;
        }
        static double const weights[7][3] = {{1 / 3.0, 1 / 3.0, 1 / 3.0}, {0.98, 0.01, 0.01},
            {0.01, 0.98, 0.01}, {0.01, 0.01, 0.98}, {0.5, 0.5, 0.0}, {0.0, 0.5, 0.5},
            {0.5, 0.0, 0.5}};
        for (uint32_t t = 0; t < mesh.triangle_count; ++t) {
            rt_triangle const& tri = sc.triangles[mesh.triangle_offset + t];
            if (!acc.lists_degenerate_triangles && rtd::triangle_never_hit(tri)) {
                continue;
            }
            for (auto const& w : weights) {
                double object[3], world[3];
                for (int k = 0; k < 3; ++k) {
                    object[k] = w[0] * tri.a.position_texu[k] + w[1] * tri.b.position_texu[k] +
                                w[2] * tri.c.position_texu[k];
                    world[k] = object[k];
                }
                if (inst.transform >= 0) {
                    rt_mat4 const& m = sc.transforms[inst.transform];
                    for (int r = 0; r < 3; ++r) {
                        world[r] = (double) m.m[0][r] * object[0] + (double) m.m[1][r] * object[1] +
                                   (double) m.m[2][r] * object[2] + (double) m.m[3][r];
                    }
                }
                ++points;
                if (!ck.find_from_tlas(acc.tlas_root, world, object, i, mesh.triangle_offset + t)) {
                    char msg[200];
                    snprintf(msg, sizeof msg,
                        "instance %u triangle %u: point (%.3f %.3f %.3f) is not led to it", i, t,
                        w[0], w[1], w[2]);
                    fail(msg);
                }
            }
        }
    }
    uint32_t const need = need_tlas + 1u + need_blas;
    if (need != acc.max_stack_bound) {
        fail("stack bound " + std::to_string(acc.max_stack_bound) + " but walks need " +
             std::to_string(need));
    }
    uint32_t never_hit = 0;
    for (uint32_t t = 0; t < sc.triangle_count; ++t) {
        never_hit += rtd::triangle_never_hit(sc.triangles[t]) ? 1u : 0u;
    }
    printf("%s: ok - %zu nodes (%zu reachable), %u triangles listed %zu times, %u instances in "
           "%zu TLAS leaves, %zu points found, stack bound %u (TLAS %u, BLAS %u), %u triangles "
           "no ray can hit %s\n",
        path, acc.nodes.size(), reached, sc.triangle_count, acc.tris.size(), sc.instance_count,
        acc.instances.size(), points, need, need_tlas, need_blas, never_hit,
        acc.lists_degenerate_triangles ? "kept" : "left out");
    rt_host_free_scene(scene);
}

}  // namespace

int main(int argc, char** argv) {
    int bad = 0;
    for (int i = 1; i < argc; ++i) {
        try {
            validate(argv[i]);
        } catch (std::exception const& e) {
            printf("%s: FAILED - %s\n", argv[i], e.what());
            bad = 1;
        }
    }
    return bad;
}
