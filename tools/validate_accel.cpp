// Host-side check of the rebuilt acceleration structure (no GPU needed): builds it for scene
// snapshots and verifies that
//   * the nodes form trees (every node reached exactly once, references in range),
//   * every TLAS leaf is reached exactly once, and the BLAS subtrees of the leaves that belong
//     to one instance (an opened instance has several) hold every triangle of its mesh
//     exactly once,
//   * every child box contains everything below it,
//   * the reported stack bound equals what the deepest walk can have pending.
// usage: validate_accel <scene.rtsnap>...      (built by __graft_entry__.build())
#include <algorithm>
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
    void grow(double x, double y, double z) {
        double const p[3] = {x, y, z};
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(dbox const& b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
};

[[noreturn]] void fail(std::string const& what) { throw std::runtime_error(what); }

struct checker {
    built_accel const& acc;
    rt_scene_desc const& sc;
    std::vector<int> node_seen, instance_seen;
    std::vector<dbox> subtree_box;         // per TLAS leaf: exact object-space bounds of its subtree
    std::vector<uint32_t>* collect = nullptr;  // triangles seen by the BLAS walk in progress

    // Returns the exact bounds of everything below `ref`; `need` = pending siblings.
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
                ++instance_seen[first];
                rt_instance const& inst = sc.instances[acc.instances[first].ref_index];
                dbox const& mb = subtree_box[first];
                for (int corner = 0; corner < 8; ++corner) {
                    double const p[3] = {(corner & 1) ? mb.hi[0] : mb.lo[0],
                        (corner & 2) ? mb.hi[1] : mb.lo[1], (corner & 4) ? mb.hi[2] : mb.lo[2]};
                    double w[3] = {p[0], p[1], p[2]};
                    if (inst.transform >= 0) {
                        rt_mat4 const& m = sc.transforms[inst.transform];
                        for (int r = 0; r < 3; ++r) {
                            w[r] = (double) m.m[0][r] * p[0] + (double) m.m[1][r] * p[1] +
                                   (double) m.m[2][r] * p[2] + (double) m.m[3][r];
                        }
                    }
                    all.grow(w[0], w[1], w[2]);
                }
            } else {
                if (first + count > acc.tris.size()) {
                    fail("bad BLAS leaf");
                }
                for (uint32_t t = first; t < first + count; ++t) {
                    tri_pos const& tp = acc.tris[t];
                    collect->push_back(tp.ref_index);
                    all.grow(tp.ax, tp.ay, tp.az);
                    all.grow(tp.bx, tp.by, tp.bz);
                    all.grow(tp.cx, tp.cy, tp.cz);
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
            double const lo[3] = {n.lo_x[c], n.lo_y[c], n.lo_z[c]};
            double const hi[3] = {n.hi_x[c], n.hi_y[c], n.hi_z[c]};
            for (int a = 0; a < 3; ++a) {
                // TLAS leaves were padded by the builder; everything else is exact
                double const slack = tlas ? 1e-4 * (1.0 + std::max(std::abs(lo[a]), std::abs(hi[a]))) : 0.0;
                if (lo[a] > below.lo[a] + slack || hi[a] < below.hi[a] - slack) {
                    char msg[256];
                    snprintf(msg, sizeof msg,
                        "node %d child %d axis %d: box [%.9g, %.9g] does not contain [%.9g, %.9g]",
                        ref, c, a, lo[a], hi[a], below.lo[a], below.hi[a]);
                    fail(msg);
                }
            }
            all.grow(below);
        }
        if (kids < 2) {
            fail("inner node with fewer than two children");
        }
        need = (kids - 1u) + deepest;
        return all;
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
        std::vector<int>(acc.instances.size(), 0), std::vector<dbox>(acc.instances.size())};
    // every distinct BLAS subtree once: its bounds, its triangles, its stack need
    struct subtree_info {
        dbox box;
        std::vector<uint32_t> triangles;
    };
    std::map<int32_t, subtree_info> subtrees;
    uint32_t need_blas = 0;
    for (size_t r = 0; r < acc.instances.size(); ++r) {
        int32_t const root = acc.instances[r].blas_root;
        auto found = subtrees.find(root);
        if (found == subtrees.end()) {
            subtree_info info;
            ck.collect = &info.triangles;
            uint32_t need = 0;
            info.box = ck.walk(root, false, need);
            need_blas = std::max(need_blas, need);
            found = subtrees.emplace(root, std::move(info)).first;
        }
        ck.subtree_box[r] = found->second.box;
    }
    // per scene instance: its leaves together hold each triangle of its mesh exactly once
    std::vector<std::vector<size_t>> leaves_of(sc.instance_count);
    for (size_t r = 0; r < acc.instances.size(); ++r) {
        if (acc.instances[r].ref_index >= sc.instance_count) {
            fail("TLAS leaf refers to a missing instance");
        }
        leaves_of[acc.instances[r].ref_index].push_back(r);
    }
    std::vector<int> seen;
    for (uint32_t i = 0; i < sc.instance_count; ++i) {
        rt_mesh const& mesh = sc.meshes[sc.instances[i].mesh];
        seen.assign(mesh.triangle_count, 0);
        for (size_t r : leaves_of[i]) {
            for (uint32_t t : subtrees[acc.instances[r].blas_root].triangles) {
                if (t < mesh.triangle_offset || t >= mesh.triangle_offset + mesh.triangle_count) {
                    fail("TLAS leaf holds a triangle of another mesh");
                }
                ++seen[t - mesh.triangle_offset];
            }
        }
        for (int n : seen) {
            if (n != 1) {
                fail("triangle not in exactly one leaf of its instance");
            }
        }
    }
    uint32_t need_tlas = 0;
    ck.walk(acc.tlas_root, true, need_tlas);
    size_t reached = 0;
    for (int n : ck.node_seen) {
        reached += n == 1 ? 1 : 0;
    }
    for (int n : ck.instance_seen) {
        if (n != 1) {
            fail("TLAS leaf not reached exactly once");
        }
    }
    uint32_t const need = need_tlas + 1u + need_blas;
    if (need != acc.max_stack_bound) {
        fail("stack bound " + std::to_string(acc.max_stack_bound) + " but walks need " +
             std::to_string(need));
    }
    printf("%s: ok - %zu nodes (%zu reachable), %zu triangles, %u instances in %zu TLAS leaves, "
           "stack bound %u (TLAS %u, BLAS %u)\n",
        path, acc.nodes.size(), reached, acc.tris.size(), sc.instance_count, acc.instances.size(),
        need, need_tlas, need_blas);
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
