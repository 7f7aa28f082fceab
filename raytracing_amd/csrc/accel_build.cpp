// Binned-SAH builder for the rebuilt acceleration structure.
//
// The host BVH that arrives through the C ABI keeps the reference builder's
// shape, whose split search never moves off "bucket 0 | rest"
// (src/renderer/bvh.cpp:198-205): its trees are tens of levels deeper than
// needed (61 levels for a 15 872-triangle sphere).  Traversal cost on the GPU
// is dominated by dependent node fetches, so the library builds its own trees
// from the uploaded triangles and instances.  Ids reported by the kernels
// still refer to the uploaded arrays (tri_pos::ref_index,
// instance_rec::ref_index).
#include "accel_build.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <stdexcept>

namespace rtd {
namespace {

struct box3 {
    float lo[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(),
        std::numeric_limits<float>::max()};
    float hi[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(),
        std::numeric_limits<float>::lowest()};
    void grow(float const p[3]) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(box3 const& b) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], b.lo[a]);
            hi[a] = std::max(hi[a], b.hi[a]);
        }
    }
    float half_area() const {
        float const dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (dx < 0.0f) {
            return 0.0f;  // empty
        }
        return dx * dy + dy * dz + dz * dx;
    }
};

struct build_prim {
    box3 bounds;
    float centroid[3];
    uint32_t id;
};

// Node of the binary tree the SAH builder produces; collapsed into wide_node afterwards.
struct bin_node {
    box3 b0, b1;
    int32_t child0, child1;
};

constexpr int kBins = 16;
constexpr uint32_t kStackEntries = 40;  // = kFastLdsEntries of the traversal kernels
constexpr float kTraversalCost = 1.2f;
constexpr float kIntersectCost = 1.0f;
// An instance may be opened when the boxes of the other instances cover its own box at least
// this many times over, by surface area.
constexpr float kOpenOverlap = 3.0f;
constexpr uint32_t kMinOpenedEntries = 2048;
constexpr uint32_t kTrianglesPerOpenedEntry = 10;

int32_t encode_leaf(uint32_t first, uint32_t count) {
    return (int32_t) ~((first << 3) | (count - 1u));
}

class tree_builder {
public:
    // emit_leaf(prims, count) appends the leaf's payload and returns its first index.
    // depth_limit: no inner node is created at this depth or beyond, i.e. a walk never has
    // more than depth_limit pending siblings.  A subtree whose remaining levels would not
    // suffice for SAH's uneven splits is split by object median from there on.
    template <typename EMIT>
    int32_t build(std::vector<bin_node>& nodes, build_prim* prims, uint32_t count,
        uint32_t max_leaf, uint32_t depth, uint32_t depth_limit, EMIT& emit_leaf,
        uint32_t& deepest) {
        if (count <= max_leaf &&
            (count == 1 || depth >= depth_limit || !worth_splitting(prims, count))) {
            return encode_leaf(emit_leaf(prims, count), count);
        }
        if (depth >= depth_limit) {
            throw std::runtime_error(
                "accel: a mesh or the instance set is too large for the traversal stack");
        }
        deepest = std::max(deepest, depth + 1u);
        uint32_t const levels_left = depth_limit - depth;
        uint32_t const leaves = (count + max_leaf - 1u) / max_leaf;
        uint32_t needed = 0;  // levels a balanced split needs: ceil(log2(leaves))
        while ((1u << needed) < leaves) {
            ++needed;
        }
        uint32_t const mid = partition(prims, count, needed + 1u >= levels_left);
        uint32_t const self = (uint32_t) nodes.size();
        nodes.push_back(bin_node{});
        box3 b0, b1;
        for (uint32_t i = 0; i < mid; ++i) {
            b0.grow(prims[i].bounds);
        }
        for (uint32_t i = mid; i < count; ++i) {
            b1.grow(prims[i].bounds);
        }
        int32_t const c0 =
            build(nodes, prims, mid, max_leaf, depth + 1, depth_limit, emit_leaf, deepest);
        int32_t const c1 = build(
            nodes, prims + mid, count - mid, max_leaf, depth + 1, depth_limit, emit_leaf, deepest);
        bin_node& n = nodes[self];
        n.b0 = b0;
        n.b1 = b1;
        n.child0 = c0;
        n.child1 = c1;
        return (int32_t) self;
    }

private:
    struct split_choice {
        int axis = -1;
        int bin = 0;
        float cost = std::numeric_limits<float>::max();
        float lo = 0.0f, scale = 0.0f;
    };

    static split_choice best_split(build_prim const* prims, uint32_t count) {
        box3 cb;
        for (uint32_t i = 0; i < count; ++i) {
            cb.grow(prims[i].centroid);
        }
        split_choice best;
        for (int axis = 0; axis < 3; ++axis) {
            float const extent = cb.hi[axis] - cb.lo[axis];
            if (!(extent > 0.0f)) {
                continue;
            }
            float const scale = (float) kBins / extent;
            box3 bin_box[kBins];
            uint32_t bin_count[kBins] = {};
            for (uint32_t i = 0; i < count; ++i) {
                int b = (int) ((prims[i].centroid[axis] - cb.lo[axis]) * scale);
                b = std::min(std::max(b, 0), kBins - 1);
                bin_box[b].grow(prims[i].bounds);
                ++bin_count[b];
            }
            float right_area[kBins];
            uint32_t right_count[kBins];
            box3 acc;
            uint32_t n = 0;
            for (int b = kBins - 1; b > 0; --b) {
                acc.grow(bin_box[b]);
                n += bin_count[b];
                right_area[b] = acc.half_area();
                right_count[b] = n;
            }
            box3 left;
            uint32_t left_n = 0;
            for (int b = 0; b < kBins - 1; ++b) {
                left.grow(bin_box[b]);
                left_n += bin_count[b];
                if (left_n == 0 || right_count[b + 1] == 0) {
                    continue;
                }
                float const cost = left.half_area() * (float) left_n +
                                   right_area[b + 1] * (float) right_count[b + 1];
                if (cost < best.cost) {
                    best.cost = cost;
                    best.axis = axis;
                    best.bin = b;
                    best.lo = cb.lo[axis];
                    best.scale = scale;
                }
            }
        }
        return best;
    }

    static bool worth_splitting(build_prim const* prims, uint32_t count) {
        box3 all;
        for (uint32_t i = 0; i < count; ++i) {
            all.grow(prims[i].bounds);
        }
        float const area = all.half_area();
        if (!(area > 0.0f)) {
            return false;
        }
        split_choice const s = best_split(prims, count);
        if (s.axis < 0) {
            return false;
        }
        float const split_cost = kTraversalCost + kIntersectCost * s.cost / area;
        return split_cost < kIntersectCost * (float) count;
    }

    static uint32_t partition(build_prim* prims, uint32_t count, bool force_median) {
        if (!force_median) {
            split_choice const s = best_split(prims, count);
            if (s.axis >= 0) {
                auto const it = std::partition(prims, prims + count, [&](build_prim const& p) {
                    int b = (int) ((p.centroid[s.axis] - s.lo) * s.scale);
                    b = std::min(std::max(b, 0), kBins - 1);
                    return b <= s.bin;
                });
                uint32_t const mid = (uint32_t) (it - prims);
                if (mid > 0 && mid < count) {
                    return mid;
                }
            }
        }
        // Object median along the widest centroid axis: bounds the depth.
        box3 cb;
        for (uint32_t i = 0; i < count; ++i) {
            cb.grow(prims[i].centroid);
        }
        int axis = 0;
        float widest = cb.hi[0] - cb.lo[0];
        for (int a = 1; a < 3; ++a) {
            if (cb.hi[a] - cb.lo[a] > widest) {
                widest = cb.hi[a] - cb.lo[a];
                axis = a;
            }
        }
        uint32_t const mid = count / 2;
        std::nth_element(prims, prims + mid, prims + count,
            [axis](build_prim const& a, build_prim const& b) {
                return a.centroid[axis] < b.centroid[axis];
            });
        return mid;
    }
};

box3 child_box(wide_node const& n, int c) {
    box3 b;
    b.lo[0] = n.lo_x[c], b.lo[1] = n.lo_y[c], b.lo[2] = n.lo_z[c];
    b.hi[0] = n.hi_x[c], b.hi[1] = n.hi_y[c], b.hi[2] = n.hi_z[c];
    return b;
}

// Height of every binary node: inner nodes on the longest way down to a leaf.
void fill_heights(std::vector<bin_node> const& bin, std::vector<uint32_t>& height) {
    height.assign(bin.size(), 0u);
    // children are built after their parent, so they have larger indices
    for (size_t i = bin.size(); i-- > 0;) {
        uint32_t const h0 = bin[i].child0 >= 0 ? height[(size_t) bin[i].child0] : 0u;
        uint32_t const h1 = bin[i].child1 >= 0 ? height[(size_t) bin[i].child1] : 0u;
        height[i] = 1u + std::max(h0, h1);
    }
}

// Collapses the binary tree under `root` into wide nodes: the children of a wide node are
// found by repeatedly opening the inner child with the largest surface area, up to four.
// A walk can have (children - 1) siblings pending per wide node on its way down; `budget`
// caps the sum along every path (it must be at least the binary height of `root`, which is
// what a walk of the uncollapsed tree needs): a child is opened only if everything below
// still fits.  Deep, narrow parts of a tree therefore stay binary, the rest becomes 4-wide.
// Returns the reference of the collapsed root; `need` = the sum actually reached.
int32_t collapse(std::vector<bin_node> const& bin, std::vector<uint32_t> const& height,
    int32_t root, uint32_t budget, std::vector<wide_node>& out, uint32_t& need) {
    if (root < 0) {
        need = 0;
        return root;  // a tree that is a single leaf
    }
    struct pending {
        int32_t ref;
        box3 box;
    };
    auto height_of = [&](int32_t ref) { return ref >= 0 ? height[(size_t) ref] : 0u; };
    pending kids[4];
    uint32_t count = 2;
    kids[0] = {bin[(size_t) root].child0, bin[(size_t) root].b0};
    kids[1] = {bin[(size_t) root].child1, bin[(size_t) root].b1};
    while (count < 4) {
        // with one more child, everything below must fit into budget - count
        if (budget < count) {
            break;
        }
        uint32_t const below = budget - count;
        int best = -1;
        float best_area = -1.0f;
        for (uint32_t i = 0; i < count; ++i) {
            if (kids[i].ref < 0 || kids[i].box.half_area() <= best_area) {
                continue;
            }
            bool fits = true;
            for (uint32_t j = 0; j < count && fits; ++j) {
                if (j == i) {
                    bin_node const& open = bin[(size_t) kids[i].ref];
                    fits = height_of(open.child0) <= below && height_of(open.child1) <= below;
                } else {
                    fits = height_of(kids[j].ref) <= below;
                }
            }
            if (fits) {
                best_area = kids[i].box.half_area();
                best = (int) i;
            }
        }
        if (best < 0) {
            break;
        }
        bin_node const& open = bin[(size_t) kids[best].ref];
        kids[best] = {open.child0, open.b0};
        kids[count++] = {open.child1, open.b1};
    }
    uint32_t const self = (uint32_t) out.size();
    out.push_back(wide_node{});
    wide_node n{};
    uint32_t deepest = 0;
    for (uint32_t i = 0; i < 4; ++i) {
        if (i < count) {
            n.lo_x[i] = kids[i].box.lo[0], n.lo_y[i] = kids[i].box.lo[1], n.lo_z[i] = kids[i].box.lo[2];
            n.hi_x[i] = kids[i].box.hi[0], n.hi_y[i] = kids[i].box.hi[1], n.hi_z[i] = kids[i].box.hi[2];
            uint32_t child_need = 0;
            n.child[i] =
                collapse(bin, height, kids[i].ref, budget - (count - 1u), out, child_need);
            deepest = std::max(deepest, child_need);
        } else {
            // inverted box: no ray interval can be non-empty
            n.lo_x[i] = n.lo_y[i] = n.lo_z[i] = std::numeric_limits<float>::max();
            n.hi_x[i] = n.hi_y[i] = n.hi_z[i] = std::numeric_limits<float>::lowest();
            n.child[i] = kNoChild;
        }
    }
    out[self] = n;
    need = (count - 1u) + deepest;
    return (int32_t) self;
}

// Renumbers nodes and instance records by how often a ray is expected to fetch them
// (surface-area ratios down from the TLAS root, a BLAS shared by several instances adding
// up), hottest first: the traversal kernel keeps a prefix of both arrays in LDS.
void order_by_heat(built_accel& out) {
    size_t const node_count = out.nodes.size();
    std::vector<double> heat(node_count, 0.0);
    std::vector<double> instance_heat(out.instances.size(), 0.0);
    std::vector<char> is_tlas(node_count, 0);
    struct visit {
        int32_t node;
        double heat;
    };
    std::vector<visit> todo;
    // Spreads `h` from `root` down one tree; `on_leaf(code, h)` sees the leaves.
    auto spread = [&](int32_t root, double h, bool tlas, auto&& on_leaf) {
        if (root < 0) {
            on_leaf(root, h);
            return;
        }
        todo.push_back({root, h});
        while (!todo.empty()) {
            visit const v = todo.back();
            todo.pop_back();
            heat[(size_t) v.node] += v.heat;
            is_tlas[(size_t) v.node] = tlas ? 1 : 0;
            wide_node const& n = out.nodes[(size_t) v.node];
            box3 all;
            for (int c = 0; c < 4; ++c) {
                if (n.child[c] != kNoChild) {
                    all.grow(child_box(n, c));
                }
            }
            float const whole = all.half_area();
            for (int c = 0; c < 4; ++c) {
                int32_t const ref = n.child[c];
                if (ref == kNoChild) {
                    continue;
                }
                double share = v.heat;
                if (whole > 0.0f) {
                    share = v.heat *
                            std::min(1.0, (double) child_box(n, c).half_area() / (double) whole);
                }
                if (ref >= 0) {
                    todo.push_back({ref, share});
                } else {
                    on_leaf(ref, share);
                }
            }
        }
    };
    spread(out.tlas_root, 1.0, true, [&](int32_t code, double h) {
        instance_heat[((uint32_t) ~code) >> 3] += h;
    });
    std::vector<double> root_heat(node_count, 0.0);
    for (size_t i = 0; i < out.instances.size(); ++i) {
        if (out.instances[i].blas_root >= 0) {
            root_heat[(size_t) out.instances[i].blas_root] += instance_heat[i];
        }
    }
    for (size_t r = 0; r < node_count; ++r) {
        if (root_heat[r] > 0.0) {
            spread((int32_t) r, root_heat[r], false, [](int32_t, double) {});
        }
    }

    auto ranking = [](std::vector<double> const& h) {
        std::vector<uint32_t> order(h.size());
        for (size_t i = 0; i < h.size(); ++i) {
            order[i] = (uint32_t) i;
        }
        std::stable_sort(order.begin(), order.end(),
            [&](uint32_t a, uint32_t b) { return h[a] > h[b]; });
        std::vector<uint32_t> rank(h.size());
        for (size_t i = 0; i < order.size(); ++i) {
            rank[order[i]] = (uint32_t) i;
        }
        return rank;
    };
    std::vector<uint32_t> const node_rank = ranking(heat);
    std::vector<uint32_t> const instance_rank = ranking(instance_heat);

    auto remap = [&](int32_t ref, bool tlas) -> int32_t {
        if (ref == kNoChild) {
            return ref;
        }
        if (ref >= 0) {
            return (int32_t) node_rank[(size_t) ref];
        }
        if (tlas) {
            return encode_leaf(instance_rank[((uint32_t) ~ref) >> 3], 1u);
        }
        return ref;  // triangle ranges stay where they are
    };
    std::vector<wide_node> nodes(node_count);
    for (size_t i = 0; i < node_count; ++i) {
        wide_node n = out.nodes[i];
        for (int c = 0; c < 4; ++c) {
            n.child[c] = remap(n.child[c], is_tlas[i] != 0);
        }
        nodes[node_rank[i]] = n;
    }
    out.nodes.swap(nodes);
    out.tlas_root = remap(out.tlas_root, true);
    std::vector<instance_rec> instances(out.instances.size());
    for (size_t i = 0; i < out.instances.size(); ++i) {
        instance_rec rec = out.instances[i];
        if (rec.blas_root >= 0) {
            rec.blas_root = (int32_t) node_rank[(size_t) rec.blas_root];
        }
        instances[instance_rank[i]] = rec;
    }
    out.instances.swap(instances);
}

void set_centroid(build_prim& p) {
    for (int a = 0; a < 3; ++a) {
        p.centroid[a] = 0.5f * (p.bounds.lo[a] + p.bounds.hi[a]);
    }
}

// Pre-splitting of large triangles ("early split clipping", Ernst and Greiner 2007).  A few
// triangles far larger than the rest (the walls and floor of a room among its furniture) end
// up in leaves whose boxes span the scene, and every ray has to test them.  Before the build,
// the references with the largest boxes are cut in two at the middle of their box's longest
// axis - the triangle clipped to either half - until a budget of extra references is used up
// or the largest box is no longer far above the median.  Each part is an ordinary build
// primitive with the part's (slightly padded) box; leaves hold whole triangles, so a triangle
// may be listed in several leaves and a ray may test it twice, with identical results.
struct tri_part {
    float poly[10][3];  // the triangle clipped to the part's box: at most 9 corners
    int corners;
    box3 box;
    uint32_t id;
};

// Clips a convex polygon to coordinate[axis] <= pos (keep_low) or >= pos.
int clip_polygon(float const (*in)[3], int n, int axis, float pos, bool keep_low,
    float (*out)[3]) {
    int m = 0;
    for (int i = 0; i < n && m < 10; ++i) {
        float const* a = in[i];
        float const* b = in[(i + 1) % n];
        bool const a_in = keep_low ? a[axis] <= pos : a[axis] >= pos;
        bool const b_in = keep_low ? b[axis] <= pos : b[axis] >= pos;
        if (a_in) {
            out[m][0] = a[0], out[m][1] = a[1], out[m][2] = a[2];
            ++m;
        }
        if (a_in != b_in && m < 10) {
            float const t = (pos - a[axis]) / (b[axis] - a[axis]);
            for (int k = 0; k < 3; ++k) {
                out[m][k] = a[k] + t * (b[k] - a[k]);
            }
            out[m][axis] = pos;
            ++m;
        }
    }
    return m;
}

void set_part_box(tri_part& part, box3 const& within, bool padded) {
    box3 b;
    for (int i = 0; i < part.corners; ++i) {
        b.grow(part.poly[i]);
    }
    for (int a = 0; a < 3; ++a) {
        if (padded) {
            // the cut points are rounded: never let rounding shave the box
            float const pad = 1e-6f * (b.hi[a] - b.lo[a]) +
                              4e-7f * std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])) + 1e-30f;
            b.lo[a] -= pad;
            b.hi[a] += pad;
        }
        b.lo[a] = std::max(b.lo[a], within.lo[a]);
        b.hi[a] = std::min(b.hi[a], within.hi[a]);
    }
    part.box = b;
}

constexpr float kPresplitAboveMedian = 4.0f;  // only boxes this many times the median area
constexpr uint32_t kPresplitBudgetShift = 3;   // at most triangles / 8 extra references

bool mesh_is_all_degenerate(rt_scene_desc const& sc, rt_mesh const& mesh) {
    for (uint32_t t = 0; t < mesh.triangle_count; ++t) {
        if (!triangle_never_hit(sc.triangles[mesh.triangle_offset + t])) {
            return false;
        }
    }
    return true;
}

// The build primitives of one mesh: one per triangle that a ray can hit (all of them when
// none can: keep_degenerate), the largest ones cut into parts.
void mesh_primitives(rt_scene_desc const& sc, rt_mesh const& mesh, bool keep_degenerate,
    std::vector<build_prim>& prims) {
    std::vector<tri_part> parts;
    std::vector<float> areas;
    parts.reserve(mesh.triangle_count);
    areas.reserve(mesh.triangle_count);
    box3 const everything = [] {
        box3 b;
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = std::numeric_limits<float>::lowest();
            b.hi[a] = std::numeric_limits<float>::max();
        }
        return b;
    }();
    for (uint32_t t = 0; t < mesh.triangle_count; ++t) {
        rt_triangle const& tri = sc.triangles[mesh.triangle_offset + t];
        if (!keep_degenerate && triangle_never_hit(tri)) {
            continue;
        }
        tri_part part{};
        for (int k = 0; k < 3; ++k) {
            part.poly[0][k] = tri.a.position_texu[k];
            part.poly[1][k] = tri.b.position_texu[k];
            part.poly[2][k] = tri.c.position_texu[k];
        }
        part.corners = 3;
        part.id = mesh.triangle_offset + t;
        set_part_box(part, everything, false);
        areas.push_back(part.box.half_area());
        parts.push_back(part);
    }
    uint32_t const listed = (uint32_t) parts.size();
    // RT_HIP_PRESPLIT: tuning knob - 0 disables, another number replaces kPresplitAboveMedian
    char const* const knob = getenv("RT_HIP_PRESPLIT");
    float const above_median = knob ? (float) atof(knob) : kPresplitAboveMedian;
    // RT_HIP_PRESPLIT_BUDGET: tuning knob - extra references allowed, as a fraction of the triangles
    char const* const budget_knob = getenv("RT_HIP_PRESPLIT_BUDGET");
    uint32_t budget = above_median > 0.0f
                          ? (budget_knob ? (uint32_t) ((double) listed * atof(budget_knob))
                                         : listed >> kPresplitBudgetShift)
                          : 0u;
    if (budget > 0u && listed >= 8u) {
        std::nth_element(areas.begin(), areas.begin() + areas.size() / 2, areas.end());
        float const large = above_median * areas[areas.size() / 2];
        auto smaller = [&](uint32_t a, uint32_t b) {
            return parts[a].box.half_area() < parts[b].box.half_area();
        };
        std::vector<uint32_t> heap;
        for (uint32_t t = 0; t < listed; ++t) {
            if (parts[t].box.half_area() > large) {
                heap.push_back(t);
            }
        }
        std::make_heap(heap.begin(), heap.end(), smaller);
        while (!heap.empty() && budget > 0u) {
            std::pop_heap(heap.begin(), heap.end(), smaller);
            uint32_t const at = heap.back();
            heap.pop_back();
            tri_part const whole = parts[at];
            int axis = 0;
            for (int a = 1; a < 3; ++a) {
                if (whole.box.hi[a] - whole.box.lo[a] > whole.box.hi[axis] - whole.box.lo[axis]) {
                    axis = a;
                }
            }
            float const pos = 0.5f * (whole.box.lo[axis] + whole.box.hi[axis]);
            if (!(pos > whole.box.lo[axis] && pos < whole.box.hi[axis])) {
                continue;  // too thin to cut
            }
            tri_part low = whole, high = whole;
            low.corners = clip_polygon(whole.poly, whole.corners, axis, pos, true, low.poly);
            high.corners = clip_polygon(whole.poly, whole.corners, axis, pos, false, high.poly);
            if (low.corners < 3 || high.corners < 3 || low.corners > 9 || high.corners > 9) {
                continue;  // degenerate: leave this one whole
            }
            box3 low_half = whole.box, high_half = whole.box;
            low_half.hi[axis] = pos;
            high_half.lo[axis] = pos;
            set_part_box(low, low_half, true);
            set_part_box(high, high_half, true);
            parts[at] = low;
            parts.push_back(high);
            --budget;
            for (uint32_t piece : {at, (uint32_t) parts.size() - 1u}) {
                if (parts[piece].box.half_area() > large) {
                    heap.push_back(piece);
                    std::push_heap(heap.begin(), heap.end(), smaller);
                }
            }
        }
    }
    prims.resize(parts.size());
    for (size_t i = 0; i < parts.size(); ++i) {
        prims[i].bounds = parts[i].box;
        set_centroid(prims[i]);
        prims[i].id = parts[i].id;
    }
}

// World box of an object-space box under an instance's transform: the 8 transformed corners,
// padded so that rounding cannot cut a hit off.
box3 world_box(rt_scene_desc const& sc, rt_instance const& inst, box3 const& ob) {
    box3 wb;
    for (int corner = 0; corner < 8; ++corner) {
        float const p[3] = {(corner & 1) ? ob.hi[0] : ob.lo[0], (corner & 2) ? ob.hi[1] : ob.lo[1],
            (corner & 4) ? ob.hi[2] : ob.lo[2]};
        float w[3] = {p[0], p[1], p[2]};
        if (inst.transform >= 0) {
            rt_mat4 const& m = sc.transforms[inst.transform];
            for (int r = 0; r < 3; ++r) {
                w[r] = m.m[0][r] * p[0] + m.m[1][r] * p[1] + m.m[2][r] * p[2] + m.m[3][r];
            }
        }
        wb.grow(w);
    }
    for (int a = 0; a < 3; ++a) {
        float const pad = 1e-5f * (wb.hi[a] - wb.lo[a]) +
                          4e-7f * std::max(std::fabs(wb.lo[a]), std::fabs(wb.hi[a])) + 1e-30f;
        wb.lo[a] -= pad;
        wb.hi[a] += pad;
    }
    return wb;
}

// A second bound on an instance's world box: the box of its mesh's bounding sphere (centre c,
// radius r in object space) under the transform, c_world +- r |row_i of the 3x3|.  The box of
// the eight transformed corners of a rotated object-space box is up to sqrt(3) times too wide
// per axis (a rotated sphere's most of all); both bounds hold, so their intersection does.
struct ball {
    double c[3] = {0.0, 0.0, 0.0};
    double r = 0.0;
};

box3 world_ball_box(rt_scene_desc const& sc, rt_instance const& inst, ball const& b) {
    double centre[3] = {b.c[0], b.c[1], b.c[2]};
    double reach[3] = {b.r, b.r, b.r};
    if (inst.transform >= 0) {
        rt_mat4 const& m = sc.transforms[inst.transform];
        for (int r = 0; r < 3; ++r) {
            centre[r] = (double) m.m[0][r] * b.c[0] + (double) m.m[1][r] * b.c[1] +
                        (double) m.m[2][r] * b.c[2] + (double) m.m[3][r];
            reach[r] = b.r * std::sqrt((double) m.m[0][r] * m.m[0][r] + (double) m.m[1][r] * m.m[1][r] +
                                       (double) m.m[2][r] * m.m[2][r]);
        }
    }
    box3 wb;
    for (int a = 0; a < 3; ++a) {
        double const lo = centre[a] - reach[a], hi = centre[a] + reach[a];
        double const pad = 1e-5 * (hi - lo) + 4e-7 * std::max(std::fabs(lo), std::fabs(hi)) + 1e-30;
        wb.lo[a] = std::nextafter((float) (lo - pad), -std::numeric_limits<float>::infinity());
        wb.hi[a] = std::nextafter((float) (hi + pad), std::numeric_limits<float>::infinity());
    }
    return wb;
}

box3 intersection(box3 const& a, box3 const& b) {
    box3 both;
    for (int k = 0; k < 3; ++k) {
        both.lo[k] = std::max(a.lo[k], b.lo[k]);
        both.hi[k] = std::min(a.hi[k], b.hi[k]);
    }
    return both;
}

uint32_t ceil_log2(uint32_t v) {
    uint32_t bits = 0;
    while ((1u << bits) < v) {
        ++bits;
    }
    return bits;
}

// One TLAS leaf: an instance, or - when the instance has been opened - one subtree of its
// mesh's BLAS.
struct tlas_entry {
    uint32_t instance;
    int32_t subtree;  // binary node, or an encoded leaf
    box3 object_box;
    box3 world;
};

}  // namespace

bool triangle_never_hit(rt_triangle const& tri) {
    // hit_triangle (rt_traverse_ref.h): edge1 = b - a, edge2 = c - a,
    // |cross(edge1, edge2)| < 1e-8 with cross = (a.y b.z - b.y a.z, a.z b.x - b.z a.x,
    // a.x b.y - b.x a.y) and length = sqrt((x x + y y) + z z), every step rounded to float
    float const* const a = tri.a.position_texu;
    float const* const b = tri.b.position_texu;
    float const* const c = tri.c.position_texu;
    float const e1x = b[0] - a[0], e1y = b[1] - a[1], e1z = b[2] - a[2];
    float const e2x = c[0] - a[0], e2y = c[1] - a[1], e2z = c[2] - a[2];
    float const p0 = e1y * e2z, p1 = e2y * e1z;
    float const p2 = e1z * e2x, p3 = e2z * e1x;
    float const p4 = e1x * e2y, p5 = e2x * e1y;
    float const cx = p0 - p1, cy = p2 - p3, cz = p4 - p5;
    float const xx = cx * cx, yy = cy * cy, zz = cz * cz;
    float const sum = (xx + yy) + zz;
    return std::fabs(std::sqrt(sum)) < 1e-8f;
}

built_accel build_wide_accel(rt_scene_desc const& sc, uint32_t max_opened_entries) {
    if (sc.instance_count == 0 || sc.mesh_count == 0 || sc.triangle_count == 0) {
        throw std::runtime_error("accel: empty scene");
    }
    if (sc.triangle_count >= (1u << 27)) {  // leaves may list a cut triangle more than once
        throw std::runtime_error("accel: too many triangles for the leaf encoding");
    }
    built_accel out;
    out.tris.reserve(sc.triangle_count);
    tree_builder builder;
    std::vector<bin_node> bin;  // binary trees of every BLAS and of the TLAS

    // Object-space bounds of every mesh.
    std::vector<box3> mesh_bounds(sc.mesh_count);
    std::vector<ball> mesh_balls(sc.mesh_count);
    for (uint32_t m = 0; m < sc.mesh_count; ++m) {
        rt_mesh const& mesh = sc.meshes[m];
        if (mesh.triangle_count == 0 ||
            (uint64_t) mesh.triangle_offset + mesh.triangle_count > sc.triangle_count) {
            throw std::runtime_error("accel: mesh triangle range out of bounds");
        }
        for (uint32_t t = 0; t < mesh.triangle_count; ++t) {
            rt_triangle const& tri = sc.triangles[mesh.triangle_offset + t];
            mesh_bounds[m].grow(tri.a.position_texu);
            mesh_bounds[m].grow(tri.b.position_texu);
            mesh_bounds[m].grow(tri.c.position_texu);
        }
        // bounding sphere around the middle of the box
        ball& b = mesh_balls[m];
        for (int k = 0; k < 3; ++k) {
            b.c[k] = 0.5 * ((double) mesh_bounds[m].lo[k] + (double) mesh_bounds[m].hi[k]);
        }
        double far2 = 0.0;
        for (uint32_t t = 0; t < mesh.triangle_count; ++t) {
            rt_triangle const& tri = sc.triangles[mesh.triangle_offset + t];
            for (const float* v : {tri.a.position_texu, tri.b.position_texu, tri.c.position_texu}) {
                double d2 = 0.0;
                for (int k = 0; k < 3; ++k) {
                    d2 += ((double) v[k] - b.c[k]) * ((double) v[k] - b.c[k]);
                }
                far2 = std::max(far2, d2);
            }
        }
        b.r = std::sqrt(far2) * (1.0 + 1e-12);
    }
    for (uint32_t i = 0; i < sc.instance_count; ++i) {
        rt_instance const& inst = sc.instances[i];
        if (inst.mesh >= sc.mesh_count ||
            (inst.transform >= 0 && (uint32_t) inst.transform >= sc.transform_count)) {
            throw std::runtime_error("accel: instance refers to a missing mesh or transform");
        }
    }

    // Which instances stand among others: those whose box is covered kOpenOverlap times over
    // (by surface area) by the boxes of the other instances.  A floor under a car or spheres
    // on a plane are not; opening them only multiplies the instance records a ray has to enter.
    char const* const gate = getenv("RT_HIP_OPEN_OVERLAP");  // tuning knob for experiments
    float const overlap_needed = gate ? (float) atof(gate) : kOpenOverlap;
    std::vector<box3> instance_world(sc.instance_count);
    for (uint32_t i = 0; i < sc.instance_count; ++i) {
        instance_world[i] =
            intersection(world_box(sc, sc.instances[i], mesh_bounds[sc.instances[i].mesh]),
                world_ball_box(sc, sc.instances[i], mesh_balls[sc.instances[i].mesh]));
    }
    std::vector<char> crowded(sc.instance_count, 0);
    uint64_t crowded_triangles = 0;
    if (max_opened_entries > 0u) {
        // by x-sorted sweep: instances whose x ranges do not meet cannot overlap
        std::vector<uint32_t> by_x(sc.instance_count);
        for (uint32_t i = 0; i < sc.instance_count; ++i) {
            by_x[i] = i;
        }
        std::sort(by_x.begin(), by_x.end(), [&](uint32_t a, uint32_t b) {
            return instance_world[a].lo[0] < instance_world[b].lo[0];
        });
        std::vector<double> covered(sc.instance_count, 0.0);
        for (uint32_t a = 0; a < sc.instance_count; ++a) {
            box3 const& ba = instance_world[by_x[a]];
            for (uint32_t b = a + 1; b < sc.instance_count; ++b) {
                box3 const& bb = instance_world[by_x[b]];
                if (bb.lo[0] > ba.hi[0]) {
                    break;
                }
                box3 both;
                bool empty = false;
                for (int k = 0; k < 3; ++k) {
                    both.lo[k] = std::max(ba.lo[k], bb.lo[k]);
                    both.hi[k] = std::min(ba.hi[k], bb.hi[k]);
                    empty = empty || both.lo[k] > both.hi[k];
                }
                if (!empty) {
                    covered[by_x[a]] += both.half_area();
                    covered[by_x[b]] += both.half_area();
                }
            }
        }
        for (uint32_t i = 0; i < sc.instance_count; ++i) {
            crowded[i] = covered[i] >= (double) overlap_needed * instance_world[i].half_area() ? 1 : 0;
            if (crowded[i]) {
                crowded_triangles += sc.meshes[sc.instances[i].mesh].triangle_count;
            }
        }
    }
    // TLAS leaves to add by opening: one per kTrianglesPerOpenedEntry triangles of the crowded
    // instances (staircase, 88 k triangles: 2 048 more leaves 214 Msamples/s, 8 192 more 229,
    // 16 384 more 225, 65 536 more 208), within [kMinOpenedEntries, max_opened_entries].
    uint32_t opened_entries = 0;
    if (crowded_triangles > 0u) {
        opened_entries = (uint32_t) std::min<uint64_t>(max_opened_entries,
            std::max<uint64_t>(kMinOpenedEntries, crowded_triangles / kTrianglesPerOpenedEntry));
    }

    // The stack holds TLAS siblings + one marker + BLAS siblings.  The TLAS gets the levels a
    // balanced tree over its entries needs plus six for SAH's uneven splits (the builder
    // falls back to median splits where those run out), the BLASes the rest - at least 12
    // levels (4 * 2^12 triangles balanced).
    uint32_t const entry_budget = sc.instance_count + opened_entries;
    uint32_t const tlas_limit =
        std::min(kStackEntries - 1u - 12u, std::max(2u, ceil_log2(entry_budget) + 6u));
    uint32_t const blas_limit = kStackEntries - 1u - tlas_limit;

    // One BLAS per mesh, shared by its instances.
    std::vector<build_prim> prims;
    std::vector<int32_t> mesh_root(sc.mesh_count);
    for (uint32_t m = 0; m < sc.mesh_count; ++m) {
        rt_mesh const& mesh = sc.meshes[m];
        bool const keep_degenerate = mesh_is_all_degenerate(sc, mesh);
        out.lists_degenerate_triangles = out.lists_degenerate_triangles || keep_degenerate;
        mesh_primitives(sc, mesh, keep_degenerate, prims);
        auto emit = [&](build_prim const* leaf, uint32_t count) -> uint32_t {
            uint32_t const first = (uint32_t) out.tris.size();
            for (uint32_t i = 0; i < count; ++i) {
                rt_triangle const& tri = sc.triangles[leaf[i].id];
                tri_pos tp{};
                tp.ax = tri.a.position_texu[0];
                tp.ay = tri.a.position_texu[1];
                tp.az = tri.a.position_texu[2];
                tp.ref_index = leaf[i].id;
                tp.bx = tri.b.position_texu[0];
                tp.by = tri.b.position_texu[1];
                tp.bz = tri.b.position_texu[2];
                tp.cx = tri.c.position_texu[0];
                tp.cy = tri.c.position_texu[1];
                tp.cz = tri.c.position_texu[2];
                out.tris.push_back(tp);
            }
            return first;
        };
        // Leaves of 4; meshes too large for the remaining levels get leaves of 8.
        uint32_t leaf_size = 4u;
        if (((uint64_t) leaf_size << (blas_limit > 1u ? blas_limit - 1u : 0u)) < prims.size()) {
            leaf_size = 8u;
        }
        uint32_t deepest = 0;
        mesh_root[m] = builder.build(bin, prims.data(), (uint32_t) prims.size(), leaf_size, 0u,
            blas_limit, emit, deepest);
    }

    // Opening.  Instances whose world boxes overlap (the walls, floor and furniture of a room
    // are separate meshes that all span the room) make every ray walk several BLASes from the
    // root.  Replacing the entry with the largest world box by the two halves of its BLAS,
    // again and again, yields TLAS leaves that are small where the scene is crowded; each
    // points at a subtree of the shared BLAS through its own instance record, so the
    // traversal is unchanged.  (Benthin, Woop, Wald, Afra: "Improved two-level BVHs using
    // partial re-braiding", HPG 2017.)
    std::vector<tlas_entry> entries;
    entries.reserve(entry_budget);
    for (uint32_t i = 0; i < sc.instance_count; ++i) {
        tlas_entry e;
        e.instance = i;
        e.subtree = mesh_root[sc.instances[i].mesh];
        e.object_box = mesh_bounds[sc.instances[i].mesh];
        e.world = instance_world[i];
        entries.push_back(e);
    }
    auto area_less = [&](uint32_t a, uint32_t b) {
        return entries[a].world.half_area() < entries[b].world.half_area();
    };
    std::vector<uint32_t> heap;  // entries that can still be opened, largest world box on top
    auto offer = [&](uint32_t at) {
        if (entries[at].subtree >= 0 && crowded[entries[at].instance]) {
            heap.push_back(at);
            std::push_heap(heap.begin(), heap.end(), area_less);
        }
    };
    for (uint32_t i = 0; i < sc.instance_count; ++i) {
        offer(i);
    }
    while (!heap.empty() && entries.size() < entry_budget) {
        std::pop_heap(heap.begin(), heap.end(), area_less);
        uint32_t const at = heap.back();
        heap.pop_back();
        tlas_entry const opened = entries[at];
        bin_node const node = bin[(size_t) opened.subtree];
        // the first half takes the opened entry's place, the second is appended
        entries[at].subtree = node.child0;
        entries[at].object_box = node.b0;
        // (a part of the instance: inside the instance's own world box, too)
        entries[at].world = intersection(
            world_box(sc, sc.instances[opened.instance], node.b0), instance_world[opened.instance]);
        offer(at);
        tlas_entry second = opened;
        second.subtree = node.child1;
        second.object_box = node.b1;
        second.world = intersection(
            world_box(sc, sc.instances[opened.instance], node.b1), instance_world[opened.instance]);
        entries.push_back(second);
        offer((uint32_t) entries.size() - 1u);
    }

    // TLAS over the entries; its leaves become the instance records.
    prims.resize(entries.size());
    for (uint32_t e = 0; e < entries.size(); ++e) {
        prims[e].bounds = entries[e].world;
        set_centroid(prims[e]);
        prims[e].id = e;
    }
    std::vector<int32_t> record_subtree;  // per instance record, until the collapse below
    auto emit_instance = [&](build_prim const* leaf, uint32_t count) -> uint32_t {
        (void) count;  // TLAS leaves hold exactly one entry
        uint32_t const first = (uint32_t) out.instances.size();
        tlas_entry const& entry = entries[leaf[0].id];
        rt_instance const& inst = sc.instances[entry.instance];
        instance_rec rec{};
        if (inst.transform >= 0) {
            rt_mat4 const& inv = sc.inverse_transforms[inst.transform];
            for (int c = 0; c < 4; ++c) {
                for (int r = 0; r < 3; ++r) {
                    rec.m[c * 3 + r] = inv.m[c][r];
                }
            }
            rec.has_transform = 1;
        } else {
            rec.m[0] = rec.m[4] = rec.m[8] = 1.0f;
            rec.has_transform = 0;
        }
        rec.blas_root = 0;  // set after the collapse
        rec.ref_index = entry.instance;
        out.instances.push_back(rec);
        record_subtree.push_back(entry.subtree);
        return first;
    };
    uint32_t deepest_tlas = 0;
    int32_t const tlas_bin_root = builder.build(bin, prims.data(), (uint32_t) entries.size(), 1u,
        0u, tlas_limit, emit_instance, deepest_tlas);

    // Collapse the binary trees into (up to) 4-wide nodes.  The traversal stack is shared by
    // the TLAS walk, the instance marker and the BLAS walk: split it between the two levels in
    // proportion to their binary heights (which is what each needs at least).
    std::vector<uint32_t> height;
    fill_heights(bin, height);
    auto height_of = [&](int32_t ref) { return ref >= 0 ? height[(size_t) ref] : 0u; };
    uint32_t const tlas_height = height_of(tlas_bin_root);
    uint32_t blas_height = 0;
    for (int32_t subtree : record_subtree) {
        blas_height = std::max(blas_height, height_of(subtree));
    }
    uint32_t const room = kStackEntries - 1u;
    if (tlas_height + blas_height > room) {
        throw std::runtime_error("accel: tree does not fit the traversal stack");
    }
    uint32_t blas_budget = blas_height;
    if (tlas_height + blas_height > 0u) {
        blas_budget = std::max(blas_height,
            std::min(room - tlas_height, room * blas_height / (tlas_height + blas_height)));
    }
    uint32_t need_blas = 0;
    // instances of one mesh opened the same way share the collapsed subtrees
    std::vector<int32_t> wide_of(bin.size(), kNoChild);
    for (size_t r = 0; r < out.instances.size(); ++r) {
        int32_t const subtree = record_subtree[r];
        if (subtree < 0) {
            out.instances[r].blas_root = subtree;  // a single leaf
            continue;
        }
        if (wide_of[(size_t) subtree] == kNoChild) {
            uint32_t need = 0;
            wide_of[(size_t) subtree] = collapse(bin, height, subtree, blas_budget, out.nodes, need);
            need_blas = std::max(need_blas, need);
        }
        out.instances[r].blas_root = wide_of[(size_t) subtree];
    }
    uint32_t need_tlas = 0;
    out.tlas_root = collapse(bin, height, tlas_bin_root, room - need_blas, out.nodes, need_tlas);
    out.max_stack_bound = need_tlas + 1u + need_blas;
    if (out.max_stack_bound > kStackEntries) {
        throw std::runtime_error("accel: tree does not fit the traversal stack");
    }
    order_by_heat(out);
    return out;
}

}  // namespace rtd
