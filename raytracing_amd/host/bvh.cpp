// Binary BVH builder with the reference's exact tree shape.
//
// The compute pass indexes instances and triangles by their position in the
// leaf-ordered arrays this builder emits, so hit ids are only comparable when
// the tree is built the same way.  Behaviour kept from
// src/renderer/bvh.cpp:126-254:
//   * a node becomes a leaf when it holds one primitive, when its box has
//     zero surface area, or when all centroids coincide on the widest axis;
//   * two primitives are ordered by centroid along the widest centroid axis;
//   * otherwise centroids are binned into 12 buckets along that axis.  The
//     reference's search for the cheapest split compares `min_cost < cost`
//     with min_cost initialised to FLT_MAX, which never holds, so the chosen
//     split is always "bucket 0 | buckets 1..11" and the cost estimate stays
//     FLT_MAX.  The consequence: 4 or more primitives always split after
//     bucket 0, exactly 3 primitives always form a leaf.  The bucket costs do
//     not influence the result and are not computed here;
//   * std::partition decides the order inside each side.
// Nodes are appended in depth-first pre-order, which is already the linear
// layout (left child = parent + 1).
#include "bvh.h"

#include <algorithm>
#include <limits>
#include <span>
#include <stdexcept>

namespace {

struct box {
    float x_min = std::numeric_limits<float>::max();
    float x_max = std::numeric_limits<float>::lowest();
    float y_min = std::numeric_limits<float>::max();
    float y_max = std::numeric_limits<float>::lowest();
    float z_min = std::numeric_limits<float>::max();
    float z_max = std::numeric_limits<float>::lowest();

    void grow(float x, float y, float z) {
        x_min = std::min(x, x_min);
        x_max = std::max(x, x_max);
        y_min = std::min(y, y_min);
        y_max = std::max(y, y_max);
        z_min = std::min(z, z_min);
        z_max = std::max(z, z_max);
    }
    void grow(box const& o) {
        x_min = std::min(x_min, o.x_min);
        x_max = std::max(x_max, o.x_max);
        y_min = std::min(y_min, o.y_min);
        y_max = std::max(y_max, o.y_max);
        z_min = std::min(z_min, o.z_min);
        z_max = std::max(z_max, o.z_max);
    }
    rt::vec3 centroid() const {
        return {0.5f * (x_min + x_max), 0.5f * (y_min + y_max), 0.5f * (z_min + z_max)};
    }
    float surface_area() const {
        float const dx = x_max - x_min;
        float const dy = y_max - y_min;
        float const dz = z_max - z_min;
        return 2 * (dx * dy + dy * dz + dz * dx);
    }
    // x only when strictly widest; otherwise y when strictly wider than z; else z.
    int widest_axis() const {
        float const dx = x_max - x_min;
        float const dy = y_max - y_min;
        float const dz = z_max - z_min;
        if (dx > dy && dx > dz) {
            return 0;
        }
        return dy > dz ? 1 : 2;
    }
    float lo(int axis) const { return axis == 0 ? x_min : (axis == 1 ? y_min : z_min); }
    float hi(int axis) const { return axis == 0 ? x_max : (axis == 1 ? y_max : z_max); }
};

struct build_item {
    box bounds;
    uint32_t object = 0;  // index into the unsorted object array
};

box bounds_of(std::span<vertex const> verts) {
    box b;
    for (vertex const& v : verts) {
        b.grow(v.position_texu[0], v.position_texu[1], v.position_texu[2]);
    }
    return b;
}

box bounds_of(std::span<vertex const> verts, rt::mat4 const& to_world) {
    box b;
    for (vertex const& v : verts) {
        rt::vec4 const p = rt::mul(
            to_world, rt::vec4{v.position_texu[0], v.position_texu[1], v.position_texu[2], 1.0f});
        b.grow(p.x, p.y, p.z);
    }
    return b;
}

constexpr uint32_t bucket_count = 12;
constexpr uint32_t max_leaf_size = 3;

template <typename OBJ>
class tree_builder {
public:
    tree_builder(std::vector<rt_bvh_node>& nodes, std::vector<OBJ> const& objects,
        std::vector<OBJ>& sorted, uint32_t sorted_cursor)
        : nodes_{nodes}, objects_{objects}, sorted_{sorted}, cursor_{sorted_cursor} {}

    // Returns the index of the subtree root inside nodes_.
    uint32_t build(std::span<build_item> items) {
        if (items.empty()) {
            throw std::runtime_error("bvh: empty primitive range");
        }
        uint32_t const self = (uint32_t) nodes_.size();
        nodes_.push_back(rt_bvh_node{});
        box bounds;
        for (build_item const& it : items) {
            bounds.grow(it.bounds);
        }
        size_t mid = 0;
        int axis = -1;
        if (choose_split(items, bounds, mid, axis)) {
            uint32_t const left = build(items.subspan(0, mid));
            uint32_t const right = build(items.subspan(mid));
            box merged;
            merged.grow(to_box(nodes_[left]));
            merged.grow(to_box(nodes_[right]));
            rt_bvh_node& n = nodes_[self];
            store_box(n, merged);
            n.right = right;
            n.first_obj = 0;
            n.obj_count = 0;
            n.split_axis = axis;
        } else {
            for (size_t i = 0; i < items.size(); ++i) {
                sorted_[cursor_ + i] = objects_[items[i].object];
            }
            rt_bvh_node& n = nodes_[self];
            store_box(n, bounds);
            n.right = 0;
            n.first_obj = cursor_;
            n.obj_count = (uint32_t) items.size();
            n.split_axis = -1;
            cursor_ += (uint32_t) items.size();
        }
        return self;
    }

private:
    static box to_box(rt_bvh_node const& n) {
        box b;
        b.x_min = n.x_min;
        b.x_max = n.x_max;
        b.y_min = n.y_min;
        b.y_max = n.y_max;
        b.z_min = n.z_min;
        b.z_max = n.z_max;
        return b;
    }
    static void store_box(rt_bvh_node& n, box const& b) {
        n.x_min = b.x_min;
        n.x_max = b.x_max;
        n.y_min = b.y_min;
        n.y_max = b.y_max;
        n.z_min = b.z_min;
        n.z_max = b.z_max;
    }

    // True when the range is to be split at `mid` (items reordered in place).
    static bool choose_split(
        std::span<build_item> items, box const& bounds, size_t& mid, int& axis) {
        if (items.size() == 1 || bounds.surface_area() == 0.0f) {
            return false;
        }
        box centroids;
        for (build_item const& it : items) {
            rt::vec3 const c = it.bounds.centroid();
            centroids.grow(c.x, c.y, c.z);
        }
        axis = centroids.widest_axis();
        float const lo = centroids.lo(axis);
        float const hi = centroids.hi(axis);
        if (lo == hi) {
            return false;
        }
        if (items.size() == 2) {
            if (items[0].bounds.centroid()[axis] > items[1].bounds.centroid()[axis]) {
                std::swap(items[0], items[1]);
            }
            mid = 1;
            return true;
        }
        if (items.size() <= max_leaf_size) {
            return false;
        }
        auto in_first_bucket = [=](build_item const& it) -> bool {
            float const offset = (it.bounds.centroid()[axis] - lo) / (hi - lo);
            uint32_t b = (uint32_t) ((float) bucket_count * offset);
            if (b == bucket_count) {
                b = bucket_count - 1;
            }
            return b <= 0;
        };
        auto const split = std::partition(items.begin(), items.end(), in_first_bucket);
        mid = (size_t) (split - items.begin());
        return true;
    }

    std::vector<rt_bvh_node>& nodes_;
    std::vector<OBJ> const& objects_;
    std::vector<OBJ>& sorted_;
    uint32_t cursor_;
};

}  // namespace

bvh create_bvh(scene const& scene) {
    if (scene.vertices.size() % 3 != 0) {
        throw std::runtime_error("bvh: vertex count is not a multiple of 3");
    }
    if (scene.mesh_vertex_start.empty()) {
        throw std::runtime_error("bvh: scene has no mesh");
    }
    size_t const mesh_total = scene.mesh_vertex_start.size();
    auto mesh_vertices = [&](uint32_t mesh) -> std::span<vertex const> {
        uint32_t const first = scene.mesh_vertex_start[mesh];
        uint32_t const end = mesh + 1 < mesh_total ? scene.mesh_vertex_start[mesh + 1] :
                                                     (uint32_t) scene.vertices.size();
        return std::span<vertex const>{scene.vertices}.subspan(first, end - first);
    };

    // Instances: every primitive in order, then every area light in order.
    std::vector<rt_instance> instances;
    std::vector<build_item> instance_items;
    for (uint32_t p = 0; p < scene.primitives.size(); ++p) {
        primitive const& prim = scene.primitives[p];
        instances.push_back(rt_instance{prim.mesh, prim.transform, prim.material, prim.medium, -1});
        instance_items.push_back(build_item{
            bounds_of(mesh_vertices(prim.mesh), scene.transformation[(uint32_t) prim.transform]),
            p});
    }
    for (uint32_t l = 0; l < scene.lights.size(); ++l) {
        light const& lt = scene.lights[l];
        if (lt.type == RT_LIGHT_DISTANT || lt.type == RT_LIGHT_SKY) {
            continue;
        }
        instances.push_back(rt_instance{lt.mesh, lt.transform, -1, -1, (int32_t) l});
        instance_items.push_back(build_item{
            bounds_of(mesh_vertices(lt.mesh), scene.transformation[(uint32_t) lt.transform]),
            (uint32_t) instances.size() - 1});
    }

    std::vector<rt_triangle> triangles;
    std::vector<build_item> triangle_items;
    uint32_t const triangle_total = (uint32_t) (scene.vertices.size() / 3);
    triangles.reserve(triangle_total);
    triangle_items.reserve(triangle_total);
    for (uint32_t t = 0; t < triangle_total; ++t) {
        triangles.push_back(rt_triangle{
            scene.vertices[3 * t + 0], scene.vertices[3 * t + 1], scene.vertices[3 * t + 2]});
        triangle_items.push_back(
            build_item{bounds_of(std::span<vertex const>{scene.vertices}.subspan(3 * t, 3)), t});
    }

    bvh out;
    out.instances.resize(instances.size());
    tree_builder<rt_instance>{out.tlas, instances, out.instances, 0}.build(instance_items);

    out.triangles.resize(triangles.size());
    for (uint32_t m = 0; m < mesh_total; ++m) {
        uint32_t const first_triangle = scene.mesh_vertex_start[m] / 3;
        uint32_t const count = (uint32_t) (mesh_vertices(m).size() / 3);
        uint32_t const root = (uint32_t) out.blas.size();
        tree_builder<rt_triangle>{out.blas, triangles, out.triangles, first_triangle}.build(
            std::span<build_item>{triangle_items}.subspan(first_triangle, count));
        out.meshes.push_back(rt_mesh{first_triangle, count, root});
    }
    return out;
}

uint32_t bvh_depth(std::vector<rt_bvh_node> const& nodes, uint32_t root) {
    // Iterative walk; the trees this builder emits can be hundreds of levels deep.
    uint32_t deepest = 0;
    std::vector<std::pair<uint32_t, uint32_t>> todo{{root, 1}};
    while (!todo.empty()) {
        auto const [idx, depth] = todo.back();
        todo.pop_back();
        deepest = std::max(deepest, depth);
        rt_bvh_node const& n = nodes[idx];
        if (n.obj_count == 0) {
            todo.push_back({idx + 1, depth + 1});
            todo.push_back({n.right, depth + 1});
        }
    }
    return deepest;
}
