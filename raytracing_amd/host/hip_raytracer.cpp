// HIP back end of the renderer table: thin wrappers from the six callbacks to
// the C ABI of librt_hip.so.  State is file-static and initialize()/destroy()
// are idempotent, like the Vulkan back end they replace
// (src/renderer/megakernel_raytracer.cpp:46, 525-529, 793-797).  Violations
// terminate, as the reference's CHECK does (src/check.h:10-17).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <exception>
#include <random>

#include "bvh.h"
#include "renderer.h"
#include "rt_hip.h"

namespace {

bool initialized = false;
hip_raytracer_settings settings{};
render_options options{};
// One renderer per GPU.  handles[0] (`handle`) owns the image: the tile-by-tile cadence and
// the preview run on it alone; batched render() calls split their passes over all of them and
// the accumulators are summed onto it when the image or the pass count is read.
std::vector<rt_hip_renderer*> handles;
rt_hip_renderer* handle = nullptr;
bool reduce_pending = false;
std::mt19937 seed_source{};
uint32_t tiles_x = 0, tiles_y = 0;
uint32_t tile_x = 0, tile_y = 0;  // the reference's tiles.current
uint32_t passes_requested = 0;    // passes traced by batched render() calls

[[noreturn]] void fail(char const* what) {
    std::fprintf(stderr, "hip raytracer: %s: %s\n", what, rt_hip_last_error());
    std::terminate();
}

void check(int code, char const* what) {
    if (code != 0) {
        fail(what);
    }
}

// src/renderer/megakernel_raytracer.cpp:80-86
uint32_t next_random_seed() {
    return (uint32_t) seed_source();
}

// :58-72
bool next_tile() {
    if (tile_x != tiles_x - 1) {
        ++tile_x;
        return false;
    }
    tile_x = 0;
    if (tile_y != tiles_y - 1) {
        ++tile_y;
        return false;
    }
    tile_y = 0;
    return true;
}

void hip_initialize(render_options const& opts) {
    if (initialized) {
        return;
    }
    initialized = true;
    options = opts;
    tiles_x = opts.resolution_x / opts.tile_width;
    tiles_y = opts.resolution_y / opts.tile_height;
    tile_x = tile_y = 0;
    passes_requested = 0;
    if (settings.fixed_seed) {
        seed_source.seed(settings.seed);
    } else {
        seed_source.seed(std::random_device{}());
    }
    rt_hip_config cfg{};
    cfg.width = opts.resolution_x;
    cfg.height = opts.resolution_y;
    cfg.max_depth = opts.max_depth;
    cfg.tile_width = opts.tile_width;
    cfg.tile_height = opts.tile_height;
    cfg.accel = settings.reference_order ? RT_HIP_ACCEL_REFERENCE : RT_HIP_ACCEL_REBUILT;
    cfg.flags = 0;
    std::vector<int> devices = settings.devices;
    if (devices.empty()) {
        devices.push_back(settings.device);
    }
    for (int device : devices) {
        cfg.device = device;
        rt_hip_renderer* h = nullptr;
        check(rt_hip_create(&cfg, &h), "create");
        handles.push_back(h);
    }
    handle = handles.front();
    reduce_pending = false;
}

// Passes [first, first + count) of rank `rank` of `world`: contiguous and balanced.
void pass_share(uint32_t total, uint32_t world, uint32_t rank, uint32_t& first, uint32_t& count) {
    uint32_t const base = total / world, extra = total % world;
    first = rank * base + std::min(rank, extra);
    count = base + (rank < extra ? 1u : 0u);
}

void reduce_if_pending() {
    if (reduce_pending) {
        check(rt_hip_reduce(handles.data(), (uint32_t) handles.size()), "reduce");
        reduce_pending = false;
    }
}

void hip_prepare_data(scene const& sc) {
    if (sc.lights.empty()) {
        std::fprintf(stderr, "hip raytracer: the scene has no light\n");
        std::terminate();
    }
    // :272-282: light count, sky light index, BVH, inverse transforms
    bvh const accel = create_bvh(sc);
    std::vector<rt::mat4> inverse;
    inverse.reserve(sc.transformation.size());
    for (rt::mat4 const& m : sc.transformation) {
        inverse.push_back(rt::inverse(m));
    }
    std::vector<rt_texture> textures;
    for (texture_data const& t : sc.textures) {
        rt_texture e{};
        e.width = t.width;
        e.height = t.height;
        e.format = (uint32_t) t.format;
        e.data = t.data.data();
        textures.push_back(e);
    }
    rt_scene_desc d{};
    d.tlas = accel.tlas.data();
    d.blas = accel.blas.data();
    d.meshes = accel.meshes.data();
    d.instances = accel.instances.data();
    d.triangles = accel.triangles.data();
    d.transforms = reinterpret_cast<rt_mat4 const*>(sc.transformation.data());
    d.inverse_transforms = reinterpret_cast<rt_mat4 const*>(inverse.data());
    d.materials = sc.materials.data();
    d.mediums = sc.mediums.data();
    d.lights = sc.lights.data();
    d.textures = textures.data();
    d.tlas_count = (uint32_t) accel.tlas.size();
    d.blas_count = (uint32_t) accel.blas.size();
    d.mesh_count = (uint32_t) accel.meshes.size();
    d.instance_count = (uint32_t) accel.instances.size();
    d.triangle_count = (uint32_t) accel.triangles.size();
    d.transform_count = (uint32_t) sc.transformation.size();
    d.material_count = (uint32_t) sc.materials.size();
    d.medium_count = (uint32_t) sc.mediums.size();
    d.light_count = (uint32_t) sc.lights.size();
    d.texture_count = (uint32_t) textures.size();
    d.sky_light =
        sc.lights.back().type == RT_LIGHT_SKY ? (int32_t) sc.lights.size() - 1 : -1;
    for (rt_hip_renderer* h : handles) {
        check(rt_hip_upload_scene(h, &d), "upload_scene");
        check(rt_hip_clear(h, nullptr, nullptr), "clear");
    }
}

void hip_update_data(scene const&) {
}

void hip_render(camera const& cam) {
    if (cam.dirty) {
        // :600-619: restart accumulation, show a 1/10-resolution single-sample preview
        tile_x = tile_y = 0;
        passes_requested = 0;
        reduce_pending = false;  // whatever the other GPUs hold is discarded with the image
        for (size_t k = 1; k < handles.size(); ++k) {
            check(rt_hip_clear(handles[k], nullptr, nullptr), "clear");
        }
        rt_camera const pc = get_glsl_raytracer_camera(
            cam, std::max(1u, options.resolution_x / 10), std::max(1u, options.resolution_y / 10));
        check(rt_hip_render_preview(handle, &pc, next_random_seed(), nullptr), "render_preview");
        return;
    }
    rt_camera const rc =
        get_glsl_raytracer_camera(cam, options.resolution_x, options.resolution_y);
    if (settings.passes_per_render > 0 && tile_x == 0 && tile_y == 0) {
        // Batched form: n whole passes in one launch.  Seeds are drawn in the order the
        // tile-by-tile cadence would draw them, so both forms give the same image.
        uint32_t const n = settings.passes_per_render;
        if (passes_requested == 0) {
            for (rt_hip_renderer* h : handles) {
                check(rt_hip_clear(h, nullptr, nullptr), "clear");
            }
        }
        std::vector<uint32_t> seeds((size_t) n * tiles_x * tiles_y);
        for (uint32_t& s : seeds) {
            s = next_random_seed();
        }
        // every GPU gets the whole call's seed words and traces its own range of the passes:
        // the launches are asynchronous, so the GPUs work side by side
        uint32_t const world = (uint32_t) handles.size();
        for (uint32_t k = 0; k < world; ++k) {
            uint32_t first = 0, count = 0;
            pass_share(n, world, k, first, count);
            if (count == 0) {
                continue;
            }
            check(rt_hip_upload_seeds(handles[k], seeds.data(), passes_requested, n),
                "upload_seeds");
            check(rt_hip_render(handles[k], &rc, passes_requested + first, count, nullptr, nullptr,
                      nullptr),
                "render");
        }
        reduce_pending = world > 1;
        passes_requested += n;
        return;
    }
    // :628-669: first tile of the first pass clears, every call traces one tile
    if (hip_raytracer_passes() == 0 && tile_x == 0 && tile_y == 0) {
        check(rt_hip_clear(handle, nullptr, nullptr), "clear");
    }
    check(rt_hip_render_tile(handle, &rc, next_random_seed(), tile_x, tile_y, nullptr),
        "render_tile");
    next_tile();
}

void hip_present() {
    // No swapchain.  The display transform (accum / passes -> Reinhard -> gamma 2.2,
    // shaders/raytracer/rect.frag:18-23) is applied by whoever reads the image.
}

void hip_destroy() {
    if (!initialized) {
        return;
    }
    initialized = false;
    for (rt_hip_renderer* h : handles) {
        check(rt_hip_destroy(h), "destroy");
    }
    handles.clear();
    handle = nullptr;
}

}  // namespace

void load_hip_raytracer(renderer& r) {
    r.initialize = hip_initialize;
    r.prepare_data = hip_prepare_data;
    r.update_data = hip_update_data;
    r.render = hip_render;
    r.present = hip_present;
    r.destroy = hip_destroy;
}

void configure_hip_raytracer(hip_raytracer_settings const& s) {
    settings = s;
}

uint32_t hip_raytracer_passes() {
    reduce_if_pending();
    uint32_t passes = 0;
    check(rt_hip_passes(handle, &passes), "passes");
    return passes;
}

std::vector<float> hip_raytracer_read_accumulation() {
    reduce_if_pending();
    std::vector<float> image((size_t) options.resolution_x * options.resolution_y * 4);
    check(rt_hip_read_accum(handle, image.data(), nullptr, nullptr), "read_accum");
    return image;
}

std::vector<float> hip_raytracer_read_preview() {
    size_t const w = std::max(1u, options.resolution_x / 10);
    size_t const h = std::max(1u, options.resolution_y / 10);
    std::vector<float> image(w * h * 4);
    check(rt_hip_read_preview(handle, image.data(), nullptr), "read_preview");
    return image;
}
