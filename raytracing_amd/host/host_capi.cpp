// C interface over the host layer; see include/rt_host.h.
#include "rt_host.h"

#include <algorithm>
#include <exception>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>

#include "bvh.h"
#include "scene.h"

struct rt_host_scene {
    render_options options;
    camera cam;
    scene sc;
    bvh accel;
    std::vector<rt::mat4> inverse_transformation;
    std::vector<rt_texture> texture_table;
    uint32_t light_count = 0;
    int32_t sky_light = -1;
};

namespace {

thread_local std::string last_error;

template <typename F>
int guarded(F&& body) {
    try {
        body();
        return 0;
    } catch (std::exception const& e) {
        last_error = e.what();
    } catch (...) {
        last_error = "unknown error";
    }
    return 1;
}

// Derives everything prepare_data() computes on the host before uploading.
void finish(rt_host_scene& s) {
    if (s.sc.lights.empty()) {
        throw std::runtime_error("scene has no light (the reference requires at least one)");
    }
    s.light_count = (uint32_t) s.sc.lights.size();
    s.sky_light = s.sc.lights.back().type == RT_LIGHT_SKY ? (int32_t) s.sc.lights.size() - 1 : -1;
    s.accel = create_bvh(s.sc);
    s.inverse_transformation.clear();
    for (rt::mat4 const& m : s.sc.transformation) {
        s.inverse_transformation.push_back(rt::inverse(m));
    }
    s.texture_table.clear();
    for (texture_data const& t : s.sc.textures) {
        rt_texture e{};
        e.width = t.width;
        e.height = t.height;
        e.format = (uint32_t) t.format;
        e.data = t.data.data();
        s.texture_table.push_back(e);
    }
}

void copy3(float out[3], rt::vec3 v) {
    out[0] = v.x;
    out[1] = v.y;
    out[2] = v.z;
}

}  // namespace

extern "C" {

int rt_host_load_scene(const char* json_path, rt_host_scene** out) {
    return guarded([&] {
        auto s = std::make_unique<rt_host_scene>();
        std::tie(s->options, s->cam, s->sc) = load_scene(json_path);
        finish(*s);
        *out = s.release();
    });
}

int rt_host_load_snapshot(const void* bytes, size_t size, rt_host_scene** out) {
    return guarded([&] {
        auto s = std::make_unique<rt_host_scene>();
        std::tie(s->options, s->cam, s->sc) = load_scene_snapshot(bytes, size);
        finish(*s);
        *out = s.release();
    });
}

int rt_host_load_snapshot_file(const char* path, rt_host_scene** out) {
    return guarded([&] {
        auto s = std::make_unique<rt_host_scene>();
        std::tie(s->options, s->cam, s->sc) = load_scene_snapshot_file(path);
        finish(*s);
        *out = s.release();
    });
}

int rt_host_save_snapshot(const rt_host_scene* s, const char* path) {
    return guarded([&] { save_scene_snapshot(path, s->options, s->cam, s->sc); });
}

int rt_host_scene_desc(const rt_host_scene* s, rt_scene_desc* out) {
    return guarded([&] {
        rt_scene_desc d{};
        d.tlas = s->accel.tlas.data();
        d.blas = s->accel.blas.data();
        d.meshes = s->accel.meshes.data();
        d.instances = s->accel.instances.data();
        d.triangles = s->accel.triangles.data();
        d.transforms = reinterpret_cast<rt_mat4 const*>(s->sc.transformation.data());
        d.inverse_transforms = reinterpret_cast<rt_mat4 const*>(s->inverse_transformation.data());
        d.materials = s->sc.materials.data();
        d.mediums = s->sc.mediums.data();
        d.lights = s->sc.lights.data();
        d.textures = s->texture_table.data();
        d.tlas_count = (uint32_t) s->accel.tlas.size();
        d.blas_count = (uint32_t) s->accel.blas.size();
        d.mesh_count = (uint32_t) s->accel.meshes.size();
        d.instance_count = (uint32_t) s->accel.instances.size();
        d.triangle_count = (uint32_t) s->accel.triangles.size();
        d.transform_count = (uint32_t) s->sc.transformation.size();
        d.material_count = (uint32_t) s->sc.materials.size();
        d.medium_count = (uint32_t) s->sc.mediums.size();
        d.light_count = s->light_count;
        d.texture_count = (uint32_t) s->texture_table.size();
        d.sky_light = s->sky_light;
        *out = d;
    });
}

int rt_host_scene_info(const rt_host_scene* s, rt_host_info* out) {
    return guarded([&] {
        rt_host_info i{};
        i.resolution_x = s->options.resolution_x;
        i.resolution_y = s->options.resolution_y;
        i.max_depth = s->options.max_depth;
        i.tile_width = s->options.tile_width;
        i.tile_height = s->options.tile_height;
        i.tlas_depth = bvh_depth(s->accel.tlas, 0);
        for (rt_mesh const& m : s->accel.meshes) {
            i.blas_depth = std::max(i.blas_depth, bvh_depth(s->accel.blas, m.bvh_start));
        }
        i.vertex_count = (uint32_t) s->sc.vertices.size();
        copy3(i.cam_position, s->cam.position);
        copy3(i.cam_front, s->cam.front);
        copy3(i.cam_right, s->cam.right);
        copy3(i.cam_up, s->cam.up);
        i.cam_fov = s->cam.fov;
        *out = i;
    });
}

int rt_host_camera(const rt_host_scene* s, uint32_t width, uint32_t height, rt_camera* out) {
    return guarded([&] { *out = get_glsl_raytracer_camera(s->cam, width, height); });
}

int rt_host_make_camera(const float lookfrom[3], const float lookat[3], float fov,
    uint32_t width, uint32_t height, rt_camera* out) {
    return guarded([&] {
        camera const cam = create_camera(rt::vec3{lookfrom[0], lookfrom[1], lookfrom[2]},
            rt::vec3{lookat[0], lookat[1], lookat[2]}, fov);
        *out = get_glsl_raytracer_camera(cam, width, height);
    });
}

int rt_host_seed_schedule(uint32_t seed, uint64_t first, uint64_t count, uint32_t* out) {
    return guarded([&] {
        std::mt19937 gen{seed};
        gen.discard(first);
        for (uint64_t i = 0; i < count; ++i) {
            out[i] = (uint32_t) gen();
        }
    });
}

void rt_host_free_scene(rt_host_scene* s) {
    delete s;
}

const char* rt_host_last_error(void) {
    return last_error.c_str();
}

}  // extern "C"
