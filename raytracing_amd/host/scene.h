// Host-side scene model: the inputs of the renderer plugin table.
//
// Mirrors the reference's host types by name and meaning so a reference
// maintainer can map one onto the other:
//   render_options  src/renderer/render_options.h:1-11
//   camera          src/asset/camera.h:18-32
//   primitive       src/asset/mesh.h:22-27
//   texture_data    src/asset/texture.h:16-22
//   scene           src/asset/scene.h:10-22
// Device-visible element types (vertex, material, medium, light) are the PODs
// of include/rt_types.h, which are byte-identical to the reference structs.
#pragma once

#include <cstdint>
#include <string>
#include <string_view>
#include <tuple>
#include <vector>

#include "rt_types.h"
#include "vecmath.h"

struct render_options {
    uint32_t resolution_x = 0;
    uint32_t resolution_y = 0;
    uint32_t max_depth = 0;
    uint32_t tile_width = 0;
    uint32_t tile_height = 0;
};

struct camera {
    rt::vec3 position;
    rt::vec3 front;
    rt::vec3 right;
    rt::vec3 up;
    float pitch = 0.0f;
    float yaw = 0.0f;
    float sensitivity = 0.01f;
    float velocity = 1.0f;
    float fov = 45.0f;
    bool dirty = false;
};

using vertex = rt_vertex;
using material = rt_material;
using medium = rt_medium;
using light = rt_light;

struct primitive {
    uint32_t mesh = 0;
    int32_t transform = -1;
    int32_t material = -1;
    int32_t medium = -1;
};

enum class texture_format : uint32_t {
    unorm = RT_TEX_UNORM8X4,
    sfloat = RT_TEX_F32X4,
};

// Always 4 channels (the reference forces RGBA on load, src/asset/texture.cpp:13).
// Pixel storage is owned by the scene (std::vector), not malloc'ed.
struct texture_data {
    uint32_t width = 0;
    uint32_t height = 0;
    texture_format format = texture_format::unorm;
    std::vector<uint8_t> data;
};

struct scene {
    std::vector<vertex> vertices;
    std::vector<uint32_t> mesh_vertex_start;

    std::vector<texture_data> textures;
    std::vector<material> materials;
    std::vector<medium> mediums;

    std::vector<rt::mat4> transformation;
    std::vector<primitive> primitives;

    std::vector<light> lights;
};

// Parses a scene JSON (README "scene format") together with the OBJ meshes and
// textures it names, resolved relative to the JSON's directory.
// Throws std::runtime_error on any failure.
std::tuple<render_options, camera, scene> load_scene(std::string_view file_path);

// src/asset/camera.cpp:26-36
camera create_camera(rt::vec3 const& lookfrom, rt::vec3 const& lookat, float fov);

// src/asset/camera.cpp:66-88: pinhole camera, focal length 1, vertical fov,
// pixel_delta_v pointing down the image.
rt_camera get_glsl_raytracer_camera(
    camera const& camera, uint32_t frame_width, uint32_t frame_height);

// src/asset/mesh.cpp:9-59.  Polygons are fan-triangulated (0,k,k+1).
std::vector<vertex> load_mesh(std::string_view file_path);

// src/asset/texture.cpp:10-27.  PNG (8/16-bit, non-interlaced) and Radiance
// .hdr files; always expanded to 4 channels.
texture_data get_texture_data(std::string_view texture_path);

// Binary snapshot of (options, camera, scene): the exact inputs of
// initialize()/prepare_data(), zlib-compressed.  Lets tests and benchmarks run
// where the original asset files are not available.
void save_scene_snapshot(std::string const& path, render_options const& options,
    camera const& camera, scene const& scene);
std::tuple<render_options, camera, scene> load_scene_snapshot(
    void const* bytes, size_t size);
std::tuple<render_options, camera, scene> load_scene_snapshot_file(
    std::string const& path);
