// Scene JSON loader: the host-side producer of every buffer the compute pass
// binds.  Behaviour follows src/asset/scene.cpp:34-400 of the reference,
// including the details that change device-visible numbering or values:
//   * JSON objects iterate sorted by key (materials, primitives, lights);
//   * transform = translate * rotate(quat w,x,y,z) * scale          (:221-246)
//   * the texture cache is looked up by full path but filled by relative
//     path, so a texture used twice is loaded twice                 (:91-107)
//   * area light direction.x = sum of object-space triangle areas
//     times scale.x*scale.y*scale.z                                 (:292-319)
//   * the sky light's total luminance loop walks i = 0, 4, 8, ... below
//     width*height and reads texel components at [i, i+1, i+2], i.e. it
//     covers only the first quarter of the image                    (:369-386)
#include "scene.h"

#include <cstring>
#include <filesystem>
#include <fstream>
#include <sstream>
#include <unordered_map>

#include "json.h"

namespace {

float luminance(float r, float g, float b) {
    return 0.212671f * r + 0.715160f * g + 0.072169f * b;
}

rt::vec3 read_vec3(rt::json_value const& v, char const* key) {
    std::string const k{key};
    return rt::vec3{
        v.at(k + "/0").as_float(),
        v.at(k + "/1").as_float(),
        v.at(k + "/2").as_float(),
    };
}

struct placed_transform {
    rt::mat4 matrix = rt::mat4::identity();
    float volume_scale = 1.0f;  // scale.x * scale.y * scale.z, 1 when absent
};

// Shared by primitives and area lights: optional position / rotation / scale.
placed_transform read_transform(rt::json_value const& v) {
    placed_transform out;
    if (v.contains("position")) {
        out.matrix = rt::translate(out.matrix, read_vec3(v, "position"));
    }
    if (v.contains("rotation")) {
        float const w = v.at("rotation/0").as_float();
        float const x = v.at("rotation/1").as_float();
        float const y = v.at("rotation/2").as_float();
        float const z = v.at("rotation/3").as_float();
        out.matrix = rt::mul(out.matrix, rt::quat_to_mat4(w, x, y, z));
    }
    if (v.contains("scale")) {
        rt::vec3 const s = read_vec3(v, "scale");
        out.matrix = rt::scale(out.matrix, s);
        out.volume_scale = s.x * s.y * s.z;
    }
    return out;
}

float triangle_area(vertex const& a, vertex const& b, vertex const& c) {
    rt::vec3 const pa{a.position_texu[0], a.position_texu[1], a.position_texu[2]};
    rt::vec3 const pb{b.position_texu[0], b.position_texu[1], b.position_texu[2]};
    rt::vec3 const pc{c.position_texu[0], c.position_texu[1], c.position_texu[2]};
    return 0.5f * rt::length(rt::cross(pb - pa, pc - pa));
}

class scene_builder {
public:
    explicit scene_builder(std::filesystem::path dir) : dir_{std::move(dir)} {}

    scene result;

    uint32_t mesh_id(std::string const& rel_path) {
        if (auto const it = mesh_ids_.find(rel_path); it != mesh_ids_.end()) {
            return it->second;
        }
        std::vector<vertex> const verts = load_mesh((dir_ / rel_path).string());
        result.mesh_vertex_start.push_back((uint32_t) result.vertices.size());
        result.vertices.insert(result.vertices.end(), verts.begin(), verts.end());
        uint32_t const id = (uint32_t) result.mesh_vertex_start.size() - 1;
        mesh_ids_[rel_path] = id;
        return id;
    }

    int32_t texture_id(std::string const& rel_path) {
        std::string const full = (dir_ / rel_path).string();
        // Lookup key and insert key differ on purpose (see file header).
        if (auto const it = texture_ids_.find(full); it != texture_ids_.end()) {
            return it->second;
        }
        result.textures.push_back(get_texture_data(full));
        int32_t const id = (int32_t) result.textures.size() - 1;
        texture_ids_[rel_path] = id;
        return id;
    }

    void add_materials(rt::json_value const& root) {
        for (auto const& [name, v] : root.at("material").object) {
            material m{};
            m.albedo[0] = v.value("albedo/0", 1.0f);
            m.albedo[1] = v.value("albedo/1", 1.0f);
            m.albedo[2] = v.value("albedo/2", 1.0f);
            m.emission[0] = v.value("emission/0", 0.0f);
            m.emission[1] = v.value("emission/1", 0.0f);
            m.emission[2] = v.value("emission/2", 0.0f);
            m.metallic = v.value("metallic", 0.0f);
            m.spec_trans = v.value("spec_trans", 0.0f);
            m.ior = v.value("ior", 1.5f);
            m.subsurface = v.value("subsurface", 0.0f);
            m.roughness = v.value("roughness", 0.5f);
            m.specular_tint = v.value("specular_tint", 0.0f);
            m.anisotropic = v.value("anisotropic", 0.0f);
            m.sheen = v.value("sheen", 0.0f);
            m.sheen_tint = v.value("sheen_tint", 0.5f);
            m.clearcoat = v.value("clearcoat", 0.0f);
            m.clearcoat_gloss = v.value("clearcoat_gloss", 1.0f);
            // Textures load in this fixed order: albedo, emission, normal, metallic-roughness.
            char const* const keys[4] = {
                "albedo_tex", "emission_tex", "normal_tex", "metallic_roughness_tex"};
            int32_t ids[4] = {-1, -1, -1, -1};
            for (int i = 0; i < 4; ++i) {
                std::string const file = v.value(keys[i], "");
                if (!file.empty()) {
                    ids[i] = texture_id(file);
                }
            }
            m.albedo_tex = ids[0];
            m.emission_tex = ids[1];
            m.normal_tex = ids[2];
            m.metallic_roughness_tex = ids[3];
            result.materials.push_back(m);
            material_ids_[name] = (int32_t) result.materials.size() - 1;
        }
    }

    void add_mediums(rt::json_value const& root) {
        // JSON group -> medium type id (shaders/common/material.glsl:26-28).
        std::pair<char const*, int32_t> const groups[3] = {
            {"medium/absorption", 0}, {"medium/emission", 1}, {"medium/anisotropic", 2}};
        for (auto const& [path, type] : groups) {
            rt::json_value const* group = root.at_path(path);
            if (!group) {
                continue;
            }
            for (auto const& [name, v] : group->object) {
                medium m{};
                rt::vec3 const color = read_vec3(v, "color");
                m.color[0] = color.x;
                m.color[1] = color.y;
                m.color[2] = color.z;
                m.density = v.at("density").as_float();
                m.anisotropic = v.at("anisotropic").as_float();
                m.type = type;
                result.mediums.push_back(m);
                medium_ids_[name] = (int32_t) result.mediums.size() - 1;
            }
        }
    }

    void add_primitives(rt::json_value const& root) {
        for (auto const& [name, v] : root.at("primitive").object) {
            primitive p{};
            p.mesh = mesh_id(v.at("mesh").as_string());
            if (v.contains("material")) {
                p.material = lookup(material_ids_, v.at("material").as_string(), "material");
            }
            if (v.contains("medium")) {
                p.medium = lookup(medium_ids_, v.at("medium").as_string(), "medium");
            }
            result.transformation.push_back(read_transform(v).matrix);
            p.transform = (int32_t) result.transformation.size() - 1;
            result.primitives.push_back(p);
        }
    }

    void add_area_lights(rt::json_value const& root) {
        rt::json_value const* group = root.at_path("light/area");
        if (!group) {
            return;
        }
        for (auto const& [name, v] : group->object) {
            light l{};
            rt::vec3 const intensity = read_vec3(v, "intensity");
            l.intensity[0] = intensity.x;
            l.intensity[1] = intensity.y;
            l.intensity[2] = intensity.z;
            bool const two_sided = v.value("two_sided", false);
            l.emission_tex = -1;
            if (v.contains("emission_tex")) {
                l.emission_tex = texture_id(v.at("emission_tex").as_string());
            }
            uint32_t const mesh = mesh_id(v.at("mesh").as_string());
            placed_transform const placed = read_transform(v);
            uint32_t const first = result.mesh_vertex_start[mesh];
            uint32_t const end = (mesh + 1 == result.mesh_vertex_start.size()) ?
                                     (uint32_t) result.vertices.size() :
                                     result.mesh_vertex_start[mesh + 1];
            float total_area = 0.0f;
            for (uint32_t t = first / 3; t < end / 3; ++t) {
                total_area += triangle_area(result.vertices.at(3 * t + 0),
                    result.vertices.at(3 * t + 1), result.vertices.at(3 * t + 2));
            }
            total_area *= placed.volume_scale;
            result.transformation.push_back(placed.matrix);
            l.direction[0] = total_area;
            l.direction[1] = 0.0f;
            l.direction[2] = 0.0f;
            l.type = two_sided ? RT_LIGHT_AREA_DOUBLE_SIDED : RT_LIGHT_AREA_SINGLE_SIDED;
            l.mesh = mesh;
            l.transform = (int32_t) result.transformation.size() - 1;
            result.lights.push_back(l);
        }
    }

    void add_distant_lights(rt::json_value const& root) {
        rt::json_value const* group = root.at_path("light/distant");
        if (!group) {
            return;
        }
        for (auto const& [name, v] : group->object) {
            light l{};
            rt::vec3 const intensity = read_vec3(v, "intensity");
            rt::vec3 const direction = read_vec3(v, "direction");
            l.intensity[0] = intensity.x;
            l.intensity[1] = intensity.y;
            l.intensity[2] = intensity.z;
            l.emission_tex = -1;
            l.direction[0] = direction.x;
            l.direction[1] = direction.y;
            l.direction[2] = direction.z;
            l.type = RT_LIGHT_DISTANT;
            l.mesh = 0;
            l.transform = -1;
            result.lights.push_back(l);
        }
    }

    void add_sky_light(rt::json_value const& root) {
        rt::json_value const* sky = root.at_path("sky_light");
        if (!sky) {
            return;
        }
        light l{};
        rt::vec3 const intensity = read_vec3(*sky, "intensity");
        l.intensity[0] = intensity.x;
        l.intensity[1] = intensity.y;
        l.intensity[2] = intensity.z;
        l.emission_tex = -1;
        float total_luminance = 0.0f;
        uint32_t tex_w = 0;
        uint32_t tex_h = 0;
        std::string const env_file = sky->value("environment_tex", "");
        if (!env_file.empty()) {
            l.emission_tex = texture_id(env_file);
            texture_data const& tex = result.textures.back();
            tex_w = tex.width;
            tex_h = tex.height;
            uint32_t const limit = tex.width * tex.height;
            uint32_t const channels = 4;
            if (tex.format == texture_format::sfloat) {
                float const* p = reinterpret_cast<float const*>(tex.data.data());
                for (uint32_t i = 0; i < limit; i += channels) {
                    total_luminance += luminance(p[i + 0], p[i + 1], p[i + 2]);
                }
            } else {
                uint8_t const* p = tex.data.data();
                for (uint32_t i = 0; i < limit; i += channels) {
                    total_luminance += luminance((float) p[i + 0] / 255.0f,
                        (float) p[i + 1] / 255.0f, (float) p[i + 2] / 255.0f);
                }
            }
        }
        l.direction[0] = total_luminance;
        l.direction[1] = (float) tex_w;
        l.direction[2] = (float) tex_h;
        l.type = RT_LIGHT_SKY;
        l.mesh = 0;
        l.transform = -1;
        result.lights.push_back(l);
    }

private:
    static int32_t lookup(std::unordered_map<std::string, int32_t> const& table,
        std::string const& name, char const* what) {
        auto const it = table.find(name);
        if (it == table.end()) {
            throw std::runtime_error(std::string{"scene: unknown "} + what + " '" + name + "'");
        }
        return it->second;
    }

    std::filesystem::path dir_;
    std::unordered_map<std::string, uint32_t> mesh_ids_;
    std::unordered_map<std::string, int32_t> texture_ids_;
    std::unordered_map<std::string, int32_t> material_ids_;
    std::unordered_map<std::string, int32_t> medium_ids_;
};

}  // namespace

std::tuple<render_options, camera, scene> load_scene(std::string_view file_path) {
    std::ifstream ifs{std::string{file_path}};
    if (!ifs) {
        throw std::runtime_error("scene: cannot open " + std::string{file_path});
    }
    std::stringstream buffer;
    buffer << ifs.rdbuf();
    rt::json_value const root = rt::parse_json(buffer.str());

    render_options options{};
    options.resolution_x = root.at("renderer/resolution/0").as_uint();
    options.resolution_y = root.at("renderer/resolution/1").as_uint();
    options.max_depth = root.at("renderer/max_depth").as_uint();
    options.tile_width = root.at("renderer/tile/0").as_uint();
    options.tile_height = root.at("renderer/tile/1").as_uint();
    if (options.tile_width == 0 || options.resolution_x % options.tile_width != 0) {
        throw std::runtime_error("Window width isn't divisible by tile width");
    }
    if (options.tile_height == 0 || options.resolution_y % options.tile_height != 0) {
        throw std::runtime_error("Window height isn't divisible by tile height");
    }

    camera const cam = create_camera(read_vec3(root, "camera/lookfrom"),
        read_vec3(root, "camera/lookat"), root.at("camera/fov").as_float());

    scene_builder builder{std::filesystem::path{std::string{file_path}}.parent_path()};
    builder.add_materials(root);
    builder.add_mediums(root);
    builder.add_primitives(root);
    builder.add_area_lights(root);
    builder.add_distant_lights(root);
    builder.add_sky_light(root);
    return std::make_tuple(options, cam, std::move(builder.result));
}
