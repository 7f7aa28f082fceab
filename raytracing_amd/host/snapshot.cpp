// Scene snapshots: (render_options, camera, scene) serialised as one
// zlib-compressed blob.  A snapshot holds exactly what load_scene() returns,
// so everything downstream (BVH build, upload, render) runs unchanged on
// machines that do not have the original OBJ / PNG / HDR files.
//
// Layout: 8-byte magic "RTSNAP01", u64 raw size, then the deflated payload.
// Payload: render_options (5 x u32), camera (17 x f32), then for each array a
// u64 element count followed by the elements' raw bytes, in the order
// vertices, mesh_vertex_start, materials, mediums, transformation, primitives,
// lights, textures (each: width, height, format, u64 byte count, bytes).
#include <zlib.h>

#include <cstring>
#include <fstream>
#include <stdexcept>

#include "scene.h"

namespace {

char const magic[8] = {'R', 'T', 'S', 'N', 'A', 'P', '0', '1'};

struct writer {
    std::vector<uint8_t> bytes;
    void raw(void const* p, size_t n) {
        uint8_t const* b = static_cast<uint8_t const*>(p);
        bytes.insert(bytes.end(), b, b + n);
    }
    void u32(uint32_t v) { raw(&v, 4); }
    void u64(uint64_t v) { raw(&v, 8); }
    void f32(float v) { raw(&v, 4); }
    template <typename T>
    void array(std::vector<T> const& v) {
        u64(v.size());
        raw(v.data(), v.size() * sizeof(T));
    }
};

struct reader {
    uint8_t const* p;
    uint8_t const* end;
    void raw(void* out, size_t n) {
        if ((size_t) (end - p) < n) {
            throw std::runtime_error("snapshot: truncated payload");
        }
        std::memcpy(out, p, n);
        p += n;
    }
    uint32_t u32() {
        uint32_t v;
        raw(&v, 4);
        return v;
    }
    uint64_t u64() {
        uint64_t v;
        raw(&v, 8);
        return v;
    }
    float f32() {
        float v;
        raw(&v, 4);
        return v;
    }
    template <typename T>
    void array(std::vector<T>& v) {
        uint64_t const n = u64();
        if (n > (uint64_t) (end - p) / sizeof(T)) {
            throw std::runtime_error("snapshot: array larger than payload");
        }
        v.resize((size_t) n);
        raw(v.data(), v.size() * sizeof(T));
    }
};

void put_vec3(writer& w, rt::vec3 v) {
    w.f32(v.x);
    w.f32(v.y);
    w.f32(v.z);
}

rt::vec3 get_vec3(reader& r) {
    rt::vec3 v;
    v.x = r.f32();
    v.y = r.f32();
    v.z = r.f32();
    return v;
}

}  // namespace

void save_scene_snapshot(std::string const& path, render_options const& options,
    camera const& cam, scene const& scene) {
    writer w;
    w.u32(options.resolution_x);
    w.u32(options.resolution_y);
    w.u32(options.max_depth);
    w.u32(options.tile_width);
    w.u32(options.tile_height);
    put_vec3(w, cam.position);
    put_vec3(w, cam.front);
    put_vec3(w, cam.right);
    put_vec3(w, cam.up);
    w.f32(cam.pitch);
    w.f32(cam.yaw);
    w.f32(cam.sensitivity);
    w.f32(cam.velocity);
    w.f32(cam.fov);
    w.array(scene.vertices);
    w.array(scene.mesh_vertex_start);
    w.array(scene.materials);
    w.array(scene.mediums);
    w.array(scene.transformation);
    w.array(scene.primitives);
    w.array(scene.lights);
    w.u64(scene.textures.size());
    for (texture_data const& t : scene.textures) {
        w.u32(t.width);
        w.u32(t.height);
        w.u32((uint32_t) t.format);
        w.array(t.data);
    }
    uLongf packed_len = compressBound((uLong) w.bytes.size());
    std::vector<uint8_t> packed(packed_len);
    if (compress2(packed.data(), &packed_len, w.bytes.data(), (uLong) w.bytes.size(), 9) !=
        Z_OK) {
        throw std::runtime_error("snapshot: compression failed");
    }
    std::ofstream ofs{path, std::ios::binary};
    if (!ofs) {
        throw std::runtime_error("snapshot: cannot write " + path);
    }
    uint64_t const raw_len = w.bytes.size();
    ofs.write(magic, 8);
    ofs.write(reinterpret_cast<char const*>(&raw_len), 8);
    ofs.write(reinterpret_cast<char const*>(packed.data()), (std::streamsize) packed_len);
}

std::tuple<render_options, camera, scene> load_scene_snapshot(void const* bytes, size_t size) {
    uint8_t const* in = static_cast<uint8_t const*>(bytes);
    if (size < 16 || std::memcmp(in, magic, 8) != 0) {
        throw std::runtime_error("snapshot: bad magic");
    }
    uint64_t raw_len = 0;
    std::memcpy(&raw_len, in + 8, 8);
    std::vector<uint8_t> raw((size_t) raw_len);
    uLongf got = (uLongf) raw.size();
    if (uncompress(raw.data(), &got, in + 16, (uLong) (size - 16)) != Z_OK || got != raw.size()) {
        throw std::runtime_error("snapshot: corrupt payload");
    }
    reader r{raw.data(), raw.data() + raw.size()};
    render_options options{};
    options.resolution_x = r.u32();
    options.resolution_y = r.u32();
    options.max_depth = r.u32();
    options.tile_width = r.u32();
    options.tile_height = r.u32();
    camera cam{};
    cam.position = get_vec3(r);
    cam.front = get_vec3(r);
    cam.right = get_vec3(r);
    cam.up = get_vec3(r);
    cam.pitch = r.f32();
    cam.yaw = r.f32();
    cam.sensitivity = r.f32();
    cam.velocity = r.f32();
    cam.fov = r.f32();
    scene sc;
    r.array(sc.vertices);
    r.array(sc.mesh_vertex_start);
    r.array(sc.materials);
    r.array(sc.mediums);
    r.array(sc.transformation);
    r.array(sc.primitives);
    r.array(sc.lights);
    uint64_t const texture_total = r.u64();
    for (uint64_t i = 0; i < texture_total; ++i) {
        texture_data t;
        t.width = r.u32();
        t.height = r.u32();
        t.format = (texture_format) r.u32();
        r.array(t.data);
        size_t const texel = t.format == texture_format::sfloat ? 16 : 4;
        if (t.data.size() != (size_t) t.width * t.height * texel) {
            throw std::runtime_error("snapshot: texture size mismatch");
        }
        sc.textures.push_back(std::move(t));
    }
    return std::make_tuple(options, cam, std::move(sc));
}

std::tuple<render_options, camera, scene> load_scene_snapshot_file(std::string const& path) {
    std::ifstream ifs{path, std::ios::binary};
    if (!ifs) {
        throw std::runtime_error("snapshot: cannot open " + path);
    }
    std::vector<uint8_t> bytes{std::istreambuf_iterator<char>(ifs), std::istreambuf_iterator<char>()};
    return load_scene_snapshot(bytes.data(), bytes.size());
}
