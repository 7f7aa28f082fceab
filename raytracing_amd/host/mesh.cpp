// Wavefront OBJ reader producing the flat, un-indexed triangle list the
// renderer consumes (three vertices per triangle, 32 bytes per vertex).
//
// Follows the reference's use of tinyobjloader (src/asset/mesh.cpp:9-59):
//   * faces are emitted in file order, every polygon triangulated;
//   * normals are mandatory (the reference indexes them unconditionally);
//   * a file with no `vt` records at all gets synthetic texture coordinates
//     (0,0), (0,1), (1,1) for the three corners of each triangle.
// tinyobjloader is an un-vendored submodule with no pinned revision; polygons
// are split as a fan (0,k,k+1), which is what its 1.x releases do.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <string>

#include "scene.h"

namespace {

struct corner {
    int v = 0, vt = 0, vn = 0;  // resolved 0-based indices, -1 when absent
};

// Parses one "v", "v/vt", "v//vn" or "v/vt/vn" token; OBJ indices are 1-based
// and may be negative (relative to the records seen so far).
corner parse_corner(char const*& p, size_t nv, size_t nvt, size_t nvn) {
    auto resolve = [](long idx, size_t count) -> int {
        if (idx > 0) {
            return (int) idx - 1;
        }
        if (idx < 0) {
            return (int) count + (int) idx;
        }
        return -1;
    };
    corner c{-1, -1, -1};
    char* end = nullptr;
    c.v = resolve(std::strtol(p, &end, 10), nv);
    p = end;
    if (*p == '/') {
        ++p;
        if (*p != '/') {
            c.vt = resolve(std::strtol(p, &end, 10), nvt);
            p = end;
        }
        if (*p == '/') {
            ++p;
            c.vn = resolve(std::strtol(p, &end, 10), nvn);
            p = end;
        }
    }
    return c;
}

void skip_blank(char const*& p) {
    while (*p == ' ' || *p == '\t') {
        ++p;
    }
}

}  // namespace

std::vector<vertex> load_mesh(std::string_view file_path) {
    std::ifstream ifs{std::string{file_path}};
    if (!ifs) {
        throw std::runtime_error("Can't load obj file " + std::string{file_path});
    }
    std::vector<float> positions, normals, texcoords;
    std::vector<std::vector<corner>> faces;
    std::string line;
    while (std::getline(ifs, line)) {
        if (!line.empty() && line.back() == '\r') {
            line.pop_back();
        }
        char const* p = line.c_str();
        skip_blank(p);
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            char* end = nullptr;
            for (int i = 0; i < 3; ++i) {
                positions.push_back((float) std::strtod(p, &end));
                p = end;
            }
        } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3;
            char* end = nullptr;
            for (int i = 0; i < 3; ++i) {
                normals.push_back((float) std::strtod(p, &end));
                p = end;
            }
        } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
            p += 3;
            char* end = nullptr;
            for (int i = 0; i < 2; ++i) {
                texcoords.push_back((float) std::strtod(p, &end));
                p = end;
            }
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            p += 2;
            std::vector<corner> face;
            while (true) {
                skip_blank(p);
                if (*p == '\0') {
                    break;
                }
                face.push_back(parse_corner(
                    p, positions.size() / 3, texcoords.size() / 2, normals.size() / 3));
            }
            if (face.size() >= 3) {
                faces.push_back(std::move(face));
            }
        }
    }
    bool const no_texcoords = texcoords.empty();
    std::vector<vertex> out;
    auto emit = [&](corner const& c, int corner_in_triangle) {
        if (c.v < 0 || (size_t) c.v * 3 + 2 >= positions.size()) {
            throw std::runtime_error("obj: bad vertex index in " + std::string{file_path});
        }
        if (c.vn < 0 || (size_t) c.vn * 3 + 2 >= normals.size()) {
            throw std::runtime_error(
                "obj: face without a valid normal in " + std::string{file_path});
        }
        float tu = 0.0f;
        float tv = 0.0f;
        if (no_texcoords) {
            tu = corner_in_triangle == 2 ? 1.0f : 0.0f;
            tv = corner_in_triangle == 0 ? 0.0f : 1.0f;
        } else {
            if (c.vt < 0 || (size_t) c.vt * 2 + 1 >= texcoords.size()) {
                throw std::runtime_error(
                    "obj: face without a valid texcoord in " + std::string{file_path});
            }
            tu = texcoords[2 * (size_t) c.vt + 0];
            tv = texcoords[2 * (size_t) c.vt + 1];
        }
        vertex v{};
        v.position_texu[0] = positions[3 * (size_t) c.v + 0];
        v.position_texu[1] = positions[3 * (size_t) c.v + 1];
        v.position_texu[2] = positions[3 * (size_t) c.v + 2];
        v.position_texu[3] = tu;
        v.normal_texv[0] = normals[3 * (size_t) c.vn + 0];
        v.normal_texv[1] = normals[3 * (size_t) c.vn + 1];
        v.normal_texv[2] = normals[3 * (size_t) c.vn + 2];
        v.normal_texv[3] = tv;
        out.push_back(v);
    };
    for (auto const& face : faces) {
        for (size_t k = 1; k + 1 < face.size(); ++k) {
            emit(face[0], 0);
            emit(face[k], 1);
            emit(face[k + 1], 2);
        }
    }
    return out;
}
