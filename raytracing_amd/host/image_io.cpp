#include "image_io.h"

#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <fstream>
#include <stdexcept>

std::vector<uint8_t> to_display_rgb8(
    std::vector<float> const& rgba, uint32_t width, uint32_t height, uint32_t passes) {
    float const frame_scalar = passes == 0 ? 1.0f : 1.0f / (float) passes;
    std::vector<uint8_t> out((size_t) width * height * 3);
    for (size_t i = 0; i < (size_t) width * height; ++i) {
        for (int c = 0; c < 3; ++c) {
            float v = frame_scalar * rgba[4 * i + c];
            v = v / (v + 1.0f);
            v = std::pow(std::max(v, 0.0f), 1.0f / 2.2f);
            out[3 * i + c] = (uint8_t) std::lround(std::min(std::max(v, 0.0f), 1.0f) * 255.0f);
        }
    }
    return out;
}

namespace {

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t) (x >> 24));
    v.push_back((uint8_t) (x >> 16));
    v.push_back((uint8_t) (x >> 8));
    v.push_back((uint8_t) x);
}

void put_chunk(std::vector<uint8_t>& file, char const type[4], std::vector<uint8_t> const& body) {
    put_be32(file, (uint32_t) body.size());
    size_t const start = file.size();
    file.insert(file.end(), type, type + 4);
    file.insert(file.end(), body.begin(), body.end());
    uint32_t const crc = (uint32_t) crc32(0L, file.data() + start, (uInt) (file.size() - start));
    put_be32(file, crc);
}

}  // namespace

void write_png_rgb8(std::string const& path, std::vector<uint8_t> const& rgb, uint32_t width,
    uint32_t height) {
    if (rgb.size() != (size_t) width * height * 3) {
        throw std::runtime_error("png: pixel buffer size does not match the image size");
    }
    std::vector<uint8_t> raw;
    raw.reserve(((size_t) width * 3 + 1) * height);
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgb.begin() + (size_t) y * width * 3,
            rgb.begin() + (size_t) (y + 1) * width * 3);
    }
    uLongf packed_len = compressBound((uLong) raw.size());
    std::vector<uint8_t> packed(packed_len);
    if (compress2(packed.data(), &packed_len, raw.data(), (uLong) raw.size(), 6) != Z_OK) {
        throw std::runtime_error("png: compression failed");
    }
    packed.resize(packed_len);
    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> header;
    put_be32(header, width);
    put_be32(header, height);
    header.insert(header.end(), {8, 2, 0, 0, 0});  // 8 bit, RGB, deflate, adaptive, no interlace
    put_chunk(file, "IHDR", header);
    put_chunk(file, "IDAT", packed);
    put_chunk(file, "IEND", {});
    std::ofstream ofs{path, std::ios::binary};
    if (!ofs) {
        throw std::runtime_error("png: cannot write " + path);
    }
    ofs.write(reinterpret_cast<char const*>(file.data()), (std::streamsize) file.size());
}

void write_raw_accumulation(std::string const& path, std::vector<float> const& rgba,
    uint32_t width, uint32_t height, uint32_t passes) {
    std::ofstream ofs{path, std::ios::binary};
    if (!ofs) {
        throw std::runtime_error("raw: cannot write " + path);
    }
    ofs.write("RTACCUM1", 8);
    ofs.write(reinterpret_cast<char const*>(&width), 4);
    ofs.write(reinterpret_cast<char const*>(&height), 4);
    ofs.write(reinterpret_cast<char const*>(&passes), 4);
    ofs.write(reinterpret_cast<char const*>(rgba.data()), (std::streamsize) (rgba.size() * 4));
}
