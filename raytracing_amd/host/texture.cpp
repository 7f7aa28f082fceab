// Texture file decoding for the scene loader.
//
// The reference decodes with stb_image and always asks for 4 channels
// (src/asset/texture.cpp:10-27): 8-bit files become RGBA8 (sampled as UNORM,
// no sRGB decode), Radiance .hdr files become linear RGBA32F with alpha 1.
// stb_image is an un-vendored submodule, so the two formats the shipped
// scenes use are decoded here directly: PNG through zlib, and RGBE .hdr.
#include <zlib.h>

#include <cmath>
#include <cstring>
#include <fstream>
#include <stdexcept>

#include "scene.h"

namespace {

std::vector<uint8_t> read_file(std::string const& path) {
    std::ifstream ifs{path, std::ios::binary};
    if (!ifs) {
        throw std::runtime_error("Can't load texture " + path);
    }
    ifs.seekg(0, std::ios::end);
    std::streamoff const n = ifs.tellg();
    ifs.seekg(0, std::ios::beg);
    std::vector<uint8_t> bytes((size_t) n);
    ifs.read(reinterpret_cast<char*>(bytes.data()), n);
    return bytes;
}

uint32_t be32(uint8_t const* p) {
    return ((uint32_t) p[0] << 24) | ((uint32_t) p[1] << 16) | ((uint32_t) p[2] << 8) | p[3];
}

int paeth(int a, int b, int c) {
    int const p = a + b - c;
    int const pa = std::abs(p - a);
    int const pb = std::abs(p - b);
    int const pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) {
        return a;
    }
    return pb <= pc ? b : c;
}

texture_data decode_png(std::vector<uint8_t> const& file, std::string const& path) {
    auto fail = [&](char const* why) -> void {
        throw std::runtime_error("png (" + path + "): " + why);
    };
    if (file.size() < 8 + 25) {
        fail("truncated");
    }
    uint32_t width = 0, height = 0;
    int depth = 0, color = 0, interlace = 0;
    std::vector<uint8_t> idat;
    std::vector<uint8_t> palette;          // rgb triples
    std::vector<uint8_t> palette_alpha;    // tRNS for palettes
    bool has_key = false;
    uint16_t key[3] = {0, 0, 0};
    size_t pos = 8;
    while (pos + 12 <= file.size()) {
        uint32_t const len = be32(&file[pos]);
        char const* type = reinterpret_cast<char const*>(&file[pos + 4]);
        uint8_t const* body = &file[pos + 8];
        if (pos + 12 + (size_t) len > file.size()) {
            fail("chunk overruns file");
        }
        if (std::memcmp(type, "IHDR", 4) == 0) {
            width = be32(body);
            height = be32(body + 4);
            depth = body[8];
            color = body[9];
            interlace = body[12];
        } else if (std::memcmp(type, "PLTE", 4) == 0) {
            palette.assign(body, body + len);
        } else if (std::memcmp(type, "tRNS", 4) == 0) {
            if (color == 3) {
                palette_alpha.assign(body, body + len);
            } else if (color == 0 && len >= 2) {
                has_key = true;
                key[0] = key[1] = key[2] = (uint16_t) ((body[0] << 8) | body[1]);
            } else if (color == 2 && len >= 6) {
                has_key = true;
                for (int i = 0; i < 3; ++i) {
                    key[i] = (uint16_t) ((body[2 * i] << 8) | body[2 * i + 1]);
                }
            }
        } else if (std::memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), body, body + len);
        } else if (std::memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t) len;
    }
    if (width == 0 || height == 0) {
        fail("missing IHDR");
    }
    if (interlace != 0) {
        fail("interlaced files are not supported");
    }
    int channels = 0;
    switch (color) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: fail("bad colour type");
    }
    size_t const bits_per_pixel = (size_t) channels * (size_t) depth;
    size_t const stride = (bits_per_pixel * width + 7) / 8;
    size_t const filter_bpp = bits_per_pixel >= 8 ? bits_per_pixel / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * height);
    uLongf raw_len = (uLongf) raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong) idat.size()) != Z_OK ||
        raw_len != raw.size()) {
        fail("zlib stream is corrupt");
    }
    // Undo the per-row prediction filters in place.
    std::vector<uint8_t> pixels(stride * height);
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t const filter = raw[(stride + 1) * y];
        uint8_t const* src = &raw[(stride + 1) * y + 1];
        uint8_t* dst = &pixels[stride * y];
        uint8_t const* up = y > 0 ? &pixels[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int const a = i >= filter_bpp ? dst[i - filter_bpp] : 0;
            int const b = up ? up[i] : 0;
            int const c = (up && i >= filter_bpp) ? up[i - filter_bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: pred = paeth(a, b, c); break;
                default: fail("bad filter type");
            }
            dst[i] = (uint8_t) (src[i] + pred);
        }
    }
    // Expand to RGBA8.  16-bit samples keep their high byte.
    texture_data out;
    out.width = width;
    out.height = height;
    out.format = texture_format::unorm;
    out.data.resize((size_t) width * height * 4);
    auto sample = [&](uint8_t const* row, size_t index) -> uint16_t {
        if (depth == 8) {
            return row[index];
        }
        if (depth == 16) {
            return (uint16_t) ((row[2 * index] << 8) | row[2 * index + 1]);
        }
        size_t const bit = index * (size_t) depth;
        uint8_t const byte = row[bit / 8];
        int const shift = 8 - depth - (int) (bit % 8);
        return (uint16_t) ((byte >> shift) & ((1 << depth) - 1));
    };
    auto to8 = [&](uint16_t v) -> uint8_t {
        if (depth == 16) {
            return (uint8_t) (v >> 8);
        }
        if (depth == 8) {
            return (uint8_t) v;
        }
        // 1/2/4-bit greyscale is scaled to the full range.
        return (uint8_t) (v * (255 / ((1 << depth) - 1)));
    };
    for (uint32_t y = 0; y < height; ++y) {
        uint8_t const* row = &pixels[stride * y];
        uint8_t* dst = &out.data[(size_t) y * width * 4];
        for (uint32_t x = 0; x < width; ++x, dst += 4) {
            if (color == 3) {
                uint16_t const idx = sample(row, x);
                if ((size_t) idx * 3 + 2 >= palette.size()) {
                    fail("palette index out of range");
                }
                dst[0] = palette[3 * idx + 0];
                dst[1] = palette[3 * idx + 1];
                dst[2] = palette[3 * idx + 2];
                dst[3] = idx < palette_alpha.size() ? palette_alpha[idx] : 255;
            } else if (color == 0 || color == 4) {
                uint16_t const g = sample(row, (size_t) x * channels);
                dst[0] = dst[1] = dst[2] = to8(g);
                dst[3] = color == 4 ? to8(sample(row, (size_t) x * channels + 1)) :
                                      ((has_key && g == key[0]) ? 0 : 255);
            } else {
                uint16_t const r = sample(row, (size_t) x * channels + 0);
                uint16_t const g = sample(row, (size_t) x * channels + 1);
                uint16_t const b = sample(row, (size_t) x * channels + 2);
                dst[0] = to8(r);
                dst[1] = to8(g);
                dst[2] = to8(b);
                dst[3] = color == 6 ? to8(sample(row, (size_t) x * channels + 3)) :
                                      ((has_key && r == key[0] && g == key[1] && b == key[2]) ?
                                              0 :
                                              255);
            }
        }
    }
    return out;
}

// Radiance RGBE.  value = mantissa * 2^(exponent - 136); exponent 0 means black.
texture_data decode_hdr(std::vector<uint8_t> const& file, std::string const& path) {
    auto fail = [&](char const* why) -> void {
        throw std::runtime_error("hdr (" + path + "): " + why);
    };
    size_t pos = 0;
    auto next_line = [&]() -> std::string {
        std::string s;
        while (pos < file.size() && file[pos] != '\n') {
            s.push_back((char) file[pos++]);
        }
        ++pos;
        return s;
    };
    std::string const magic = next_line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") {
        fail("not a Radiance file");
    }
    bool format_ok = false;
    while (pos < file.size()) {
        std::string const line = next_line();
        if (line.empty()) {
            break;
        }
        if (line == "FORMAT=32-bit_rle_rgbe") {
            format_ok = true;
        }
    }
    if (!format_ok) {
        fail("unsupported pixel format");
    }
    std::string const dims = next_line();
    int height = 0, width = 0;
    if (std::sscanf(dims.c_str(), "-Y %d +X %d", &height, &width) != 2 || width <= 0 ||
        height <= 0) {
        fail("unsupported orientation");
    }
    texture_data out;
    out.width = (uint32_t) width;
    out.height = (uint32_t) height;
    out.format = texture_format::sfloat;
    out.data.resize((size_t) width * height * 16);
    float* dst = reinterpret_cast<float*>(out.data.data());
    auto convert = [](float* o, uint8_t const* rgbe) {
        if (rgbe[3] != 0) {
            float const f = (float) std::ldexp(1.0f, (int) rgbe[3] - (128 + 8));
            o[0] = rgbe[0] * f;
            o[1] = rgbe[1] * f;
            o[2] = rgbe[2] * f;
        } else {
            o[0] = o[1] = o[2] = 0.0f;
        }
        o[3] = 1.0f;
    };
    auto need = [&](size_t n) {
        if (pos + n > file.size()) {
            fail("truncated pixel data");
        }
    };
    auto read_flat = [&](size_t first_pixel) {
        for (size_t i = first_pixel; i < (size_t) width * height; ++i) {
            need(4);
            convert(dst + 4 * i, &file[pos]);
            pos += 4;
        }
    };
    if (width < 8 || width >= 32768) {
        read_flat(0);
        return out;
    }
    std::vector<uint8_t> scanline((size_t) width * 4);
    for (int y = 0; y < height; ++y) {
        need(4);
        uint8_t const c1 = file[pos], c2 = file[pos + 1], len_hi = file[pos + 2];
        if (c1 != 2 || c2 != 2 || (len_hi & 0x80)) {
            // Not run-length encoded: the rest of the file is flat RGBE.
            if (y != 0) {
                fail("mixed encodings");
            }
            read_flat(0);
            return out;
        }
        int const len = (len_hi << 8) | file[pos + 3];
        pos += 4;
        if (len != width) {
            fail("corrupt scanline width");
        }
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < width) {
                need(1);
                int count = file[pos++];
                if (count > 128) {
                    count -= 128;
                    need(1);
                    uint8_t const value = file[pos++];
                    if (i + count > width) {
                        fail("run overflows scanline");
                    }
                    for (int z = 0; z < count; ++z) {
                        scanline[(size_t) (i++) * 4 + k] = value;
                    }
                } else {
                    if (count == 0 || i + count > width) {
                        fail("run overflows scanline");
                    }
                    need((size_t) count);
                    for (int z = 0; z < count; ++z) {
                        scanline[(size_t) (i++) * 4 + k] = file[pos++];
                    }
                }
            }
        }
        for (int x = 0; x < width; ++x) {
            convert(dst + ((size_t) y * width + x) * 4, &scanline[(size_t) x * 4]);
        }
    }
    return out;
}

}  // namespace

texture_data get_texture_data(std::string_view texture_path) {
    std::string const path{texture_path};
    std::vector<uint8_t> const file = read_file(path);
    static uint8_t const png_magic[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() >= 8 && std::memcmp(file.data(), png_magic, 8) == 0) {
        return decode_png(file, path);
    }
    if (file.size() >= 2 && file[0] == '#' && file[1] == '?') {
        return decode_hdr(file, path);
    }
    throw std::runtime_error("Can't load texture " + path + " (only PNG and Radiance HDR)");
}
