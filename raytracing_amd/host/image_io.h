// Headless image output: the display transform of the reference's present
// pass and two file writers.  The reference itself can only show images in a
// window (its stb_image_write stub is unused), so this is additive.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

// accumulation / passes -> Reinhard -> gamma 2.2 -> 8 bit
// (shaders/raytracer/rect.frag:18-23, shaders/common/to_ldr.glsl:1-7,
// frame_scalar = 1 / accumulation_counter: src/renderer/megakernel_raytracer.cpp:748-753).
std::vector<uint8_t> to_display_rgb8(
    std::vector<float> const& rgba, uint32_t width, uint32_t height, uint32_t passes);

// 8-bit RGB PNG (zlib deflate, filter 0).
void write_png_rgb8(std::string const& path, std::vector<uint8_t> const& rgb, uint32_t width,
    uint32_t height);

// Raw dump: "RTACCUM1", u32 width, u32 height, u32 passes, then width*height*4 float32.
void write_raw_accumulation(std::string const& path, std::vector<float> const& rgba,
    uint32_t width, uint32_t height, uint32_t passes);
