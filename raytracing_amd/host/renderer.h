// The renderer plug-in table, as the reference declares it
// (src/renderer/renderer.h:5-21): six function pointers a back end fills in.
// load_hip_raytracer() is the MI355X back end; it takes the place of
// load_megakernel_raytracer() (src/renderer/megakernel_raytracer.cpp:815-822)
// and needs no create_render_context(): there is no window or Vulkan device.
#pragma once

#include <cstdint>
#include <vector>

#include "scene.h"

struct renderer {
    void (*initialize)(render_options const& options) = nullptr;

    void (*prepare_data)(struct scene const& scene) = nullptr;

    void (*update_data)(struct scene const& scene) = nullptr;

    void (*render)(struct camera const& camera) = nullptr;

    void (*present)() = nullptr;

    void (*destroy)() = nullptr;
};

void load_hip_raytracer(renderer& renderer);

// Additive controls of the HIP back end (the table itself is unchanged).
struct hip_raytracer_settings {
    int device = -1;              // HIP device ordinal, -1 = current
    std::vector<int> devices;     // more than one GPU: the batched render() calls split
                                  // their passes over these (scene replicated, one RCCL
                                  // sum-reduce when the image is read); overrides `device`
    bool reference_order = false; // walk the host BVH in the shader's order (exact ids)
    bool fixed_seed = true;       // mt19937(seed) instead of mt19937(random_device)
    uint32_t seed = 0x5EED;
    uint32_t passes_per_render = 0; // 0: one tile per render() like the reference;
                                    // n: render() traces n whole passes in one launch
};
void configure_hip_raytracer(hip_raytracer_settings const& settings);

// Number of complete passes accumulated (the reference's accumulation_counter).
uint32_t hip_raytracer_passes();

// Copies the accumulation image (resolution_x * resolution_y RGBA32F, sum over passes).
std::vector<float> hip_raytracer_read_accumulation();

// Copies the preview image ((resolution_x/10) * (resolution_y/10) RGBA32F).
std::vector<float> hip_raytracer_read_preview();
