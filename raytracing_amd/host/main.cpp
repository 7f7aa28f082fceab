// `Raytracing <scene.json>`: the reference's entry point (src/main.cpp:11-42)
// with the window loop replaced by a headless one.  The call sequence on the
// renderer table is the reference's: load_* -> initialize -> prepare_data ->
// loop { update_data; render; present; camera.dirty = false } -> destroy.
// The loop ends after a fixed number of passes instead of on window close and
// the result is written to disk instead of a swapchain.
//
//   Raytracing <scene.json | scene.rtsnap>
//       [--spp N]            passes to accumulate (default 16)
//       [--out image.png]    tonemapped image (default render.png)
//       [--raw accum.bin]    raw float accumulation dump
//       [--seed S]           seed of the per-dispatch seed stream (default 0x5EED)
//       [--random-seed]      seed it from std::random_device like the reference
//       [--reference-order]  traverse the host BVH in the shader's order
//       [--batch N]          trace N whole passes per render() call
//       [--device D]         HIP device ordinal
//       [--devices A,B,...]  several GPUs: with --batch, every render() call splits its
//                            passes over them (scene replicated per GPU) and the
//                            accumulators are summed with one RCCL reduce at the end
//       [--preview-first]    start with camera.dirty = true (exercises the preview path)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>

#include "image_io.h"
#include "renderer.h"

int main(int argc, char* argv[]) {
    std::string scene_file;
    std::string out_png = "render.png";
    std::string out_raw;
    uint32_t spp = 16;
    bool preview_first = false;
    hip_raytracer_settings settings{};
    for (int i = 1; i < argc; ++i) {
        std::string const a = argv[i];
        auto value = [&]() -> char const* {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value after %s\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--spp") {
            spp = (uint32_t) std::strtoul(value(), nullptr, 0);
        } else if (a == "--out") {
            out_png = value();
        } else if (a == "--raw") {
            out_raw = value();
        } else if (a == "--seed") {
            settings.seed = (uint32_t) std::strtoul(value(), nullptr, 0);
        } else if (a == "--random-seed") {
            settings.fixed_seed = false;
        } else if (a == "--reference-order") {
            settings.reference_order = true;
        } else if (a == "--batch") {
            settings.passes_per_render = (uint32_t) std::strtoul(value(), nullptr, 0);
        } else if (a == "--device") {
            settings.device = std::atoi(value());
        } else if (a == "--devices") {
            std::string const list = value();
            size_t at = 0;
            while (at <= list.size()) {
                size_t const comma = std::min(list.find(',', at), list.size());
                if (comma > at) {
                    settings.devices.push_back(std::atoi(list.substr(at, comma - at).c_str()));
                }
                at = comma + 1;
            }
        } else if (a == "--preview-first") {
            preview_first = true;
        } else if (!a.empty() && a[0] != '-' && scene_file.empty()) {
            scene_file = a;
        } else {
            std::fprintf(stderr, "unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (settings.devices.size() > 1 && settings.passes_per_render == 0) {
        // the tile-by-tile cadence is one dispatch at a time by definition; several GPUs need
        // whole passes to share out
        settings.passes_per_render = 16;
    }
    if (scene_file.empty()) {
        std::fprintf(stderr, "usage: Raytracing <scene.json> [--spp N] [--out image.png]\n");
        return 2;
    }
    try {
        bool const snapshot =
            scene_file.size() > 7 && scene_file.rfind(".rtsnap") == scene_file.size() - 7;
        auto [render_options, camera, scene] =
            snapshot ? load_scene_snapshot_file(scene_file) : load_scene(scene_file);
        renderer renderer{};
        configure_hip_raytracer(settings);
        load_hip_raytracer(renderer);
        renderer.initialize(render_options);
        renderer.prepare_data(scene);
        camera.dirty = preview_first;
        auto const t0 = std::chrono::steady_clock::now();
        uint64_t frames = 0;
        uint32_t const target =
            settings.passes_per_render ? 0 : spp;  // batched calls count requests instead
        uint32_t requested = 0;
        for (;;) {
            if (settings.passes_per_render) {
                if (requested >= spp && !camera.dirty) {
                    break;
                }
            } else if (hip_raytracer_passes() >= target && !camera.dirty) {
                break;
            }
            renderer.update_data(scene);
            renderer.render(camera);
            renderer.present();
            if (!camera.dirty && settings.passes_per_render) {
                requested += settings.passes_per_render;
            }
            camera.dirty = false;
            ++frames;
        }
        std::vector<float> const accum = hip_raytracer_read_accumulation();
        double const seconds =
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        uint32_t const passes = hip_raytracer_passes();
        std::printf("%u x %u, %u passes, %llu render() calls, %.3f s, %.2f Msamples/s\n",
            render_options.resolution_x, render_options.resolution_y, passes,
            (unsigned long long) frames, seconds,
            (double) render_options.resolution_x * render_options.resolution_y * passes /
                seconds / 1e6);
        write_png_rgb8(out_png,
            to_display_rgb8(accum, render_options.resolution_x, render_options.resolution_y, passes),
            render_options.resolution_x, render_options.resolution_y);
        if (!out_raw.empty()) {
            write_raw_accumulation(out_raw, accum, render_options.resolution_x,
                render_options.resolution_y, passes);
        }
        renderer.destroy();
    } catch (std::exception const& e) {
        // The reference's CHECK prints and terminates (src/check.h:10-17).
        std::fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
