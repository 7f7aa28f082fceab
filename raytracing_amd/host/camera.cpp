// Camera set-up (src/asset/camera.cpp:9-36, 66-88 of the reference).
// Only the parts that feed the compute pass are kept; window input handling
// belongs to the replaced Vulkan/GLFW layer.
#include <cmath>

#include "scene.h"

namespace {

// Rebuilds front/right/up from yaw and pitch; world up is +y unless the view
// direction is (almost) parallel to it.
void refresh_basis(camera& cam) {
    float const cos_yaw = std::cos(cam.yaw);
    float const sin_yaw = std::sin(cam.yaw);
    float const cos_pitch = std::cos(cam.pitch);
    float const sin_pitch = std::sin(cam.pitch);
    cam.front = rt::normalize(rt::vec3{cos_pitch * cos_yaw, sin_pitch, cos_pitch * sin_yaw});
    rt::vec3 const world_up =
        cam.front.y < 0.99f ? rt::vec3{0.0f, 1.0f, 0.0f} : rt::vec3{-1.0f, 0.0f, 0.0f};
    cam.right = rt::normalize(rt::cross(cam.front, world_up));
    cam.up = rt::normalize(rt::cross(cam.right, cam.front));
}

}  // namespace

camera create_camera(rt::vec3 const& lookfrom, rt::vec3 const& lookat, float fov) {
    camera cam{};
    cam.position = lookfrom;
    cam.fov = fov;
    cam.front = rt::normalize(lookat - lookfrom);
    cam.pitch = std::asin(cam.front.y);
    cam.yaw = std::atan2(cam.front.z, cam.front.x);
    refresh_basis(cam);
    return cam;
}

rt_camera get_glsl_raytracer_camera(
    camera const& cam, uint32_t frame_width, uint32_t frame_height) {
    float const focal_length = 1.0f;
    float const fov_radians = cam.fov * 0.01745329251994329576923690768489f;
    float const viewport_height = 2 * std::tan(fov_radians / 2) * focal_length;
    float const viewport_width =
        viewport_height * ((float) frame_width / (float) frame_height);
    rt::vec3 const viewport_u = viewport_width * cam.right;
    rt::vec3 const viewport_v = viewport_height * (-cam.up);
    rt::vec3 const delta_u = viewport_u / (float) frame_width;
    rt::vec3 const delta_v = viewport_v / (float) frame_height;
    rt::vec3 const upper_left =
        cam.position + focal_length * cam.front - 0.5f * (viewport_u + viewport_v);
    rt::vec3 const first_pixel = upper_left + 0.5f * (delta_u + delta_v);
    rt_camera out{};
    out.pixel_delta_u[0] = delta_u.x;
    out.pixel_delta_u[1] = delta_u.y;
    out.pixel_delta_u[2] = delta_u.z;
    out.pixel_delta_v[0] = delta_v.x;
    out.pixel_delta_v[1] = delta_v.y;
    out.pixel_delta_v[2] = delta_v.z;
    out.upper_left_pixel[0] = first_pixel.x;
    out.upper_left_pixel[1] = first_pixel.y;
    out.upper_left_pixel[2] = first_pixel.z;
    out.position[0] = cam.position.x;
    out.position[1] = cam.position.y;
    out.position[2] = cam.position.z;
    return out;
}
