// renderer.h -- reference: class Renderer (renderer.h:7-16, 48-73).  tx, ty are accepted for source compatibility; the
// library always works on 8x8 tiles = one wave64 (which is also what the reference's main passes, kernel.cu:50-51).
// No per-pixel RNG state is allocated: the Philox streams are keyed on (pixel, sample, bounce).
#pragma once

#include <vector>

#include "camera.h"

class Renderer {
public:
    Renderer(uint32_t width, uint32_t height, uint16_t tx, uint16_t ty, uint32_t spp) : m_width(width), m_height(height), m_samples_per_pixel(spp)
    {
        (void)tx; (void)ty;
        std::clog << "- Renderer Initialized" << std::endl; // renderer.h:61
    }
    uint32_t max_depth = 10;  // camera.h:66
    uint64_t seed = 1984;     // renderer.h:45
    // Sample chunks per pixel (include/crt.h: crt_render_params.sample_chunks).  UINT32_MAX (default): one per sample, whatever the
    // sample count (the reference's own 10, kernel.cu:49, included) -- the fastest setting, and the frame is still the reference's
    // sequential sample-order sum bit for bit (renderer.h:26-32); the library keeps the scratch this needs bounded whatever the
    // frame size (slabs while they fit a budget, else per-wave rings).  0: one lane sums a pixel's samples in order (also bit-exact,
    // slower).  Any other divisor of spp: chunk sums added in chunk order.
    uint32_t sample_chunks = UINT32_MAX;
    // More than one GPU (new; the reference's launch_kernel_render drives one, renderer.h:64-73): the HIP devices that share
    // the frame, one logical rank each -- 8x8 tiles dealt round robin and gathered (CRT_MULTI_TILE, bitwise the one-GPU
    // frame) or the samples split and the float sums reduced (CRT_MULTI_SAMPLE) over RCCL.  Empty: one device.  A world
    // committed to other devices is committed again to these.
    std::vector<int> devices;
    int multi_mode = CRT_MULTI_TILE;
    crt_stats last_stats{};

    // framebuffer: width*height vec3, host memory (the reference passes managed memory, kernel.cu:60) or a HIP device pointer
    void launch_kernel_render(vec3 *framebuffer, Camera **d_camera, Hittable **d_world)
    {
        std::clog << "- Renderering" << std::endl; // renderer.h:65
        crt_render_params p{};
        p.width = m_width; p.height = m_height; p.spp = m_samples_per_pixel; p.max_depth = max_depth; p.seed = seed;
        p.tile_rank = 0; p.tile_ranks = 1;
        p.sample_chunks = sample_chunks != UINT32_MAX ? sample_chunks : (m_samples_per_pixel > 1 ? m_samples_per_pixel : 0u);
        crt_camera cam = (*d_camera)->pod;
        cam.background[0] = (*d_camera)->background.x; // Camera::background is a public member (camera.h:23)
        cam.background[1] = (*d_camera)->background.y;
        cam.background[2] = (*d_camera)->background.z;
        // ... and so is Camera::defocus_angle (camera.h:20), the value get_ray tests (camera.h:94): the constructor leaves the
        // member at its default 0.6 (its parameter shadows it), a caller that writes 0 to it gets the pinhole branch
        cam.defocus_angle = (*d_camera)->defocus_angle;
        if (devices.empty()) {
            checkCrtErrors(crt_render((*d_world)->scene, &cam, &p, &framebuffer[0].x, nullptr, &last_stats));
            return;
        }
        if ((*d_world)->devices != devices) {
            checkCrtErrors(crt_scene_commit_devices((*d_world)->scene, devices.data(), (uint32_t)devices.size(), CRT_BUILD_AUTO));
            (*d_world)->devices = devices;
        }
        checkCrtErrors(crt_render_multi((*d_world)->scene, &cam, &p, multi_mode, &framebuffer[0].x, nullptr, &last_stats));
    }

private:
    uint32_t m_width, m_height;
    uint32_t m_samples_per_pixel;
};
