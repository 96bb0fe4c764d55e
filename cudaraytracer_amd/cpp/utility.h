// utility.h -- host-side mirror of the reference's utility.h for the drop-in C++ surface (reference:
// CudaRayTracer/utility.h).  vec3 stands in for glm::vec3 (glm is not a dependency here; define CRT_USE_GLM before
// including to use glm::vec3 instead -- same 12-byte layout).  All device work happens in libcrt_hip.so (include/crt.h).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>

#include "../../include/crt.h"

#ifdef CRT_USE_GLM
#include <glm/glm.hpp>
using glm::vec3;
#else
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    vec3(float v) : x(v), y(v), z(v) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float &operator[](int i) { return (&x)[i]; }
    const float &operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator-(const vec3 &a, const vec3 &b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator+(const vec3 &a, const vec3 &b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
#endif
static_assert(sizeof(vec3) == 12, "vec3 must be three packed floats (framebuffer layout, kernel.cu:58)");

// reference: checkCudaErrors / check_cuda (utility.h:31-40): message to stderr, then exit(99).  The C ABI itself never
// exits; this macro restores the reference's process-fatal convention for code written against it.
#define checkCrtErrors(val) check_crt((val), #val, __FILE__, __LINE__)
inline void check_crt(int result, char const *const func, const char *const file, uint32_t const line)
{
    if (result) {
        std::cerr << "CRT error = " << static_cast<uint32_t>(result) << " at " << file << ":" << line << " '" << func << "' : " << crt_last_error() << "\n";
        exit(99);
    }
}

// reference: export_framebuffer_to_png (utility.h:122-145): same name, same quantisation (x*255 clamped to [0,255],
// truncated; done by the library's HIP kernel), PNG written by the library's own encoder instead of stb_image_write.
// `framebuffer` may be host memory or a HIP device pointer.
static inline void export_framebuffer_to_png(const vec3 *framebuffer, uint32_t width, uint32_t height, const char *filename, int device = 0)
{
    std::vector<uint8_t> bytes((size_t)width * height * 3);
    checkCrtErrors(crt_quantize_rgb8(&framebuffer[0].x, width, height, bytes.data(), device, nullptr));
    checkCrtErrors(crt_write_png(filename, bytes.data(), width, height));
}

// The same quantisation on the host, written as binary PPM (no library call: handy for diffing against the PNG).
// `framebuffer` is a HOST pointer here: Renderer::launch_kernel_render fills host memory.
static inline void export_framebuffer_to_ppm(const vec3 *framebuffer, uint32_t width, uint32_t height, const char *filename)
{
    FILE *f = fopen(filename, "wb");
    if (!f) { std::cerr << "cannot write " << filename << "\n"; return; }
    fprintf(f, "P6\n%u %u\n255\n", width, height);
    for (uint32_t i = 0; i < width * height; ++i) {
        const float c[3] = { framebuffer[i].x, framebuffer[i].y, framebuffer[i].z };
        unsigned char px[3];
        for (int k = 0; k < 3; k++) {
            float v = c[k] * 255.0f;
            v = v < 0.0f ? 0.0f : v;       // std::min(std::max(x * 255.0f, 0.0f), 255.0f), utility.h:134
            v = v > 255.0f ? 255.0f : v;
            px[k] = static_cast<unsigned char>(v);
        }
        fwrite(px, 1, 3, f);
    }
    fclose(f);
}
