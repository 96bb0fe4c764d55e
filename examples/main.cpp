// examples/main.cpp -- the reference's driver flow (kernel.cu:44-103: camera -> scene -> renderer -> render -> export ->
// free) written against the drop-in headers.  Own code; build:
//   g++ -O2 -std=c++17 examples/main.cpp -Lcudaraytracer_amd -lcrt_hip -Wl,-rpath,$PWD/cudaraytracer_amd -o crt_demo
//   ./crt_demo [model.obj] [width] [height] [spp] [--gpus N | --gpus d0,d1,...] [--mode tile|sample] [--mtl]
// --gpus N shares the frame among HIP devices 0..N-1; a device list may name a device more than once (logical ranks on
// one GPU).
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>

#include "../cudaraytracer_amd/cpp/renderer.h"
#include "../cudaraytracer_amd/cpp/scene.h"

int main(int argc, char **argv)
{
    std::vector<int> devices;
    int multi_mode = CRT_MULTI_TILE;
    uint32_t chunks = UINT32_MAX; // Renderer's default
    bool from_mtl = false; // --mtl: every part of the model gets its .mtl material (Scene::materials_from_mtl, row N4) instead of the reference's one Lambertian
    { // options are taken out of argv; the positional arguments stay where the reference-style code below expects them
        int kept = 1;
        for (int k = 1; k < argc; k++) {
            if (!strcmp(argv[k], "--gpus") && k + 1 < argc) {
                const char *v = argv[++k];
                if (strchr(v, ',')) { for (const char *q = v; *q;) { devices.push_back(atoi(q)); q = strchr(q, ','); if (!q) break; q++; } }
                else for (int d = 0; d < atoi(v); d++) devices.push_back(d);
            } else if (!strcmp(argv[k], "--mode") && k + 1 < argc) {
                multi_mode = !strcmp(argv[++k], "sample") ? CRT_MULTI_SAMPLE : CRT_MULTI_TILE;
            } else if (!strcmp(argv[k], "--chunks") && k + 1 < argc) { // sample chunks per pixel (default: one per sample); 0 = whole pixels
                chunks = (uint32_t)atoi(argv[++k]);
            } else if (!strcmp(argv[k], "--mtl")) {
                from_mtl = true;
            } else argv[kept++] = argv[k];
        }
        argc = kept;
    }
    const float ASPECT_RATIO = 16.0f / 9.0f;
    const std::string model = argc > 1 ? argv[1] : "models/cat.obj";
    const uint32_t IMAGE_WIDTH = argc > 2 ? (uint32_t)atoi(argv[2]) : 1000, IMAGE_HEIGHT = argc > 3 ? (uint32_t)atoi(argv[3]) : 1000;
    const uint32_t spp = argc > 4 ? (uint32_t)atoi(argv[4]) : 10, tx = 8, ty = 8;
    std::cerr << "Rendering a " << IMAGE_WIDTH << "x" << IMAGE_HEIGHT << " image with " << spp << " samples per pixel in " << tx << "x" << ty << " blocks.\n";

    std::vector<vec3> framebuffer((size_t)IMAGE_WIDTH * IMAGE_HEIGHT);

    // "spheres" instead of a model path: the reference's sphere scene (launch_random_scene_kernel, scene.h:197-235) seen
    // from the Camera class's own defaults (camera.h:13-15) under a sky-coloured background
    const bool spheres = model == "spheres";
    // init_camera (kernel.cu:30-42)
    vec3 look_from(8.f, 5.f, -6.0f), look_at(-4, 0, 4), vup(0, 1, 0);
    if (spheres) { look_from = vec3(13.0f, 2.0f, 3.0f); look_at = vec3(0, 0, 0); }
    const float focus_distance = std::sqrt(look_from.x * look_from.x + look_from.y * look_from.y + look_from.z * look_from.z);
    Camera *camera = new Camera(IMAGE_WIDTH, look_from, look_at, vup, spheres ? 20.0f : 25.0f, ASPECT_RATIO, 0.6f, focus_distance);
    if (spheres) camera->background = vec3(0.7f, 0.8f, 1.0f);
    Camera **d_camera = &camera;

    Hittable *world = nullptr;
    Hittable **d_world = &world;
    Scene scene;
    scene.devices = devices;
    scene.materials_from_mtl = from_mtl;
    if (spheres) scene.launch_random_scene_kernel(d_world);
    else {
        scene.load_models({ model });
        scene.launch_triangles_scene(d_world);
    }
    std::clog << "- Scene Created\n";

    Renderer renderer(IMAGE_WIDTH, IMAGE_HEIGHT, tx, ty, spp);
    renderer.devices = devices; // empty: the reference's single-GPU path
    renderer.multi_mode = multi_mode;
    renderer.sample_chunks = chunks;
    const auto start = std::chrono::steady_clock::now();
    renderer.launch_kernel_render(framebuffer.data(), d_camera, d_world);
    const double timer_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count();
    std::cerr << "\nRender time: " << timer_seconds << " seconds (kernel " << renderer.last_stats.kernel_ms << " ms, " << renderer.last_stats.rays
              << " rays).\n";

    export_framebuffer_to_png(framebuffer.data(), IMAGE_WIDTH, IMAGE_HEIGHT, "output.png"); // kernel.cu:91
    export_framebuffer_to_ppm(framebuffer.data(), IMAGE_WIDTH, IMAGE_HEIGHT, "output.ppm"); // same bytes, host quantisation
    delete world;   // free_world<<<1,1>>> (kernel.cu:25-28,93)
    delete camera;
    return 0;
}
