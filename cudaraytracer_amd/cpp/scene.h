// scene.h -- reference: class Scene (scene.h:20-51, 189-308).  launch_triangles_scene() builds the reference scene --
// every mesh triangle Lambertian(0.7, 0.3, 0.16), ground Sphere((0,-1000,-1), 1000, Lambertian(0.5)), light
// Sphere((3,6.5,3), 2, DiffuseLight(15,15,15)) (scene.h:147-148,164-165) -- through the C ABI and commits it to the GPU.
#pragma once

#include <string>
#include <vector>

#include "hittable.h"
#include "model.h"

class Scene {
public:
    Scene() : model_paths{ "../models/teapot.obj" } {} // the reference's literal (scene.h:241); not shipped with it either
    // reference: declared at scene.h:25, never defined; here it selects the OBJ files launch_triangles_scene() loads
    void load_models(const std::vector<std::string> &modelPaths) { model_paths = modelPaths; }
    int device = 0;
    // More than one GPU (new; the reference is single-GPU): the HIP devices the scene is replicated to, one per logical
    // rank of Renderer's tile / sample partition (crt_scene_commit_devices).  Empty: `device` alone.
    std::vector<int> devices;
    // false (reference, scene.h:147-148): every triangle Lambertian(0.7, 0.3, 0.16).  true (row N4): each part of a model
    // gets its .mtl material -- DiffuseLight(Ke) when it emits, else Lambertian(Kd).
    bool materials_from_mtl = false;

    void launch_triangles_scene(Hittable **d_world)
    {
        Hittable *world = new Hittable();
        const float tri_rgb[3] = { 0.7f, 0.3f, 0.16f }, ground_rgb[3] = { 0.5f, 0.5f, 0.5f }, light_rgb[3] = { 15.0f, 15.0f, 15.0f };
        uint32_t tri_mat = 0, ground_mat = 0, light_mat = 0;
        checkCrtErrors(crt_scene_add_material(world->scene, CRT_MAT_LAMBERTIAN, tri_rgb, 0.0f, UINT32_MAX, &tri_mat));
        uint32_t num_triangles = 0;
        for (const std::string &path : model_paths) {
            Model model(path);
            if (model.num_triangles == 0) continue;
            if (materials_from_mtl) {
                for (const crt_obj_group &g : model.groups) {
                    const bool emits = g.Ke[0] > 0.0f || g.Ke[1] > 0.0f || g.Ke[2] > 0.0f;
                    uint32_t m = 0;
                    checkCrtErrors(crt_scene_add_material(world->scene, emits ? CRT_MAT_DIFFUSE_LIGHT : CRT_MAT_LAMBERTIAN, emits ? g.Ke : g.Kd, 0.0f, UINT32_MAX, &m));
                    checkCrtErrors(crt_scene_add_mesh(world->scene, &model.d_vertex[0].position.x, sizeof(Vertex) / sizeof(float), model.num_vertices,
                                                      model.d_indices + 3 * (size_t)g.first_triangle, g.num_triangles, m));
                }
            } else
                checkCrtErrors(crt_scene_add_mesh(world->scene, &model.d_vertex[0].position.x, sizeof(Vertex) / sizeof(float), model.num_vertices,
                                                  model.d_indices, model.num_triangles, tri_mat));
            num_triangles += model.num_triangles;
        }
        checkCrtErrors(crt_scene_add_material(world->scene, CRT_MAT_LAMBERTIAN, ground_rgb, 0.0f, UINT32_MAX, &ground_mat));
        checkCrtErrors(crt_scene_add_material(world->scene, CRT_MAT_DIFFUSE_LIGHT, light_rgb, 0.0f, UINT32_MAX, &light_mat));
        const float ground_c[3] = { 0.0f, -1000.0f, -1.0f }, light_c[3] = { 3.0f, 6.5f, 3.0f };
        checkCrtErrors(crt_scene_add_sphere(world->scene, ground_c, 1000.0f, ground_mat));
        checkCrtErrors(crt_scene_add_sphere(world->scene, light_c, 2.0f, light_mat));
        printf("- Scene Created with %d objects\n", num_triangles + 2); // scene.h:175
        commit(world);
        printf("- BVH constructed\n"); // bvh.h:184
        *d_world = world;
    }

    // reference: launch_random_scene_kernel + kernel_create_world (scene.h:53-90, 197-235): ground, 22 x 22 small spheres
    // (80 % Lambertian(RND*RND), 15 % Metal(0.5(1+RND), 0.5 RND), 5 % Dielectric(1.5)), three r = 1 spheres.  RND is drawn
    // in the reference's order from a small generator seeded like rand_init (scene.h:183: seed 1984); the reference's
    // numbers come from curand XORWOW, so the layout is statistically the same scene, not the same spheres.
    uint64_t random_seed = 1984;
    void launch_random_scene_kernel(Hittable **d_world)
    {
        Hittable *world = new Hittable();
        uint64_t state = random_seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
        auto RND = [&state]() { // splitmix64 -> (0, 1] on a 2^-24 grid, the range of curand_uniform
            state += 0x9E3779B97F4A7C15ull;
            uint64_t z = state;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
            z ^= z >> 31;
            return (float)((uint32_t)(z >> 40) + 1u) * 5.9604644775390625e-08f;
        };
        uint32_t n = 0;
        auto add = [&](float cx, float cy, float cz, float r, int type, float cr, float cg, float cb, float param) {
            const float rgb[3] = { cr, cg, cb }, c[3] = { cx, cy, cz };
            uint32_t m = 0;
            checkCrtErrors(crt_scene_add_material(world->scene, type, rgb, param, UINT32_MAX, &m));
            checkCrtErrors(crt_scene_add_sphere(world->scene, c, r, m));
            n++;
        };
        add(0.0f, -1000.0f, -1.0f, 1000.0f, CRT_MAT_LAMBERTIAN, 0.5f, 0.5f, 0.5f, 0.0f);
        for (int a = -11; a < 11; a++)
            for (int b = -11; b < 11; b++) {
                const float choose_mat = RND();
                const float cx = (float)a + RND(), cz = (float)b + RND();
                if (choose_mat < 0.8f) {
                    const float r = RND() * RND(), g = RND() * RND(), bl = RND() * RND();
                    add(cx, 0.2f, cz, 0.2f, CRT_MAT_LAMBERTIAN, r, g, bl, 0.0f);
                } else if (choose_mat < 0.95f) {
                    const float r = 0.5f * (1.0f + RND()), g = 0.5f * (1.0f + RND()), bl = 0.5f * (1.0f + RND());
                    add(cx, 0.2f, cz, 0.2f, CRT_MAT_METAL, r, g, bl, 0.5f * RND());
                } else {
                    add(cx, 0.2f, cz, 0.2f, CRT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f);
                }
            }
        add(0.0f, 1.0f, 0.0f, 1.0f, CRT_MAT_DIELECTRIC, 1.0f, 1.0f, 1.0f, 1.5f);
        add(-4.0f, 1.0f, 0.0f, 1.0f, CRT_MAT_LAMBERTIAN, 0.4f, 0.2f, 0.1f, 0.0f);
        add(4.0f, 1.0f, 0.0f, 1.0f, CRT_MAT_METAL, 0.7f, 0.6f, 0.5f, 0.0f);
        printf("- Scene Created with %u objects\n", n);
        commit(world);
        printf("- BVH constructed\n"); // bvh.h:184
        *d_world = world;
    }

private:
    void commit(Hittable *world)
    {
        if (devices.empty()) checkCrtErrors(crt_scene_commit(world->scene, device));
        else {
            checkCrtErrors(crt_scene_commit_devices(world->scene, devices.data(), (uint32_t)devices.size(), CRT_BUILD_AUTO));
            world->devices = devices;
        }
    }
    std::vector<std::string> model_paths;
};
