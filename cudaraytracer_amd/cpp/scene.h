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
        checkCrtErrors(crt_scene_commit(world->scene, device));
        printf("- BVH constructed\n"); // bvh.h:184
        *d_world = world;
    }

private:
    std::vector<std::string> model_paths;
};
