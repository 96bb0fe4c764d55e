// model.h -- reference: class Model (model.h:6-32).  The reference uploads vertices/indices itself; here they stay on
// the host (the library uploads its own SoA layout at commit), so d_vertex / d_indices point into host vectors.
#pragma once

#include "model_loader.h"

class Model {
public:
    Vertex *d_vertex = nullptr;
    uint32_t *d_indices = nullptr;
    uint32_t material = 0; // reference: Material* (never used, scene.h:147-148 hard-codes the Lambertian)
    uint32_t num_triangles = 0;
    uint32_t num_vertices = 0;

    Model(const std::string &modelFilePath)
    {
        if (!LoadModel(modelFilePath, h_vertex, h_indices, &groups)) {
            std::cerr << "Failed to load OBJ model!" << std::endl; // model.h:19
            return;
        }
        num_triangles = static_cast<uint32_t>(h_indices.size() / 3);
        num_vertices = static_cast<uint32_t>(h_vertex.size());
        d_vertex = h_vertex.data();
        d_indices = h_indices.data();
    }
    std::vector<Vertex> h_vertex;
    std::vector<uint32_t> h_indices;
    std::vector<crt_obj_group> groups; // parts of the file with their .mtl records (not in the reference)
};
