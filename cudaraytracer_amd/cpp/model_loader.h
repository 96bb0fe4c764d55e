// model_loader.h -- reference: struct Vertex + bool LoadModel(path, vertices, indices) (model_loader.h:10-55), with the
// library's own OBJ reader instead of Assimp.
#pragma once

#include <string>
#include <vector>

#include "utility.h"

struct Vertex {
    vec3 position;
    vec3 normal;
};

inline bool LoadModel(const std::string &modelFilePath, std::vector<Vertex> &outVertices, std::vector<uint32_t> &outIndices)
{
    float *xyz_nrm = nullptr;
    uint32_t *idx = nullptr;
    uint32_t nverts = 0, ntris = 0;
    if (crt_load_obj(modelFilePath.c_str(), &xyz_nrm, &nverts, &idx, &ntris) != CRT_OK) {
        std::cerr << "Error: " << crt_last_error() << std::endl; // model_loader.h:21
        return false;
    }
    const uint32_t base = static_cast<uint32_t>(outVertices.size()); // (the reference omits this offset, model_loader.h:26-49)
    for (uint32_t v = 0; v < nverts; v++) {
        Vertex vert;
        vert.position = vec3(xyz_nrm[6 * v], xyz_nrm[6 * v + 1], xyz_nrm[6 * v + 2]);
        vert.normal = vec3(xyz_nrm[6 * v + 3], xyz_nrm[6 * v + 4], xyz_nrm[6 * v + 5]);
        outVertices.push_back(vert);
    }
    for (uint32_t k = 0; k < 3 * ntris; k++) outIndices.push_back(base + idx[k]);
    crt_free(xyz_nrm);
    crt_free(idx);
    printf("Model loaded\n");
    return true;
}
