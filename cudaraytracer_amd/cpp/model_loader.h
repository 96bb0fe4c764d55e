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

// Parts of a model file (an `o`/`g` object under one `usemtl`) with their .mtl record: crt_obj_group (include/crt.h).
// The reference's loader drops this information; Scene::materials_from_mtl uses it.
inline bool LoadModel(const std::string &modelFilePath, std::vector<Vertex> &outVertices, std::vector<uint32_t> &outIndices,
                      std::vector<crt_obj_group> *outGroups = nullptr)
{
    float *xyz_nrm = nullptr;
    uint32_t *idx = nullptr, *tri_group = nullptr;
    crt_obj_group *groups = nullptr;
    uint32_t nverts = 0, ntris = 0, ngroups = 0;
    const uint32_t tri_base = static_cast<uint32_t>(outIndices.size() / 3);
    if (crt_load_obj_ex(modelFilePath.c_str(), &xyz_nrm, &nverts, &idx, &ntris, outGroups ? &tri_group : nullptr, outGroups ? &groups : nullptr,
                        outGroups ? &ngroups : nullptr) != CRT_OK) {
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
    for (uint32_t g = 0; g < ngroups; g++) {
        crt_obj_group grp = groups[g];
        grp.first_triangle += tri_base;
        outGroups->push_back(grp);
    }
    crt_free(xyz_nrm);
    crt_free(idx);
    crt_free(tri_group);
    crt_free(groups);
    printf("Model loaded\n");
    return true;
}
