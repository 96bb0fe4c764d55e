// hittable.h -- the reference hands the world around as an opaque `Hittable**` (kernel.cu:70-74, renderer.h:10); main()
// never looks inside.  Here Hittable is the owner of the library's scene handle.
#pragma once

#include <vector>

#include "utility.h"

class Hittable {
public:
    Hittable() : scene(crt_scene_create()) {}
    ~Hittable() { crt_scene_destroy(scene); } // reference: free_world<<<1,1>>> deletes *d_world (kernel.cu:25-28)
    Hittable(const Hittable &) = delete;
    Hittable &operator=(const Hittable &) = delete;
    crt_scene *scene;
    std::vector<int> devices; // HIP devices the scene was committed to with crt_scene_commit_devices (empty: one device, crt_scene_commit)
};
