// camera.h -- reference: class Camera (camera.h:11-58).  Same constructor signature and public fields; the derived
// viewport quantities live in the crt_camera POD that the kernel takes by value (crt_camera_init keeps the reference
// constructor's quirks: `aspect` ignored, parameter shadowing of vfov / vup / defocus_angle).
#pragma once

#include "hittable.h"

class Camera {
public:
    float aspect_ratio = 16.0f / 9.0f;
    float vfov = 20;
    uint32_t image_width = 1280;
    vec3 look_from = vec3(13.0f, 2.0f, 3.0f);
    vec3 look_at = vec3(0, 0, 0);
    vec3 vup = vec3(0, 1, 0);
    float defocus_angle = 0.6f;
    float focus_distance = 10.0f;
    vec3 background = vec3(0, 0, 0);

    Camera(uint32_t width, vec3 &lookfrom, vec3 &lookat, vec3 &vup_, float vfov_, float aspect, float defocus_angle_, float focus_dist)
    {
        image_width = width;
        look_from = lookfrom;
        look_at = lookat;
        focus_distance = focus_dist;
        checkCrtErrors(crt_camera_init(&pod, width, &lookfrom.x, &lookat.x, &vup_.x, vfov_, aspect, defocus_angle_, focus_dist));
    }
    crt_camera pod;
};
