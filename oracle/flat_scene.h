/* TEST INFRASTRUCTURE ONLY (oracle): flat scene file shared by the reference
 * harness and the CPU restatement.  Little-endian binary, written by
 * tests/golden/scenes.py (struct layout below, no padding surprises: every
 * record is a multiple of 8 bytes).
 *
 *   char     magic[8]  = "BDPTSCN1"
 *   uint32   n_materials, n_triangles, n_lights, reserved
 *   material[n_materials] : int32 type, int32 pad, double rgb[3], double fuzz, double ior
 *        type 0 lambertian (rgb = albedo)        src/materials/material.h:42-67
 *             1 metal      (rgb = albedo, fuzz)  src/materials/material.h:69-90
 *             2 dielectric (ior)                 src/materials/material.h:92-131
 *             3 diffuse_light (rgb = emission)   src/materials/material.h:133-147
 *   triangle[n_triangles] : double v0[3], v1[3], v2[3], int32 material, int32 pad
 *   uint32   light_triangle[n_lights]   (+ one uint32 pad if n_lights is odd)
 *        the `lights` collection handed to camera::render (src/main.cpp:67): copies of these
 *        world triangles, in this order
 *   camera : double lookfrom[3], lookat[3], vup[3], vfov, aspect_ratio, focus_dist,
 *            defocus_angle, background[3]; int32 image_width, samples_per_pixel, max_depth, pad
 *        (the public fields of src/camera.h:26-41)
 */
#ifndef ORACLE_FLAT_SCENE_H
#define ORACLE_FLAT_SCENE_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int32_t type; int32_t pad; double rgb[3]; double fuzz; double ior; } fs_material;
typedef struct { double v0[3], v1[3], v2[3]; int32_t material; int32_t pad; } fs_triangle;
typedef struct {
    double lookfrom[3], lookat[3], vup[3];
    double vfov, aspect_ratio, focus_dist, defocus_angle;
    double background[3];
    int32_t image_width, samples_per_pixel, max_depth, pad;
} fs_camera;

typedef struct {
    uint32_t n_materials, n_triangles, n_lights;
    fs_material* materials;
    fs_triangle* triangles;
    uint32_t* lights;
    fs_camera camera;
} fs_scene;

static inline int fs_load(const char* path, fs_scene* s) {
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "flat scene: cannot open %s\n", path); return -1; }
    char magic[8];
    uint32_t hdr[4];
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BDPTSCN1", 8) != 0) { fclose(f); return -2; }
    if (fread(hdr, 4, 4, f) != 4) { fclose(f); return -3; }
    s->n_materials = hdr[0]; s->n_triangles = hdr[1]; s->n_lights = hdr[2];
    s->materials = (fs_material*)malloc(sizeof(fs_material) * (s->n_materials ? s->n_materials : 1));
    s->triangles = (fs_triangle*)malloc(sizeof(fs_triangle) * (s->n_triangles ? s->n_triangles : 1));
    uint32_t nl_padded = s->n_lights + (s->n_lights & 1u);
    s->lights = (uint32_t*)malloc(sizeof(uint32_t) * (nl_padded ? nl_padded : 1));
    int ok = fread(s->materials, sizeof(fs_material), s->n_materials, f) == s->n_materials
          && fread(s->triangles, sizeof(fs_triangle), s->n_triangles, f) == s->n_triangles
          && fread(s->lights, 4, nl_padded, f) == nl_padded
          && fread(&s->camera, sizeof(fs_camera), 1, f) == 1;
    fclose(f);
    return ok ? 0 : -4;
}

static inline void fs_free(fs_scene* s) { free(s->materials); free(s->triangles); free(s->lights); }

#endif
