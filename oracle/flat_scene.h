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
 *
 * Optional trailing block for composite scenes (instances, textures, media):
 *   char     magic[8]  = "BDPTEXT1"
 *   uint32   n_textures, n_objects, reserved, reserved
 *   texture[n_textures] : int32 type, even, odd, width, height, pad; double rgb[3]; double scale
 *        type 0 solid_color (rgb), 1 checker_texture (scale, even / odd = indices of earlier textures),
 *             2 noise_texture (scale), 3 image_texture (the file "<scene file>.tex<index>.png")
 *        src/materials/textures/texture.h:14-87
 *   object[n_objects] : uint32 first_triangle, n_triangles, flags; int32 phase_material; double rotate_y_degrees;
 *                       double offset[3]; double density
 *        flags 1 rotate_y, 2 translate (outside the rotation), 4 constant_medium around the result
 *        src/objects/hittable.h:46-157, src/materials/volumes/constant_medium.h
 *   With the block present, a material's `pad` is a texture index + 1 (0 = its solid colour), material type 4 is
 *   isotropic (src/materials/material.h:149-172), the triangles of an object are given in the object's own
 *   coordinates, and the world's hittable_list is: triangles of no object and objects, in triangle-array order.
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

typedef struct { int32_t type, even, odd, width, height, pad; double rgb[3]; double scale; } fs_texture;
typedef struct { uint32_t first_triangle, n_triangles, flags; int32_t phase_material; double rotate_y_degrees; double offset[3]; double density; } fs_object;

typedef struct {
    uint32_t n_materials, n_triangles, n_lights;
    fs_material* materials;
    fs_triangle* triangles;
    uint32_t* lights;
    fs_camera camera;
    int composite;                 /* 1 when the trailing block is present */
    uint32_t n_textures, n_objects;
    fs_texture* textures;
    fs_object* objects;
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
    s->composite = 0; s->n_textures = 0; s->n_objects = 0; s->textures = NULL; s->objects = NULL;
    if (ok && fread(magic, 1, 8, f) == 8) {
        uint32_t eh[4];
        if (memcmp(magic, "BDPTEXT1", 8) != 0 || fread(eh, 4, 4, f) != 4) { fclose(f); return -5; }
        s->composite = 1; s->n_textures = eh[0]; s->n_objects = eh[1];
        s->textures = (fs_texture*)malloc(sizeof(fs_texture) * (eh[0] ? eh[0] : 1));
        s->objects = (fs_object*)malloc(sizeof(fs_object) * (eh[1] ? eh[1] : 1));
        ok = fread(s->textures, sizeof(fs_texture), eh[0], f) == eh[0] && fread(s->objects, sizeof(fs_object), eh[1], f) == eh[1];
    }
    fclose(f);
    return ok ? 0 : -4;
}

static inline void fs_free(fs_scene* s) { free(s->materials); free(s->triangles); free(s->lights); free(s->textures); free(s->objects); }

#endif
