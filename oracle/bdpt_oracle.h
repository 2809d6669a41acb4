/* TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement (oracle/bdpt_oracle.cpp) of the reference's
 * per-pixel radiance-estimation path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and only as the checker /
 * the timed CPU baseline - never as a fallback of the product.
 *
 * All arrays are plain little-endian doubles / ints owned by the caller.
 */
#ifndef BDPT_ORACLE_H
#define BDPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_scene oracle_scene;

/* material types */
enum { ORACLE_LAMBERTIAN = 0, ORACLE_METAL = 1, ORACLE_DIELECTRIC = 2, ORACLE_DIFFUSE_LIGHT = 3, ORACLE_ISOTROPIC = 4 };

/* tri_vertices: n_tri * 9 doubles (v0 v1 v2); mat_params: n_mat * 5 doubles (r g b fuzz ior);
 * lights: indices of the world triangles that make up the `lights` collection
 * (src/main.cpp:67; pass n_lights = 0 to get the "whole world" fallback). */
oracle_scene* oracle_scene_create(const double* tri_vertices, const int32_t* tri_material, uint32_t n_tri,
                                  const int32_t* mat_type, const double* mat_params, uint32_t n_mat,
                                  const uint32_t* lights, uint32_t n_lights);
/* A composite scene - translate / rotate_y instances over meshes of their own, checker / noise / image textures,
 * constant_medium with its isotropic phase function (src/objects/hittable.h:46-157, src/materials/textures/texture.h,
 * perlin.h, src/materials/volumes/constant_medium.h, src/materials/material.h:149-172); arguments in bdpt_oracle.cpp. */
oracle_scene* oracle_scene_create_composite(const double* tri_vertices, const int32_t* tri_material, uint32_t n_tri,
                                            const int32_t* mat_type, const double* mat_params, const int32_t* mat_tex, uint32_t n_mat,
                                            const int32_t* tex_ints, const double* tex_doubles, uint32_t n_tex,
                                            const uint8_t* image_bytes, uint64_t n_image_bytes, const int32_t* obj_ints,
                                            const double* obj_doubles, uint32_t n_obj, const uint32_t* lights, uint32_t n_lights);
uint32_t oracle_world_leaves(const oracle_scene* s, int32_t* out, uint32_t capacity);
uint32_t oracle_world_boxes(const oracle_scene* s, double* out, uint32_t capacity_nodes);
uint32_t oracle_perlin_tables(const oracle_scene* s, double* out);
oracle_scene* oracle_scene_load(const char* flat_scene_path, double* camera_public_out /* 17 doubles */,
                                int32_t* camera_ints_out /* image_width, samples_per_pixel, max_depth */);
void oracle_scene_destroy(oracle_scene* s);

uint32_t oracle_scene_triangle_count(const oracle_scene* s);
uint32_t oracle_bvh_node_count(const oracle_scene* s);
/* leaf triangle ids in the reference's left-then-right visiting order (a span-1 node lists its triangle twice) */
uint32_t oracle_bvh_dfs_leaves(const oracle_scene* s, uint32_t* out, uint32_t capacity);
/* node boxes in pre-order: xmin xmax ymin ymax zmin zmax per node */
void oracle_bvh_boxes(const oracle_scene* s, double* out);
/* per-triangle precomputed data: edge1[3] edge2[3] normal[3] area bbox[6]  (16 doubles each) */
void oracle_triangle_data(const oracle_scene* s, double* out);

/* camera::initialize (src/camera.h:160-197).
 * pub: lookfrom[3] lookat[3] vup[3] vfov aspect_ratio focus_dist   (12 doubles)
 * out: center[3] pixel00_loc[3] pixel_delta_u[3] pixel_delta_v[3]  (12 doubles) */
void oracle_camera_initialize(const double* pub, int32_t image_width, int32_t samples_per_pixel,
                              double* out, int32_t* image_height, int32_t* sqrt_spp);

/* closest hit of bvh_node::hit (src/acceleration/bvh.h:50-59).
 * rays: n * 8 doubles (origin[3] dir[3] tmin tmax); out: n * 11 doubles
 * (triangle id or -1, t, u, v, front_face, p[3], normal[3]); stats: 7 uint64 (BvhStats order) or NULL */
void oracle_trace_closest(const oracle_scene* s, const double* rays, uint64_t n, double* out, uint64_t* stats);

/* camera::visible (src/camera.h:425-438); pairs: n * 6 doubles (a[3] b[3]); out: n doubles (0/1) */
void oracle_visible(const oracle_scene* s, const double* pairs, uint64_t n, double* out);

/* The worker loop of camera::render (src/camera.h:116-124) over the pixel rectangle [x0,x1) x [y0,y1)
 * of a W-wide image, each (pixel, sample) on its own counter-based stream (oracle_rng.h).
 * camera: center pixel00_loc pixel_delta_u pixel_delta_v background defocus_disk_u defocus_disk_v defocus_angle (22 doubles).
 * fb_out: (y1-y0)*(x1-x0)*3 doubles, the per-pixel SUM over samples.
 * per_sample_out (optional): 4 doubles per sample (r g b draws), pixel-major then s_j, s_i.
 * stats (optional): 7 BvhStats counters, then total RNG draws, then shadow rays, then connection pairs evaluated. */
void oracle_render(const oracle_scene* s, const double* camera, int32_t W, int32_t sqrt_spp, int32_t max_depth,
                   uint64_t seed, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                   double* fb_out, double* per_sample_out, uint64_t* stats);

/* The same loop with camera::path_trace_color (src/camera.h:255-292) as the estimator: what ray_color selects when
 * `lights` is not a triangle_collection (camera.h:245-253).  stats as above (no shadow rays / pairs). */
void oracle_render_unidirectional(const oracle_scene* s, const double* camera, int32_t W, int32_t sqrt_spp,
                                  int32_t max_depth, uint64_t seed, int32_t x0, int32_t y0, int32_t x1, int32_t y1,
                                  double* fb_out, uint64_t* stats);

/* Same loop over the whole W x H image on ONE serial default-seeded mt19937
 * (the unmodified reference's generator, src/main.h:28-32) - pins the restatement
 * against the unshimmed reference. */
void oracle_render_serial_mt(const oracle_scene* s, const double* camera, int32_t W, int32_t H, int32_t sqrt_spp,
                             int32_t max_depth, double* fb_out, uint64_t* stats);

/* colors_to_rgb8 (src/image/wpng.h:14-35) */
void oracle_colors_to_rgb8(const double* fb, int32_t W, int32_t H, int32_t samples_per_pixel, uint8_t* out);

/* the deterministic math shared with the HIP kernels */
void oracle_sincos_array(const double* x, uint64_t n, double* s, double* c);
void oracle_pow5_array(const double* x, uint64_t n, double* out);
void oracle_log_array(const double* x, uint64_t n, double* out);
/* first n doubles of stream (seed, pixel, sample) */
void oracle_rng_array(uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n, double* out);
void oracle_mt_array(uint64_t n, double* out);

#ifdef __cplusplus
}
#endif
#endif
