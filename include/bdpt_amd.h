/* bdpt_amd.h - C ABI of the MI355X (gfx950) radiance-estimation library.
 *
 * Drop-in boundary for the per-pixel hot path of teehee567/Bidirectional-Path-Tracer:
 * everything between "the scene is a triangle soup + material table + camera"
 * (what load_scene_from_yaml returns, src/scene/scene_loader.h:22-28,480-523) and
 * "a framebuffer of per-pixel radiance SUMS" (src/camera.h:55,124).  The reference has
 * no FFI of its own; each entry point names the C++ call of the reference it replaces.
 *
 * Conventions: plain C, no exceptions cross the boundary.  Every function returns
 * BDPT_OK (0) or a negative error code; bdpt_last_error() returns the message of the
 * calling thread's last failure.  Handles are opaque.  The caller owns every host
 * buffer it passes; the library owns all device memory except buffers the caller
 * passes to the *_device entry points.  A context belongs to one GPU and is not
 * thread-safe; different contexts may be used from different threads.  All
 * floating-point data is IEEE binary64, as in the reference (src/core/vec3.h:9).
 */
#ifndef BDPT_AMD_H
#define BDPT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDPT_OK 0
#define BDPT_ERR_INVALID_ARGUMENT (-1)
#define BDPT_ERR_DEVICE (-2)          /* a HIP call failed / no usable GPU */
#define BDPT_ERR_OUT_OF_MEMORY (-3)
#define BDPT_ERR_UNSUPPORTED (-4)

typedef struct bdpt_context bdpt_context;
typedef struct bdpt_scene bdpt_scene;

/* Material kinds (src/materials/material.h:42-172). */
enum {
    BDPT_LAMBERTIAN = 0,     /* rgb = albedo                      material.h:42-67   */
    BDPT_METAL = 1,          /* rgb = albedo, fuzz (clamped to 1) material.h:69-90   */
    BDPT_DIELECTRIC = 2,     /* ior                               material.h:92-131  */
    BDPT_DIFFUSE_LIGHT = 3,  /* rgb = emitted radiance            material.h:133-147 */
    BDPT_ISOTROPIC = 4       /* rgb = albedo; the phase function of a constant_medium, material.h:149-172
                                (bdpt_scene_create_composite only) */
};

typedef struct bdpt_material {
    int32_t type;
    int32_t reserved;        /* bdpt_scene_create_composite: 0 = the solid colour `rgb`, k > 0 = texture k - 1
                                (lambertian, diffuse_light and isotropic take a texture, material.h:45,135,152) */
    double rgb[3];
    double fuzz;
    double ior;
} bdpt_material;

/* Public camera fields, src/camera.h:26-38.  defocus_angle > 0 turns the defocus disk on (camera.h:208,
 * 230-234); the reference's scene loader always sets it to 0 (src/scene/scene_loader.h:463). */
typedef struct bdpt_camera_public {
    double lookfrom[3];
    double lookat[3];
    double vup[3];
    double vfov;
    double aspect_ratio;
    double focus_dist;
    double defocus_angle;
    double background[3];
    int32_t image_width;
    int32_t samples_per_pixel;
    int32_t max_depth;
    int32_t reserved;
} bdpt_camera_public;

/* What camera::initialize derives (src/camera.h:148-155,160-197). */
typedef struct bdpt_camera_state {
    double center[3];
    double pixel00_loc[3];
    double pixel_delta_u[3];
    double pixel_delta_v[3];
    double background[3];
    int32_t image_width;
    int32_t image_height;
    int32_t sqrt_spp;        /* int(sqrt(samples_per_pixel)); sqrt_spp^2 samples are taken (camera.h:164) */
    int32_t max_depth;
    double defocus_disk_u[3];   /* camera.h:193-196 */
    double defocus_disk_v[3];
    double defocus_angle;       /* <= 0: rays start at `center` (camera.h:208) */
} bdpt_camera_state;

/* One render call = the worker loop of camera::render (src/camera.h:111-129) over the
 * pixels this call owns.  Pixels are owned when they lie in the rectangle [x0,x1) x [y0,y1)
 * AND their row j satisfies (j / shard_rows) % shard_count == shard_index (row bands dealt
 * round-robin to the GPUs of a node; shard_count = 1 owns everything).  Pixels not owned
 * are left untouched, so summing the framebuffers of all shards of a zero-initialised
 * image gives the unsharded image exactly.  Random numbers: sample s of pixel p always
 * uses stream (seed, p = j*image_width + i, s = s_j*sqrt_spp + s_i), whatever the sharding. */
typedef struct bdpt_render_params {
    uint64_t seed;
    int32_t x0, y0, x1, y1;     /* x1 <= 0 or y1 <= 0 means "up to the image edge" */
    int32_t shard_index;
    int32_t shard_count;        /* 0 is treated as 1 */
    int32_t shard_rows;         /* 0 is treated as 1 */
    int32_t flags;              /* BDPT_RENDER_* */
} bdpt_render_params;

#define BDPT_RENDER_COUNT_STATS 1   /* accumulate bdpt_stats with the reference's BvhStats semantics: visibility queries
                                       run the reference's full closest-hit traversal (camera.h:437) so every counter
                                       equals what the CPU renderer prints (slower; for verification) */
#define BDPT_RENDER_COUNT_WORK 2    /* accumulate bdpt_stats of the work the production path does (visibility queries
                                       stop at the first hit); ignored when BDPT_RENDER_COUNT_STATS is set */
#define BDPT_RENDER_MEGAKERNEL 8    /* render with the single persistent kernel (one thread per sample) instead of the
                                       wavefront pipeline; same framebuffer bit for bit (BDPT_RENDER_COUNT_STATS implies it) */
#define BDPT_RENDER_TIME_KERNEL 4   /* bracket the render kernels of the call with HIP events on the caller's stream
                                       (bdpt_render_get_kernel_ms) */
#define BDPT_RENDER_UNIDIRECTIONAL 32 /* estimate with camera::path_trace_color (src/camera.h:255-292: mixture of light and
                                       BSDF sampling) instead of bidirectional_color - what the reference's ray_color
                                       does when `lights` is not a triangle_collection (camera.h:245-253).  Runs on the
                                       wavefront pipeline like the default estimator (on the single kernel with
                                       BDPT_RENDER_MEGAKERNEL / BDPT_RENDER_COUNT_STATS) */
#define BDPT_RENDER_TIME_STAGES 16  /* additionally record a HIP event pair around every ray-traversal launch, on the
                                       internal stream it is launched on (bdpt_render_get_stage_ms); wavefront pipeline only */

/* Work counters.  The first six have the meaning of the reference's BvhStats
 * (src/core/stats.h:8-16); a triangle referenced twice by a one-triangle BVH node is
 * tested once here and counted twice, as the reference does. */
typedef struct bdpt_stats {
    uint64_t rays_traced;       /* closest-hit queries of trace_path (camera.h:334) */
    uint64_t aabb_tests;
    uint64_t aabb_hits;
    uint64_t bvh_node_visits;   /* with BDPT_RENDER_COUNT_WORK: node records fetched, in units of 64 bytes */
    uint64_t triangle_tests;
    uint64_t triangle_hits;
    uint64_t shadow_rays;       /* visibility queries that reached the BVH (camera.h:437) */
    uint64_t connection_pairs;  /* non-delta (eye, light) vertex pairs (camera.h:441) */
    uint64_t rng_draws;
    uint64_t samples;           /* (pixel, sample) pairs rendered */
    uint64_t path_vertices;     /* non-delta path vertices kept for the connection step (camera.h:342-349) */
} bdpt_stats;

typedef struct bdpt_scene_info {
    uint32_t n_triangles;
    uint32_t n_materials;
    uint32_t n_lights;          /* size of the light collection actually used (whole world if none emissive) */
    uint32_t n_nodes;           /* bvh_node objects the reference would build */
    uint32_t max_depth;         /* depth of the tree */
    uint32_t reserved;
    uint64_t device_bytes;
} bdpt_scene_info;

const char* bdpt_last_error(void);

/* Library / device set-up.  No counterpart in the reference (it is a CPU program). */
int bdpt_context_create(int device_index, bdpt_context** out);
int bdpt_context_destroy(bdpt_context* ctx);

/* Upper limit, in bytes, of the device memory the context may take for the per-sample path state of
 * bdpt_render / bdpt_render_device (0 restores the default: 32 GiB, at most a quarter of the free device memory).
 * Larger batches mean fewer, longer kernel launches: a host that renders full frames at thousands of samples per
 * pixel gains about a tenth of throughput by allowing ~170 GB of the 288 GB.  Clipped to 90 % of the free memory.
 * No counterpart in the reference (host memory is not budgeted there). */
int bdpt_context_set_workspace_limit(bdpt_context* ctx, uint64_t bytes);

/* Which BVH walk answers bdpt_trace_closest (without counters), bdpt_trace_closest_device and bdpt_visible:
 * the traversal kernels of the render pipeline (default; four-wide hierarchy, FP32 box tests, FP64 triangle
 * tests) or a stackless walk over the reference's own tree in the reference's visiting order.  Both return the
 * bits of bvh_node::hit (src/acceleration/bvh.h:50-59) / camera::visible (src/camera.h:425-438); the second
 * exists for verification. */
#define BDPT_WALKER_PIPELINE 0
#define BDPT_WALKER_REFERENCE_TREE 1
int bdpt_context_set_query_walker(bdpt_context* ctx, int walker);

/* camera::initialize(), src/camera.h:160-197 (host arithmetic, glibc tan). */
int bdpt_camera_initialize(const bdpt_camera_public* cam, bdpt_camera_state* out);

/* Replaces `make_shared<bvh_node>(scene.world.to_hittable_list())` and the choice of the light
 * collection in main (src/main.cpp:66-67, src/acceleration/bvh.h:13-48,
 * src/objects/primatives/triangle.h:21-39,191-197): precomputes the triangles, builds the
 * reference-equivalent BVH on the host (same comparator, same std::sort, same box padding ->
 * same topology and left-to-right order) and uploads the flattened tables.
 *   tri_vertices : n_triangles * 9 doubles (v0 v1 v2), world order of triangle_collection::add
 *   tri_material : n_triangles indices into materials
 *   light_triangles : ids of the triangles of the `lights` collection in insertion order
 *                     (scene_loader.h:190-202); n_lights == 0 selects the whole world (main.cpp:67). */
int bdpt_scene_create(bdpt_context* ctx, const double* tri_vertices, const int32_t* tri_material,
                      uint32_t n_triangles, const bdpt_material* materials, uint32_t n_materials,
                      const uint32_t* light_triangles, uint32_t n_lights, bdpt_scene** out);
int bdpt_scene_destroy(bdpt_scene* scene);
int bdpt_scene_get_info(const bdpt_scene* scene, bdpt_scene_info* out);
/* Triangle ids in the reference's left-then-right leaf visiting order (a one-triangle node lists
 * its triangle twice, bvh.h:35-36).  *count receives the number of entries (<= 2*n_triangles). */
int bdpt_scene_dfs_leaves(const bdpt_scene* scene, uint32_t* out, uint32_t capacity, uint32_t* count);
/* Boxes of the bvh_node objects in pre-order: xmin xmax ymin ymax zmin zmax (6 doubles per node). */
int bdpt_scene_node_boxes(const bdpt_scene* scene, double* out, uint32_t capacity_nodes);


/* ---- composite scenes: instances, textures, participating media ----
 * The parts of the reference's object model that no scene file reaches but a C++ host can build:
 * translate / rotate_y (src/objects/hittable.h:46-157), checker / noise / image textures
 * (src/materials/textures/texture.h:28-87, perlin.h) and constant_medium with its isotropic phase function
 * (src/materials/volumes/constant_medium.h, src/materials/material.h:149-172). */
enum {
    BDPT_TEX_SOLID = 0,      /* rgb                                                  texture.h:14-26 */
    BDPT_TEX_CHECKER = 1,    /* scale, even, odd (texture indices)                   texture.h:28-50 */
    BDPT_TEX_NOISE = 2,      /* scale, data = bdpt_perlin_tables                     texture.h:76-87 */
    BDPT_TEX_IMAGE = 3       /* width, height, data = width*height*3 bytes           texture.h:52-74 */
};
/* What a `perlin` object holds after its constructor ran (perlin.h:53-57). */
typedef struct bdpt_perlin_tables {
    double randvec[256][3];
    int32_t perm_x[256], perm_y[256], perm_z[256];
} bdpt_perlin_tables;
typedef struct bdpt_texture {
    int32_t type;
    int32_t even, odd;       /* checker: the two sub-textures */
    int32_t width, height;   /* image: size in pixels; height <= 0 = "no image data" (solid cyan, texture.h:58) */
    int32_t reserved;
    double rgb[3];           /* solid */
    double scale;            /* checker: the constructor's `scale` (1/scale is formed as in texture.h:31); noise: `scale` */
    const void* data;        /* noise: const bdpt_perlin_tables*; image: 8-bit RGB rows, top row first, 3 bytes per pixel
                                (rtw_image::bdata, src/image/rtw_stb_image.h:98-112); read during the call only */
} bdpt_texture;

#define BDPT_OBJECT_ROTATE_Y 1   /* rotate_y(object, rotate_y_degrees), hittable.h:76-157 */
#define BDPT_OBJECT_TRANSLATE 2  /* translate(object, offset), hittable.h:46-74; applied after the rotation, i.e. the
                                    host's translate(rotate_y(mesh, angle), offset) */
#define BDPT_OBJECT_MEDIUM 4     /* constant_medium(boundary = the (transformed) mesh, density, phase), constant_medium.h */
/* One entry of the world's hittable_list that is not a bare triangle: a mesh of its own
 * (`bvh_node(hittable_list of its triangles)`, given in the mesh's own coordinates), optionally rotated about the
 * y axis, translated, and turned into a volume of constant density. */
typedef struct bdpt_object {
    uint32_t first_triangle;     /* the mesh: triangles [first_triangle, first_triangle + n_triangles) */
    uint32_t n_triangles;
    uint32_t flags;              /* BDPT_OBJECT_* */
    int32_t phase_material;      /* BDPT_OBJECT_MEDIUM: index of a BDPT_ISOTROPIC material */
    double rotate_y_degrees;
    double offset[3];
    double density;              /* BDPT_OBJECT_MEDIUM: neg_inv_density = -1/density, constant_medium.h:11 */
} bdpt_object;

/* bdpt_scene_create for a world that also holds such objects.  The world's hittable_list is: the triangles that
 * belong to no object, each as its own entry, and the objects, each entry at the place of its first triangle -
 * i.e. the order of the triangle array; `bvh_node(list)` is built over it as in src/main.cpp:66.  Two objects may name
 * the SAME range: they share one mesh (one bvh_node behind several wrappers, as shared_ptr allows in the reference)
 * and stand in the list in the order of their indices; different ranges must not overlap.  Light triangles (the `lights` collection) must be triangles that
 * belong to no object; n_lights == 0 selects all of those.
 * Every composite scene renders on the wavefront pipeline, like any other scene.  Without a medium its traversal
 * kernels transform the ray in front of an instanced triangle's test, its shading kernels evaluate
 * texture::value(u, v, p) at the hit (u, v formed again from ray and triangle by the triangle test's own operations)
 * and keep the resulting colour in the vertex record.  With a medium the rays are traced over the reference's own
 * trees in the reference's order instead, the pairs of a sample one after the other - constant_medium::hit draws its
 * random number INSIDE the BVH walk (constant_medium.h:40), so visiting order, running upper bound and the order of a
 * sample's shadow rays are part of the result.  For a composite scene BDPT_RENDER_MEGAKERNEL, BDPT_RENDER_COUNT_STATS and
 * BDPT_RENDER_UNIDIRECTIONAL take the single-kernel path (one thread per sample over the same trees), and so does
 * BDPT_RENDER_COUNT_WORK for a scene with a medium; bdpt_render_device_static and BDPT_RENDER_TIME_STAGES do not apply there.  bdpt_trace_closest* / bdpt_visible answer for every composite scene; query i draws from stream
 * (seed 0, pixel i, sample 0).  A hit inside a medium reports -(2 + object index) as its triangle id and u = v = 0 (the
 * reference leaves them unset); a hit on a shared mesh reports the triangle of the mesh.
 * An isotropic material with an image texture is refused: the reference reads unset u, v there (constant_medium.h:45-50, material.h:155). */
int bdpt_scene_create_composite(bdpt_context* ctx, const double* tri_vertices, const int32_t* tri_material,
                                uint32_t n_triangles, const bdpt_material* materials, uint32_t n_materials,
                                const bdpt_texture* textures, uint32_t n_textures, const bdpt_object* objects,
                                uint32_t n_objects, const uint32_t* light_triangles, uint32_t n_lights, bdpt_scene** out);

/* bvh_node::hit(r, interval(tmin, tmax), rec), src/acceleration/bvh.h:50-59, for n rays.
 *   rays : n * 8 doubles (origin[3] direction[3] tmin tmax)
 *   hits : n * 11 doubles (triangle id or -1, t, u, v, front_face, p[3], normal[3]) = hit_record
 *          (src/objects/hittable.h:10-27) plus the id of the triangle
 *   stats: optional. */
int bdpt_trace_closest(bdpt_context* ctx, const bdpt_scene* scene, const double* rays, uint64_t n,
                       double* hits, bdpt_stats* stats);
/* Same with rays / hits already in device memory (HBM-resident, asynchronous on `stream`,
 * a hipStream_t or NULL). */
int bdpt_trace_closest_device(bdpt_context* ctx, const bdpt_scene* scene, const void* d_rays, uint64_t n,
                              void* d_hits, void* stream);

/* camera::visible(a, b, world), src/camera.h:425-438, for n point pairs (a[3] b[3]); out: n bytes (0/1). */
int bdpt_visible(bdpt_context* ctx, const bdpt_scene* scene, const double* pairs, uint64_t n, uint8_t* out);

/* The worker loop of camera::render, src/camera.h:111-129 (-> get_ray :199-223, ray_color :245-253,
 * bidirectional_color :294-323 and everything below it).  framebuffer: image_width*image_height*3
 * doubles, row-major, the per-pixel SUM over the sqrt_spp^2 samples (camera.h:124); only owned
 * pixels are written.  stats: optional, filled when BDPT_RENDER_COUNT_STATS is set. */
int bdpt_render(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam,
                const bdpt_render_params* params, double* framebuffer, bdpt_stats* stats);
/* Same, framebuffer in device memory; returns after enqueueing on `stream` (hipStream_t or NULL).
 * Nothing in the call waits for the device (the table of owned rows is filled by a kernel).  The context keeps
 * the per-sample path state of the batches in flight in one device allocation made at the first call and grown on
 * demand up to the context's workspace limit (bdpt_context_set_workspace_limit); growing it frees the previous
 * allocation, which waits for earlier work.  One context renders one frame at a time. */
int bdpt_render_device(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam,
                       const bdpt_render_params* params, void* d_framebuffer, void* stream);
/* bdpt_render_device for callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc, code that
 * runs between the capture calls of others): same arguments, same framebuffer bit for bit, but the call allocates
 * nothing - the table of owned rows and the per-sample path state live in a workspace of 256 MiB that is part of the
 * library's code object (every GPU the library runs on holds its copy until the process ends), and the pipeline cuts
 * the frame into batches that fit it.  No counterpart in the reference.  Limits: the samples of ONE pixel must fit the
 * workspace - about samples x (220 + 290 x max_depth) + 21000 x max_depth x max_depth bytes, e.g. 1024 samples at depth
 * 12 need 7 MB, fine; 65536 samples at depth 40 do not - otherwise BDPT_ERR_UNSUPPORTED; flags other than BDPT_RENDER_COUNT_WORK,
 * BDPT_RENDER_TIME_KERNEL and BDPT_RENDER_UNIDIRECTIONAL are refused (they need buffers of their own).  All calls on one
 * GPU share that GPU's workspace, whatever the context or stream: the library orders them itself - a call's work starts
 * after the previous call's has ended (one event per GPU, made when the first context is) - so they never overlap, and
 * never race, on the device.  The call forks onto the library's internal streams and sets kernel attributes: it is
 * not meant to be recorded into a HIP graph itself.  Smaller batches than bdpt_render_device's make it the slower of
 * the two on large frames. */
int bdpt_render_device_static(bdpt_context* ctx, const bdpt_scene* scene, const bdpt_camera_state* cam,
                              const bdpt_render_params* params, void* d_framebuffer, void* stream);
/* Counters of the last bdpt_render_device call that requested them (synchronises). */
int bdpt_render_get_stats(bdpt_context* ctx, bdpt_stats* out);
/* Device time of the render kernel of the last call that had BDPT_RENDER_TIME_KERNEL set, from HIP
 * events recorded around the launch on the launch's own stream (waits for the kernel). */
int bdpt_render_get_kernel_ms(bdpt_context* ctx, float* out_ms);
/* Sum of the launch durations, and number of launches, of one traversal kernel in the last call that had
 * BDPT_RENDER_TIME_STAGES set: stage 0 = closest-hit traversal (bvh_node::hit under trace_path,
 * camera.h:334), stage 1 = visibility traversal (camera.h:437).  Launches of different batches overlap on
 * the device, so the sum exceeds the wall time of the call.  Waits for the kernels. */
int bdpt_render_get_stage_ms(bdpt_context* ctx, int stage, double* total_ms, uint64_t* launches);

/* colors_to_rgb8(), src/image/wpng.h:14-35: sum/spp -> clamp [0,0.999] -> sqrt -> *256 -> u8.
 * NaN (undefined behaviour in the reference) maps to 0. */
int bdpt_colors_to_rgb8(const double* framebuffer, int32_t width, int32_t height, int32_t samples_per_pixel,
                        uint8_t* out_rgb);
int bdpt_colors_to_rgb8_device(bdpt_context* ctx, const void* d_framebuffer, int32_t width, int32_t height,
                               int32_t samples_per_pixel, void* d_out_rgb, void* stream);

/* Multi-GPU: the ranks render disjoint pixel sets (shard_index / shard_count of bdpt_render_params) into zeroed
 * full-size framebuffers; this sums them in place on every rank with one all-reduce over xGMI, enqueued on
 * `stream` after the render.  No counterpart in the reference (one process, one framebuffer, camera.h:124); it
 * stands where the reference's worker threads write their rows into the shared framebuffer.  `rccl_comm` is the
 * caller's ncclComm_t (one process per GPU).  The RCCL library is looked up at the first call; a host that
 * already reduces through its own framework (e.g. torch.distributed) does not need this entry point. */
int bdpt_reduce_framebuffer(void* rccl_comm, void* d_framebuffer, uint64_t n_doubles, void* stream);

/* ---- scene files and image output (host only, no GPU work) ---- */
/* load_scene_from_yaml(path), src/scene/scene_loader.h:480-523, without yaml-cpp: the camera block (:427-476),
 * the named materials (:173-188, build_material :101-170), and the surfaces `TriMesh` (:244-272), `Sphere`
 * (16 x 32 tessellation, :212-242,274-294), `mesh` (:297-343) and `object` = Wavefront OBJ file with `v` / `f`
 * lines, negative indices and fan triangulation (:346-425).  The result is the flat input of bdpt_scene_create:
 * triangles in the order triangle_collection::add received them, materials in order of creation, and the ids of
 * the triangles the loader also put into `lights` (:190-202).  Errors (what the reference throws) return
 * BDPT_ERR_INVALID_ARGUMENT with the reference's message in bdpt_last_error(). */
typedef struct bdpt_scene_file bdpt_scene_file;
typedef struct bdpt_scene_file_view {
    bdpt_camera_public camera;
    const char* output_name;            /* camera::file_name, "image.png" unless the file names one */
    const double* tri_vertices;         /* n_triangles * 9 */
    const int32_t* tri_material;
    const bdpt_material* materials;
    const uint32_t* light_triangles;
    uint32_t n_triangles, n_materials, n_lights, reserved;
} bdpt_scene_file_view;
int bdpt_scene_file_load(const char* yaml_path, bdpt_scene_file** out);
int bdpt_scene_file_destroy(bdpt_scene_file* file);
/* Pointers stay valid until bdpt_scene_file_destroy. */
int bdpt_scene_file_get(const bdpt_scene_file* file, bdpt_scene_file_view* out);
/* write_png(filename, rgb, W, H), src/image/wpng.h:38-89: 8-bit RGB PNG at ./output/<file_name> (directory created). */
int bdpt_write_png(const char* file_name, const uint8_t* rgb, int32_t width, int32_t height);

/* Test hook: the parse tree of a YAML text (the subset the scene files use) as JSON, scalars as strings. */
int bdpt_debug_yaml_to_json(const char* text, char* out, uint64_t capacity);

/* Test hooks for the device-side building blocks the path is made of. */
/* std::cos / std::sin of src/core/vec3.h:154-155, src/materials/textures/texture.h:81 and src/objects/hittable.h:80-81
 * as evaluated on the GPU (|x| <= 2^30). */
int bdpt_device_sincos(bdpt_context* ctx, const double* x, uint64_t n, double* sin_out, double* cos_out);
/* first n values random_double() returns inside sample `sample` of pixel `pixel` (src/main.h:28-32). */
int bdpt_device_random_doubles(bdpt_context* ctx, uint64_t seed, uint64_t pixel, uint64_t sample, uint64_t n,
                               double* out);

/* element-wise IEEE checks on the GPU: op 0 a/b, 1 sqrt(a), 2 fma(a,b,a), 3 a*b+a (two roundings),
 * 4 fmin, 5 fmax, 6 a^5 as the device evaluates pow(.,5) of src/materials/material.h:129, 7 1/a,
 * 8 log(a) as the device evaluates the std::log of src/materials/volumes/constant_medium.h:40 */
int bdpt_device_arith(bdpt_context* ctx, int op, const double* a, const double* b, uint64_t n, double* out);

/* Test hook of the cross-check build (libbdpt_amd_tiecheck.so, `make tiecheck`): visibility rays whose far end lies ON a
 * surface tie with the end of their interval (src/camera.h:432-437), and whether the reference gets to test that surface's
 * triangle is decided by its own boxes (src/acceleration/aabb.h:41-66).  The library settles most of those ties in place;
 * the cross-check build also walks the reference's boxes for every one of them (and for every other accepted hit of a
 * visibility ray away from the triangle's outline), answers with the walk's result and counts:
 *   out[0] far-end ties seen, out[1] settled by the face bits alone, out[2] settled by the exact-plane distance,
 *   out[3] left to the boxes anyway, out[4] / out[5] wrong answers among [1] / [2],
 *   out[6] other accepted hits (taken as reached without a test), out[7] of those, not reached by the reference.
 * Counters cover every render and query since the last reset (process-wide per GPU).  BDPT_ERR_UNSUPPORTED in the normal
 * build.  Waits for the device. */
int bdpt_debug_tie_counts(bdpt_context* ctx, uint64_t* out8, int reset);

#ifdef __cplusplus
}
#endif
#endif
