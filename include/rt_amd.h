/*
 * rt_amd.h — C ABI of the MI355X (gfx950) path-tracing hot path.
 *
 * The reference program (skius/raytracer, one Rust file) exposes no FFI; the seam this library
 * replaces is the function boundary its hot path already sits behind:
 *
 *   Scene::from_objects   src/main.rs:750-774   -> rt_scene_from_objects / rt_scene_create
 *   Scene::hit            src/main.rs:776-814   -> rt_trace
 *   render::<W,H>         src/main.rs:977-1039  -> rt_render_accumulate / rt_render_device
 *   load_obj              src/main.rs:1130-1221 -> rt_mesh_load_obj
 *   PolygonMesh::rotate_* src/main.rs:627-658   -> rt_mesh_rotate_x/y/z
 *   PolygonMesh::translate src/main.rs:660-664  -> rt_mesh_translate
 *   Object::from_mesh     src/main.rs:463-469   -> rt_objects_push_mesh
 *   Object::sphere        src/main.rs:454-461   -> rt_objects_push_sphere
 *   image::io::Reader::open(..).decode()  src/main.rs:1226-1232 -> rt_image_load_png
 *   sphere_scene / lens_scene             src/main.rs:1223-1450 -> rt_scene_file_load (scenes as data)
 *   impl Into<Rgba<u8>> for RGBA + display loop  src/main.rs:29-37, 1099-1112 -> rt_quantize_rgba8
 *   BVH::build (crate bvh) src/main.rs:771       -> inside rt_scene_create; rt_scene_rebuild_on_device
 *
 * Conventions: every function returns RT_OK (0) or a negative RT_ERR_* code and never unwinds
 * across the boundary; rt_last_error() returns a thread-local message for the last failure.
 * All pointers are plain host pointers unless the name says "device". A scene may be used from
 * one host thread at a time. Geometry and shading are computed in IEEE binary64 without fused
 * multiply-add contraction, in the reference's operation order.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument / inconsistent description            */
#define RT_ERR_NO_DEVICE (-2) /* no HIP device, or the scene is not uploaded        */
#define RT_ERR_HIP (-3)       /* a HIP runtime call failed                          */
#define RT_ERR_IO (-4)        /* file could not be read or parsed                   */
#define RT_ERR_OOM (-5)       /* host or device allocation failed                   */

/* Material kinds — enum Material, src/main.rs:414-425. */
#define RT_MAT_OPAQUE 0
#define RT_MAT_DIELECTRIC 1

/* Object kinds — enum Object, src/main.rs:444-448. */
#define RT_OBJ_MESH 0
#define RT_OBJ_SPHERE 1

/* rt_render_params.flags */
#define RT_FLAG_NO_JITTER 1u  /* pixel-centre rays: x_offset = y_offset = 0 (parity checks) */
#define RT_FLAG_TIMING 2u     /* bracket the path kernel with HIP events                    */
#define RT_FLAG_COUNT_WORK 4u /* run the counting variant: node visits, triangle tests      */
#define RT_FLAG_NO_REGEN 8u   /* single kernel, one thread per path, no regeneration (A/B)  */
#define RT_FLAG_MEGAKERNEL 16u /* single persistent kernel with path regeneration (A/B); the  */
                              /* default is the wavefront form: one launch per bounce over a */
                              /* compacted ray queue                                         */
#define RT_FLAG_PIPELINE 32u  /* rt_render_device / rt_render_sum_f32_device only: the call may  */
                              /* return with its last kernels (the thin tail of long paths and   */
                              /* the sum into the canvas) queued on a stream of the library's    */
                              /* own instead of the caller's, so that they overlap the next      */
                              /* call. The canvas is complete after rt_scene_wait (or any        */
                              /* rt_scene_last_* call, or the next call without this flag); until */
                              /* then it must neither be freed nor handed to anything else.      */
                              /* Consecutive pipelined calls on one stream form a run: their     */
                              /* counters, launch log and timing accumulate. The reference's     */
                              /* display loop (src/main.rs:1099-1112) is such a run.             */

#define RT_FLAG_STORE 64u     /* the pixels the call renders are overwritten with the frame (canvas  */
                              /* = mean, sum buffer = sum) instead of added to: saves the caller a    */
                              /* clear of a buffer it fills once per frame; other pixels (tile split) */
                              /* are left alone                                                       */

/* One per object, indexed by object index. Material::Opaque / Material::Dielectric,
 * src/main.rs:414-425; ColorKind src/main.rs:386-390; NormalKind src/main.rs:335-339. */
typedef struct rt_material {
  int32_t kind;       /* RT_MAT_*                                              */
  int32_t color_tex;  /* texture id, or -1 for SolidColor(color)               */
  int32_t normal_tex; /* texture id of the normal map, or -1 for SimpleNormal  */
  int32_t reserved;
  double metallic;
  double diffuse;
  double how_metallic;
  double emission[3];
  double color[3];    /* SolidColor                                            */
  double ior;         /* Dielectric(index_of_refraction)                       */
} rt_material;

/* Decoded texture, 8-bit RGBA, row 0 at the top (what image::DynamicImage::get_pixel sees,
 * src/main.rs:353, 402). The library copies the pixels. */
typedef struct rt_texture {
  const uint8_t* rgba;
  uint32_t width;
  uint32_t height;
} rt_texture;

/* Flat scene description: what Scene::from_objects (src/main.rs:750-774) produces, with the
 * per-triangle HashMap lookups of src/main.rs:481-483 and 526-529 already resolved into
 * per-corner arrays. Triangles are in the reference's flat order (object order, then triangle
 * order); that order defines tie-breaking between equal hit distances (src/main.rs:791). */
typedef struct rt_scene_desc {
  uint32_t n_objects;
  const int32_t* object_kind;     /* [n_objects] RT_OBJ_*                                   */
  const rt_material* materials;   /* [n_objects]                                            */
  uint64_t n_triangles;
  const double* tri_vertices;     /* [n_triangles][3][3]  FlatTriangle.vertices             */
  const double* tri_normals;      /* [n_triangles][3][3]  vertex normals per corner         */
  const double* tri_uvs;          /* [n_triangles][3][2]  texture coordinates per corner    */
  const int32_t* tri_obj;         /* [n_triangles]        FlatTriangle.obj_idx              */
  const int32_t* tri_idx;         /* [n_triangles]        FlatTriangle.triangle_idx         */
  uint32_t n_spheres;
  const double* spheres;          /* [n_spheres][4]  centre xyz, radius                     */
  const int32_t* sphere_obj;      /* [n_spheres]     object index, ascending                */
  uint32_t n_textures;
  const rt_texture* textures;     /* [n_textures]                                           */
} rt_scene_desc;

/* enum Hit, src/main.rs:668-681. obj = -1 means no hit (None); tri = -1 means Hit::Sphere. */
typedef struct rt_hit {
  double t;
  double u;
  double v;
  int32_t obj;
  int32_t tri;
} rt_hit;

/* The constants of src/main.rs:1041-1052 and the camera of src/main.rs:978-1013 as run-time
 * parameters. The defaults of the reference are: 1600x1200, spp 10, max_depth 200,
 * fov_y_deg 60, dof_strength 0. */
typedef struct rt_render_params {
  uint32_t width;
  uint32_t height;
  uint32_t spp;          /* NUM_SAMPLES_PER_PIXEL of this call                              */
  uint32_t first_sample; /* index of this call's first sample in the pixel's sample stream  */
  uint64_t seed;         /* key of the counter-based generator                              */
  int32_t max_depth;     /* MAX_DEPTH                                                       */
  uint32_t flags;        /* RT_FLAG_*                                                       */
  double fov_y_deg;
  double dof_strength;   /* DOF_STRENGTH                                                    */
  uint32_t spp_total;    /* divisor of the per-pixel mean; 0 means spp (src/main.rs:1029)   */
  uint32_t tile_size;    /* 0: whole image. Otherwise square tiles of this many pixels ...  */
  uint32_t tile_rank;    /* ... of which this call renders those with index % tile_ranks    */
  uint32_t tile_ranks;   /*     == tile_rank (row-major tile index)                         */
} rt_render_params;

typedef struct rt_scene_info {
  uint64_t n_triangles;
  uint32_t n_spheres;
  uint32_t n_objects;
  uint32_t n_textures;
  uint32_t bvh_nodes;   /* interior nodes, 64 B each           */
  uint32_t bvh_leaves;
  uint32_t bvh_depth;   /* longest root-to-leaf path, in nodes */
  uint32_t uploaded;    /* device index + 1, or 0              */
  uint32_t bvh_spheres; /* spheres that have a place in the tree: the tree culls them like triangles */
  uint32_t reserved;
  uint64_t device_bytes;
} rt_scene_info;

/* Work counters of the last render / trace call made with RT_FLAG_COUNT_WORK (rays is
 * always maintained). rays counts calls of the nearest-hit query, primary and secondary. */
typedef struct rt_stats {
  uint64_t rays;
  uint64_t node_visits;   /* interior nodes fetched (two child boxes each) */
  uint64_t tri_tests;     /* ray/triangle tests                            */
  uint64_t sphere_tests;
  uint64_t shaded_hits;   /* hits whose shading record was fetched         */
  uint64_t paths;
  /* Divergence accounting (RT_FLAG_COUNT_WORK): waves that ran, and how many iterations of
   * each per-lane loop the waves executed (a wave runs a loop as long as its slowest lane). */
  uint64_t waves;
  uint64_t wave_iters_inner;  /* traversal steps                     */
  uint64_t wave_iters_leaf;   /* triangle tests                      */
  uint64_t wave_iters_reject; /* scatter-direction rejection sampling */
} rt_stats;

/* Durations of the last call made with RT_FLAG_TIMING, measured with HIP events on the
 * stream the kernels ran on. */
typedef struct rt_timing {
  float ms_paths;    /* sum over launches of the path (trace + shade) kernels */
  float ms_resolve;  /* sum over launches of the sample-resolve kernel       */
  uint32_t path_launches;
  uint32_t resolve_launches;
} rt_timing;

/* One kernel launch of the last render call, in launch order. Launches over the ray queue take
 * their extent from counts kept on the device, so a call can queue more of them than the frame
 * turns out to need; those show up with items = 0. */
typedef struct rt_launch_info {
  uint32_t kind;        /* 0 first bounce (camera + trace), 1 bounce (shade + trace) over the ray
                           queue, 2 tail (loops to the end), 3 single-kernel form, 4 resolve     */
  uint32_t items;       /* paths / queue entries / pixels the launch started with               */
  float ms;             /* duration (RT_FLAG_TIMING), else 0                                    */
  uint32_t folded;      /* 0: one launch. n > 0: the sums over n launches of this kind — a run of
                           thousands of pipelined calls keeps its latest launches one by one and
                           folds the older ones into one such entry per kind, at the front        */
  uint64_t rays;        /* nearest-hit queries the launch made                                   */
  uint64_t node_visits; /* RT_FLAG_COUNT_WORK                                                   */
  uint64_t tri_tests;   /* RT_FLAG_COUNT_WORK                                                   */
  uint64_t waves;             /* RT_FLAG_COUNT_WORK; see rt_stats                                */
  uint64_t wave_iters_inner;
  uint64_t wave_iters_leaf;
  uint64_t wave_iters_reject;
} rt_launch_info;

typedef struct rt_image rt_image;     /* a decoded texture: DynamicImage, src/main.rs:338, 389 */
typedef struct rt_mesh rt_mesh;       /* PolygonMesh, src/main.rs:566-574 */
typedef struct rt_objects rt_objects; /* Vec<Object> being assembled      */
typedef struct rt_scene rt_scene;     /* Scene, src/main.rs:747           */

/* ---- host side: meshes and scene assembly (no GPU needed) ------------------------------ */

/* load_obj, src/main.rs:1130-1221: "v", "vt", "f" lines; vertex normals are synthesised from
 * face normals, "vn" lines are ignored. */
int rt_mesh_load_obj(const char* path, rt_mesh** out);
int rt_mesh_clone(const rt_mesh* mesh, rt_mesh** out);
void rt_mesh_destroy(rt_mesh* mesh);
int rt_mesh_counts(const rt_mesh* mesh, uint64_t* n_vertices, uint64_t* n_triangles, uint64_t* n_uvs);
/* PolygonMesh::rotate_x/y/z, src/main.rs:627-658 (vertices and vertex normals). */
int rt_mesh_rotate_x(rt_mesh* mesh, double angle);
int rt_mesh_rotate_y(rt_mesh* mesh, double angle);
int rt_mesh_rotate_z(rt_mesh* mesh, double angle);
/* PolygonMesh::translate, src/main.rs:660-664 (vertices only). */
int rt_mesh_translate(rt_mesh* mesh, const double t[3]);

/* image::io::Reader::open(path).decode(), src/main.rs:1226-1232, 1273-1279, as the 8-bit RGBA
 * pixels get_pixel returns (src/main.rs:353, 402). Non-interlaced PNG, 8 bits per sample or a
 * palette; gamma and colour-profile chunks are ignored, as the reference's decoder ignores them. */
int rt_image_load_png(const char* path, rt_image** out);
void rt_image_destroy(rt_image* img);
/* *rgba points at [height][width][4] bytes owned by the image. */
int rt_image_info(const rt_image* img, uint32_t* width, uint32_t* height, const uint8_t** rgba);
/* No counterpart in the reference (it draws into a window, src/main.rs:1116-1122): writes
 * [height][width][4] 8-bit RGBA, e.g. the output of rt_quantize_rgba8, as a PNG file. */
int rt_image_write_png(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height);

int rt_objects_create(rt_objects** out);
void rt_objects_destroy(rt_objects* objs);
/* Registers a decoded texture; returns its id (>= 0) or a negative error. */
int rt_objects_add_texture(rt_objects* objs, const uint8_t* rgba, uint32_t width, uint32_t height);
/* Object::from_mesh, src/main.rs:463-469. Returns the object index or a negative error. */
int rt_objects_push_mesh(rt_objects* objs, const rt_mesh* mesh, const rt_material* mat);
/* Object::sphere, src/main.rs:454-461. Returns the object index or a negative error. */
int rt_objects_push_sphere(rt_objects* objs, const double centre[3], double radius, const rt_material* mat);

/* Scene description file: the calls sphere_scene() / lens_scene() make (src/main.rs:1270-1450,
 * 1223-1268) read from text instead of being compiled in (src/main.rs:1077-1078). One statement
 * per line, '#' comments, paths relative to the file:
 *   texture NAME FILE.png
 *   mesh FILE.obj [rotate_x DEG | rotate_y DEG | rotate_z DEG | translate X Y Z]... MATERIAL
 *   sphere X Y Z RADIUS MATERIAL
 *   render [width W] [height H] [spp S] [its N] [seed K] [max_depth D] [fov F] [dof X]
 *   MATERIAL = opaque [color R G B | color texture NAME] [normal NAME] [metallic M]
 *              [how_metallic H] [diffuse D] [emission R G B]  |  dielectric IOR
 * Angles become DEG * PI / 180.0 as at src/main.rs:1340. params (may be NULL) receives the
 * reference's constants overridden by the file's render statement; its (may be NULL) the number
 * of render calls to accumulate (the reference's display loop has no end; default 1). */
int rt_scene_file_load(const char* path, rt_objects** out, rt_render_params* params, uint32_t* its);
/* The constants of src/main.rs:1041-1052 and 978: 1600x1200, 10 spp, depth 200, fov 60, dof 0. */
void rt_render_params_default(rt_render_params* p);

/* Scene::from_objects, src/main.rs:750-774: flattens meshes in object order and builds the
 * acceleration structure (replaces BVH::build of crate bvh 0.6.0, src/main.rs:771). */
int rt_scene_from_objects(const rt_objects* objs, rt_scene** out);
/* Same, from an already flat description. All inputs are copied. */
int rt_scene_create(const rt_scene_desc* desc, rt_scene** out);
/* The same two with the tree named: RT_TREE_HOST is what they do by themselves (binned surface-
 * area build on the host's threads: 0.25 s for a million triangles, 2.5 s for fifteen million).
 * RT_TREE_NONE leaves the scene without a tree, for callers that build it on the GPU at once
 * (rt_scene_rebuild_on_device: a tenth of the time, frames 6-17 % slower); until then the scene
 * cannot be uploaded, traced or rendered (RT_ERR_INVALID). */
#define RT_TREE_HOST 0u
#define RT_TREE_NONE 1u
int rt_scene_from_objects_with(const rt_objects* objs, uint32_t tree, rt_scene** out);
int rt_scene_create_with(const rt_scene_desc* desc, uint32_t tree, rt_scene** out);
void rt_scene_destroy(rt_scene* scene);
int rt_scene_get_info(const rt_scene* scene, rt_scene_info* out);
/* Copies the flat triangle arrays out (any pointer may be NULL). Layouts as in rt_scene_desc. */
int rt_scene_export_flat(const rt_scene* scene, double* tri_vertices, double* tri_normals,
                         double* tri_uvs, int32_t* tri_obj, int32_t* tri_idx);
/* Copies the object list out: kinds and materials are [n_objects], spheres [n_spheres][4]
 * (centre, radius) with their object indices (any pointer may be NULL). */
int rt_scene_export_objects(const rt_scene* scene, int32_t* object_kind, rt_material* materials,
                            double* spheres, int32_t* sphere_obj);
/* Size of texture `id`, and its [height][width][4] bytes if rgba is not NULL. */
int rt_scene_export_texture(const rt_scene* scene, uint32_t id, uint32_t* width, uint32_t* height, uint8_t* rgba);
/* Copies the acceleration structure out: nodes as bvh_nodes x 16 32-bit words (layout in
 * raytracer_amd/csrc/rt_types.h), and for each of the n_triangles + bvh_spheres slots in leaf order
 * what sits there: a flat triangle index, or n_triangles + i for the place of sphere i (the tree
 * culls the first 64 finite spheres like triangles, by their boxes; the sphere tests themselves
 * run on the spheres whose place a ray reached). */
int rt_scene_export_bvh(const rt_scene* scene, uint32_t* node_words, int32_t* leaf_order);
/* No counterpart in the reference. The launches over the ray queue walk a half-size copy of the
 * tree whose planes are 16-bit positions on a grid over the scene's bounds; this copies it out:
 * nodes as bvh_nodes x 8 32-bit words (twelve 16-bit planes: per axis child0.min, child0.max,
 * child1.min, child1.max; then the two child refs), lo_step = {lo x, y, z, step x, y, z} with
 * plane = lo + q * step. *usable = 0, and nothing is copied, for a scene that does not qualify
 * (a coordinate beyond 2^20 grid steps, i.e. a scene some sixteen times its own size away from
 * the origin, or a non-finite box): such a scene is walked through the binary32 nodes everywhere. */
int rt_scene_export_grid_bvh(const rt_scene* scene, uint32_t* node_words, float* lo_step, int32_t* usable);

/* ---- device side ------------------------------------------------------------------------ */

int rt_device_count(void);
/* Copies the scene into the HBM of `device` (idempotent for the same device). The kernels address
 * nodes and triangles with 32-bit byte offsets: a scene of 2^32 / 80 (53 687 091) triangles or
 * more, or of 2^26 interior nodes or more, is refused with RT_ERR_INVALID. */
int rt_scene_upload(rt_scene* scene, int device);

/* Replaces the scene's tree (built on the host when the scene was created; BVH::build,
 * src/main.rs:771) by one built on `device`, in milliseconds instead of tenths of a second for a
 * million triangles, at the price of somewhat slower traversal: for scenes that change from
 * frame to frame. Hit records do not depend on the tree. The scene must be uploaded again
 * afterwards (the render calls do that themselves). RT_ERR_INVALID, with the host tree kept, if
 * the new tree would be deeper than the traversal stack allows (62 levels).
 * Two constructions, both starting from one radix sort of the triangles along the Morton curve:
 *   RT_BUILD_PLOC  bottom-up clustering: every cluster looks a few places along the curve for
 *                  the neighbour it would share the smallest box with, mutual choices merge,
 *                  repeat until one is left; subtrees of up to four triangles become leaves where
 *                  the surface-area cost says so. Traverses 6-17 % slower than the host's tree.
 *                  On a soup of triangles whose boxes all overlap the clusters chain up; if the
 *                  tree gets too deep the call falls back to RT_BUILD_LBVH by itself.
 *   RT_BUILD_LBVH  the hierarchy implied by the Morton codes' common prefixes (Karras 2012):
 *                  a fixed handful of launches, trees that traverse 12-27 % slower. */
#define RT_BUILD_LBVH 0u
#define RT_BUILD_PLOC 1u
#define RT_BUILD_DEFAULT RT_BUILD_PLOC
int rt_scene_rebuild_on_device(rt_scene* scene, int device); /* RT_BUILD_DEFAULT */
int rt_scene_rebuild_on_device_with(rt_scene* scene, int device, uint32_t method);

/* Batch Scene::hit, src/main.rs:776-814. org3/dir3 are [n][3]; dir need not be unit length.
 * The result is the nearest hit of the exhaustive binary64 test over every triangle and sphere.
 * Two cases can differ from the Rust binary, whose candidates come from crate bvh 0.6.0: of two
 * triangles at exactly equal distance the one with the lower flat index wins here (the reference
 * keeps the first in the crate's traversal order, src/main.rs:791), and a grazing hit that the
 * crate's binary32 boxes and ray (src/main.rs:621, 777) cull is still found here. */
int rt_trace(rt_scene* scene, const double* org3, const double* dir3, size_t n, rt_hit* out);

/* Nearest hit of the pixel-centre camera ray of every pixel (render's ray set-up,
 * src/main.rs:999-1013, with x_offset = y_offset = 0). Outputs are [height][width]. */
int rt_primary_hits(rt_scene* scene, const rt_render_params* p, int32_t* obj, int32_t* tri,
                    double* t, double* u, double* v);

/* render::<W,H>, src/main.rs:977-1039: canvas[v][u][c] += mean over p->spp samples.
 * canvas is a host buffer [height][width][3]. rays_out (may be NULL) receives the number of
 * nearest-hit queries. */
int rt_render_accumulate(rt_scene* scene, const rt_render_params* p, double* canvas, uint64_t* rays_out);

/* Same with the canvas in device memory and the work queued on `stream` (a hipStream_t, NULL
 * for the default stream); returns without waiting. The scene remembers the stream (later calls
 * and rt_scene_wait synchronise with it: a call on another stream starts behind what the previous
 * call queued), so it must outlive the scene's next call.
 * The sizes of the launches over the ray queue are decided on the device, so the call waits for
 * nothing, on any frame: a frame of a shape (image size, samples, tile split, depth cap) the scene
 * has rendered before is queued with the launches that frame needed, a frame of a new shape with
 * launches that cover a queue of any length (slower; it is one frame). A frame can therefore be
 * captured into a HIP graph from the first call on. */
int rt_render_device(rt_scene* scene, const rt_render_params* p, double* device_canvas, void* stream);

/* No counterpart in the reference. rt_render_device for callers that cannot allocate (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): the sample buffer,
 * the running per-pixel totals and both ray queues come out of one static array in
 * device memory, RT_NOALLOC_SCRATCH_MIB MiB per GPU, held by every GPU the library runs on until
 * the process ends; the call makes no hipMalloc, hipFree, stream or event creation. The scene
 * must already be uploaded (rt_scene_upload allocates); RT_ERR_NO_DEVICE otherwise. The array
 * bounds the image: 24 + 248 bytes per pixel (296 with emissive materials) for one sample of
 * every pixel of the call plus 18 MiB must fit, i.e. about 1 100 000 pixels per call (1280x720 fits);
 * larger frames are RT_ERR_INVALID and are rendered as several calls over tile_size / tile_rank /
 * tile_ranks, or with rt_render_device. More samples per pixel only take more passes. The
 * canvas holds the same bits as after rt_render_device. RT_FLAG_PIPELINE and RT_FLAG_TIMING are
 * ignored. All calls on one GPU share that GPU's array: calls that overlap in time on one GPU
 * (two scenes, two streams, two host threads) race; one call at a time per GPU. */
#define RT_NOALLOC_SCRATCH_MIB 320u
int rt_render_device_noalloc(rt_scene* scene, const rt_render_params* p, double* device_canvas, void* stream);

/* Sample-split rendering for several GPUs: device_sum[v][u][c] += sum (not mean) of this
 * call's samples, as binary32 — the buffer a sum-reduction over ranks then combines. */
int rt_render_sum_f32_device(rt_scene* scene, const rt_render_params* p, float* device_sum, void* stream);

/* Waits for everything the scene has queued, on the caller's stream and on the library's own. */
int rt_scene_wait(rt_scene* scene);
/* The same wait on the device instead of the host: work queued on `stream` after this call
 * starts only when everything the scene has queued so far (on the stream of its render calls and
 * on the library's own) has finished. Returns at once and does not end a run of pipelined calls:
 * a reduction of the frame's sums can follow each pipelined call on a stream of its own while
 * the next frame renders. */
int rt_scene_stream_wait(rt_scene* scene, void* stream);

/* Waits for the scene's queued work, then reports (for a run of pipelined calls: the run's
 * totals). The launch log is bounded: a run keeps its latest few thousand launches one by one and
 * folds older ones into one entry per kind (rt_launch_info.folded); events and counter snapshots
 * are reused once read, so a frame loop may run for any length of time. */
int rt_scene_last_stats(rt_scene* scene, rt_stats* out);
int rt_scene_last_timing(rt_scene* scene, rt_timing* out);
/* Per-launch log of the last render call: copies up to `capacity` entries, *n_out = total. */
int rt_scene_last_launches(rt_scene* scene, rt_launch_info* out, uint32_t capacity, uint32_t* n_out);

/* ---- several GPUs from one process ------------------------------------------------------- */

/* The reference is one process (main, src/main.rs:1054-1128) and spreads render's pixels and
 * samples over whatever the machine has (rayon, src/main.rs:988, 994). A group does the same
 * over several GPUs of a node for such a caller: every member holds its own copy of the scene
 * in its own HBM; the calling thread queues every member's share of the frame on that member's
 * stream (a render call only queues work) and the GPUs run them side by side; behind them, ordered
 * by events, the first member's GPU reads the others' binary32 sums directly over xGMI (peer
 * access, no staging copies, every link into that GPU used at once) and adds the frame to the
 * canvas. Callers that run one process per GPU use rt_render_sum_f32_device and their own
 * collective instead. RT_GROUP_LOG=1 in the environment prints which devices the first one
 * reaches by peer access when a group is created. */
typedef struct rt_group rt_group;
#define RT_GROUP_MAX 16u
#define RT_SPLIT_SAMPLES 0u /* member g renders samples [g*spp/n, (g+1)*spp/n) of every pixel    */
#define RT_SPLIT_TILES 1u   /* member g renders the square tiles whose row-major index % n == g */

/* `devices` lists one HIP device per member; the first is where frames are combined. The scene
 * is only read (its host arrays are copied to each device) and may be destroyed afterwards.
 * Naming a device twice is allowed (two members then share that GPU): meant for rehearsing a
 * group on a machine with fewer GPUs. RT_ERR_HIP if the first device cannot reach another one
 * by peer access. */
int rt_group_create(const rt_scene* scene, const int* devices, uint32_t n_devices, rt_group** out);
void rt_group_destroy(rt_group* group);
int rt_group_size(const rt_group* group, uint32_t* n_members);

/* render::<W,H> over the group, src/main.rs:977-1039: canvas[v][u][c] += mean over p->spp
 * samples, canvas a host buffer as for rt_render_accumulate. p describes the whole frame:
 * tile_rank/tile_ranks must be zero, tile_size is the tile edge of RT_SPLIT_TILES (0: 64),
 * RT_FLAG_PIPELINE and RT_FLAG_TIMING are ignored. Every member's sum of samples is rounded to
 * binary32 once per pixel before the members are added in binary64, so a frame agrees with a
 * single GPU's to about 1e-7 relative, not bit for bit. Returns when the canvas is complete;
 * rays_out (may be NULL) receives the nearest-hit queries of all members. */
int rt_group_render_accumulate(rt_group* group, const rt_render_params* p, uint32_t split, double* canvas,
                               uint64_t* rays_out);
/* Same with the canvas in the memory of the group's first device, as part of a frame loop: the
 * call only queues work and returns. `stream` (a hipStream_t of the first device, NULL for its
 * default stream) orders the canvas: the frame is added behind everything the stream held when the
 * call was made (a clear of the canvas, the previous frame), and whatever is queued on the stream
 * afterwards sees the frame in the canvas. Nothing else on the device is waited for. Every member
 * keeps two sum buffers, so the members render frame k + 1 while the first device still combines
 * frame k. rt_group_wait collects a run of such calls. */
int rt_group_render_device(rt_group* group, const rt_render_params* p, uint32_t split, double* device_canvas, void* stream);
/* Waits for every frame queued so far. rays_out (may be NULL) receives the nearest-hit queries
 * all members answered in the frames since the previous rt_group_wait (or the group's creation).
 * RT_ERR_INVALID if one of those frames could not be queued completely (its call said so too). */
int rt_group_wait(rt_group* group, uint64_t* rays_out);
/* What member `member` did in the frames the latest rt_group_wait (or rt_group_render_accumulate)
 * covered: its device, the nearest-hit queries it answered in them, and the time its share of the
 * latest of them took on its GPU (between two events on the member's stream) — the load balance
 * of the split. */
int rt_group_member_info(const rt_group* group, uint32_t member, int* device, uint64_t* rays, double* ms);

/* impl Into<Rgba<u8>> for RGBA, src/main.rs:29-37, applied to canvas / its as the display loop
 * does at src/main.rs:1106-1112. Host buffers; out is [height][width][4]. */
int rt_quantize_rgba8(const double* canvas, uint32_t width, uint32_t height, uint32_t its, uint8_t* out);

/* No counterpart in the reference. The kernels divide a vector by its length (normalize,
 * src/main.rs:87-90) and colour bytes by 255 (src/main.rs:359, 404-406) with one refined
 * reciprocal per denominator instead of three full divisions; this runs n pseudo-random and
 * boundary operand triples through both forms on device 0 and counts the triples whose quotients
 * differ in any bit from the plain IEEE divisions (expected: 0). */
int rt_selftest_division(uint64_t n, uint64_t seed, uint64_t* mismatches);

const char* rt_last_error(void);

/* No counterpart in the reference. Twelve hex digits: a digest of the source files this library
 * was built from (made by the build, raytracer_amd/csrc/Makefile), "unknown" for a build that did
 * not pass it. Measurements record it, so that counters replayed from a file can be told from
 * counters of the library at hand. */
const char* rt_source_digest(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
