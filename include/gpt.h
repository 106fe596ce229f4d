/*
 * gpt.h -- C-ABI of the MI355X path-tracing hot path (libgpt_hip.so).
 *
 * The library replaces ONE function of the reference renderer: the per-pass
 * device entry point
 *
 *     extern "C" void launchKernel(int numPolys, Poly* polys, int numSpheres,
 *                                  Sphere* spheres, int numPixels, Color* pixels,
 *                                  int counter, Camera renderCamera);
 *
 * declared at  GPUPathTracer/GPUPathTracer/path_tracer_kernel.h:9-10  and
 * defined at   GPUPathTracer/GPUPathTracer/path_tracer_kernel.cu:760-882.
 *
 * Everything here is plain C: pointers, sizes, PODs.  No C++ or torch types.
 * All functions except launchKernel return a gpt_status; launchKernel keeps
 * the reference's convention (print to stderr + exit(EXIT_FAILURE),
 * cuda_safe_call.h:6-12).
 */
#ifndef GPT_H
#define GPT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#else
#include <stdbool.h>
#endif

/* ------------------------------------------------------------------------- */
/* Scene PODs.  Byte-for-byte the reference layouts, so a host array prepared */
/* for the reference can be handed over unchanged.  Define                    */
/* GPT_NO_POD_TYPES before including this header when the reference's own     */
/* headers (camera.h, sphere.h, ...) are already in the translation unit.     */
/* ------------------------------------------------------------------------- */
#ifndef GPT_NO_POD_TYPES

#if !defined(__VECTOR_TYPES_H__) && !defined(HIP_INCLUDE_HIP_AMD_DETAIL_HIP_VECTOR_TYPES_H) && !defined(GPT_HAVE_VECTOR_TYPES)
/* CUDA's float3 is 12 bytes, 4-aligned; float2 is 8 bytes, 8-aligned. */
typedef struct gpt_float3 { float x, y, z; } gpt_float3;
#if defined(__cplusplus)
typedef struct alignas(8) gpt_float2 { float x, y; } gpt_float2;
#else
typedef struct gpt_float2 { _Alignas(8) float x; float y; } gpt_float2;
#endif
#define GPT_FLOAT3 gpt_float3
#define GPT_FLOAT2 gpt_float2
#else
#define GPT_FLOAT3 float3
#define GPT_FLOAT2 float2
#endif

/* color.h:9 */
typedef GPT_FLOAT3 Color;

/* absorption_and_scattering_properties.h:7-12 (16 bytes) */
typedef struct AbsorptionAndScatteringProperties {
    Color absorptionCoefficient;
    float reducedScatteringCoefficient;
} AbsorptionAndScatteringProperties;

/* medium.h:8-13 (20 bytes) */
typedef struct Medium {
    float refractiveIndex;
    AbsorptionAndScatteringProperties absorptionAndScatteringProperties;
} Medium;

/* material.h:8-16 (60 bytes; hasTransmission is a 1-byte bool + 3 pad) */
typedef struct Material {
    Color diffuseColor;
    Color emittedColor;
    Color specularColor;
    bool  hasTransmission;
    Medium medium;
} Material;

/* sphere.h:7-14 (76 bytes) */
typedef struct Sphere {
    GPT_FLOAT3 position;
    float radius;
    Material material;
} Sphere;

/* poly.h:7-17 (120 bytes).  `position` is never read by the renderer. */
typedef struct Poly {
    GPT_FLOAT3 position;
    GPT_FLOAT3 p0;
    GPT_FLOAT3 p1;
    GPT_FLOAT3 p2;
    GPT_FLOAT3 n;
    Material material;
} Poly;

/* ray.h:6-11 (24 bytes) */
typedef struct Ray {
    GPT_FLOAT3 origin;
    GPT_FLOAT3 direction;
} Ray;

/* camera.h:6-14 (64 bytes: float2 is 8-aligned, so 4 pad bytes precede fov) */
typedef struct Camera {
    GPT_FLOAT2 resolution;
    GPT_FLOAT3 position;
    GPT_FLOAT3 view;
    GPT_FLOAT3 up;
    GPT_FLOAT2 fov;
    float apertureRadius;
    float focalDistance;
} Camera;

#if defined(__cplusplus)
static_assert(sizeof(Color) == 12, "Color layout");
static_assert(sizeof(AbsorptionAndScatteringProperties) == 16, "A&S layout");
static_assert(sizeof(Medium) == 20, "Medium layout");
static_assert(sizeof(Material) == 60, "Material layout");
static_assert(sizeof(Sphere) == 76, "Sphere layout");
static_assert(sizeof(Poly) == 120, "Poly layout");
static_assert(sizeof(Ray) == 24, "Ray layout");
static_assert(sizeof(Camera) == 64, "Camera layout");
#else
_Static_assert(sizeof(Material) == 60, "Material layout");
_Static_assert(sizeof(Sphere) == 76, "Sphere layout");
_Static_assert(sizeof(Poly) == 120, "Poly layout");
_Static_assert(sizeof(Camera) == 64, "Camera layout");
#endif

#endif /* GPT_NO_POD_TYPES */

/* ------------------------------------------------------------------------- */
/* Status codes                                                               */
/* ------------------------------------------------------------------------- */
typedef enum gpt_status {
    GPT_OK = 0,
    GPT_ERR_INVALID_ARGUMENT = 1,
    GPT_ERR_HIP = 2,            /* a HIP runtime call failed; see gpt_last_error() */
    GPT_ERR_NO_DEVICE = 3,      /* no usable gfx950 device                           */
    GPT_ERR_OUT_OF_MEMORY = 4,
    GPT_ERR_IO = 5,
    GPT_ERR_PARSE = 6,
    GPT_ERR_TOO_LARGE = 7
} gpt_status;

/* Text of the last failure on the calling thread ("" if none). */
const char* gpt_last_error(void);

/* ------------------------------------------------------------------------- */
/* Render parameters.  The reference hard-codes these as #defines             */
/* (path_tracer_kernel.cu:46-50); gpt_render_params_default() returns exactly */
/* those values (max_depth 30 included).  The fields the reference has no     */
/* counterpart for default to row_stride_is_width = 1 and use_kdtree = 1:     */
/* both deviate from its behaviour as described at the fields (launchKernel   */
/* itself keeps row_stride_is_width = 0).                                     */
/* ------------------------------------------------------------------------- */
typedef struct gpt_render_params {
    int    max_depth;         /* MAX_TRACE_DEPTH, 30                    (:47)  */
    float  ray_bias;          /* RAY_BIAS_DISTANCE, 0.0002 narrowed to float at its use (:48,:659) */
    double min_ray_weight;    /* MIN_RAY_WEIGHT, 0.00001, compared in double (:49,:583,:705) */
    float  ground_elevation;  /* HARD_CODED_GROUND_ELEVATION, -0.8f     (:50)  */
    int    ground_enabled;    /* 1: the infinite ground plane exists (as :510) */
    /* Pixel -> (x,y).  0: the reference's arithmetic, row stride =
     * resolution.y (:135-136; only right for square images).  1: row stride =
     * resolution.x (right for any image).  Identical for square images.       */
    int    row_stride_is_width;
    /* 0: brute-force loop over every triangle, as the reference (:536-557).
     * 1: kd-tree traversal returning the same hit as that loop (triangle index
     *    and the bits of t).  What that covers, asserted by the GPU tests
     *    (tests/test_gpu_kd_exactness.py): every ray the renderer itself traces on
     *    the benchmark configurations -- all rays entering depths 0..2 of a 1080p
     *    frame of the 1.49 M-triangle cow, the primary rays of the other scenes --
     *    ZERO differences.  Supported origin distance for arbitrary rays: 8 extents
     *    of the mesh's bounding box.  Within it, rays aimed EXACTLY at mesh vertices
     *    (the worst case) differ on at most 100 rays per million, and only where
     *    the loop's own float arithmetic does not report a geometric intersection:
     *    an edge-on silhouette triangle (determinant ~1e-11) whose test passes with
     *    exact zeros and noise for t, which the ray misses -- or grazes somewhere
     *    else -- in exact arithmetic.  The loop finds those because it tests every
     *    triangle with every ray; a traversal only tests the triangles of the cells
     *    the ray passes through up to its first hit.  Beyond 8 extents the count
     *    grows about linearly with the distance.                                  */
    int    use_kdtree;
    int    kd_small_threshold; /* <= this many triangles: SAH stage (cukd: 64).  Default 0: automatic -- trees for 12 and
                                * 96 are built and the one with the smaller expected traversal work is kept, which
                                * doubles the build time and the peak host memory of gpt_scene_create (and of every
                                * re-index launchKernel does after a scene change); name a threshold to build one tree */
    /* Passes traced together by gpt_render_range: every kernel launch covers the
     * live paths of this many consecutive pass counters (at most 512).  Results
     * do not depend on it; throughput does (every launch ends in a tail of its
     * longest rays), and so does memory: 128 bytes per pixel and pass in flight.
     * 0 (default): as many as fit in half of the free device memory, up to
     * 512. */
    int    passes_in_flight;
    int    reserved[6];
} gpt_render_params;

void gpt_render_params_default(gpt_render_params* p);

/* ------------------------------------------------------------------------- */
/* Scene: device-resident geometry + kd-tree + per-path state buffers          */
/* ------------------------------------------------------------------------- */
typedef struct gpt_scene gpt_scene;

/* Upload `spheres`/`polys` (HOST arrays in the reference layout, the arrays
 * PathTracer::createDeviceData builds at path_tracer.cpp:116-278) and build the
 * kd-tree over the polys: on the host for small meshes, by kernels on `device`
 * from 200 000 triangles (the same tree, word for word; see gpt_kdtree_build_gpu).
 * `device` is the HIP device ordinal. */
gpt_status gpt_scene_create(const Sphere* spheres, int numSpheres,
                            const Poly* polys, int numPolys,
                            const gpt_render_params* params, int device,
                            gpt_scene** out_scene);
gpt_status gpt_scene_destroy(gpt_scene* scene);

/* (Re)size the per-pixel buffers for `numPixels` pixels and clear the
 * accumulated image and its pass count (PathTracer::reset, path_tracer.cpp:88-91). */
gpt_status gpt_scene_reset(gpt_scene* scene, int numPixels);

/* One sample-per-pixel pass with seed `counter`; semantics of launchKernel:
 * `host_pixels` (numPixels x 3 floats, may be NULL) receives THIS pass's
 * radiance (overwritten, not accumulated) -- path_tracer_kernel.cu:869.
 * The pass is also added into the scene's device-resident image, which is
 * what PathTracer::render does on the host (path_tracer.cpp:100-103). */
gpt_status gpt_render_pass(gpt_scene* scene, const Camera* camera, int counter,
                           float* host_pixels);

/* Passes counter_begin .. counter_end-1 back to back, no host round trip in
 * between; image += each pass in counter order.  Asynchronous with respect to
 * the host unless `sync` is non-zero. */
gpt_status gpt_render_range(gpt_scene* scene, const Camera* camera,
                            int counter_begin, int counter_end, int sync);

/* Block until all queued work of the scene has finished. */
gpt_status gpt_scene_sync(gpt_scene* scene);

/* The accumulated image: sum over passes of per-pass radiance, numPixels x 4
 * floats (r,g,b,passes) on the device / 3 floats per pixel on the host. */
gpt_status gpt_accum_device_ptr(gpt_scene* scene, void** out_ptr, size_t* out_bytes);
gpt_status gpt_accum_download(gpt_scene* scene, float* host_rgb, int* out_passes);
/* Replace the accumulated image (e.g. with the RCCL-reduced one). */
gpt_status gpt_accum_upload(gpt_scene* scene, const float* host_rgb, int passes);

/* Multi-GPU (no counterpart in the reference, which drives one GPU): samples per
 * pixel shard across the GPUs of a node -- scene i, on its own device, renders its
 * own range of pass counters -- and this call sums the n accumulated images
 * (r, g, b and the pass count) with ONE RCCL all-reduce over xGMI, queued on each
 * scene's stream behind its rendering; it returns when every scene holds the sum.
 * The scenes must be on n different devices and have the same pixel count.
 * RCCL (librccl.so) is loaded on first use. */
gpt_status gpt_accum_allreduce(gpt_scene* const* scenes, int n);
/* Releases what the library keeps for the life of the process: the RCCL communicators
 * gpt_accum_allreduce caches per device list (ncclCommDestroy) and launchKernel's cached scene,
 * side stream and staging buffers.  Scenes created by the caller are not touched.  The library
 * can be used again afterwards (everything is set up again on demand). */
gpt_status gpt_shutdown(void);

/* Work counters since the last gpt_scene_reset. */
typedef struct gpt_stats {
    uint64_t path_bounces;        /* sum over passes and depths of live paths  */
    uint64_t live_at_depth[64];   /* live paths entering depth d, all passes   */
    uint64_t passes;
    uint64_t kd_nodes_visited;    /* only counted when built with stats on     */
    uint64_t kd_tris_tested;
    uint64_t hits;                /* closest-hit queries that found a primitive (same build) */
    double   device_ms_trace;     /* HIP-event time of the intersect kernels   */
    double   device_ms_total;     /* HIP-event time of whole passes            */
    uint64_t trace_launches;
    uint64_t kd_nodes_at_depth[64];   /* instrumented build only */
    uint64_t kd_tris_at_depth[64];
    /* SIMD utilisation of the traversal kernel (instrumented build only): how often a wave
     * executed a walk iteration (one tree level for every walking lane) / a 64-pair chunk of
     * the triangle phase -- divide kd_nodes_visited / kd_tris_tested by them for lanes per
     * iteration -- and the walk iterations summed over lanes holding a ray vs over waves. */
    uint64_t wave_iters_node, wave_iters_tri, busy_lane_rounds, wave_rounds;
} gpt_stats;
gpt_status gpt_get_stats(gpt_scene* scene, gpt_stats* out);

/* Profiling switches (both off by default).  time_kernels: bracket every
 * intersect kernel and every pass with HIP events on the scene's stream and
 * report the sums in gpt_stats.device_ms_*.  count_kd_work: run the
 * instrumented intersect kernel that counts kd nodes visited / triangles tested. */
gpt_status gpt_scene_set_profiling(gpt_scene* scene, int time_kernels, int count_kd_work);
/* Size of the scene's kd-tree, and the number of passes traced together per
 * launch that the last gpt_scene_reset settled on (0 before the first one). */
gpt_status gpt_scene_info(gpt_scene* scene, int* n_nodes, int* n_refs, int* tree_depth, int* passes_in_flight);

/* The SAH stage threshold the scene's kd-tree was built with (what kd_small_threshold = 0 chose). */
gpt_status gpt_scene_kd_small_threshold(gpt_scene* scene, int* out);

/* First-hit record of every primary ray of pass `counter` (debug/parity):
 * prim[i] = -1 miss, -2 ground, 0x40000000|s sphere s, otherwise triangle
 * index; t[i] = hit distance. Host arrays of numPixels entries. */
gpt_status gpt_debug_primary_hits(gpt_scene* scene, const Camera* camera, int counter,
                                  int32_t* prim, float* t);

/* Closest hit of arbitrary rays against the scene's triangles only (no ground,
 * no spheres), `use_kdtree` choosing traversal or the brute-force loop.
 * rays: n x 6 floats (origin, direction) on the host. tri[i] = -1 on a miss. */
gpt_status gpt_debug_intersect(gpt_scene* scene, const float* rays, int n, int use_kdtree,
                               int32_t* tri, float* t);

/* gpt_render_pass plus a record of what every bounce of every path did (debug/parity).
 * events[pixel * max_depth + depth]: 0 = the path was not alive at that depth, else the branch
 * traceRayKernel left through -- 1 scattered in a medium (:567-590), 2 specular reflection
 * (:665-676), 3 transmission (:678-683), 4 diffuse bounce (:685-703), 5 miss / sky (:709-716) --
 * plus 8 when the weight test (:583, :705) ended the path there.  prims[...]: the primitive hit at
 * that bounce (encoding of gpt_debug_primary_hits; -1 also where not alive).  Host arrays of
 * numPixels x max_depth entries; host_pixels as in gpt_render_pass. */
gpt_status gpt_debug_render_events(gpt_scene* scene, const Camera* camera, int counter,
                                   unsigned char* events, int32_t* prims, float* host_pixels);

/* The path queue entering depth `depth` of a batch of P passes starting at counter0 (experiments on ray
 * order): its first max_rays rays (n x 6 floats, origin and direction) in queue order.  Host arrays. */
gpt_status gpt_debug_dump_queue(gpt_scene* scene, const Camera* camera, int counter0, int P, int depth,
                                float* rays, int max_rays, int* out_n, unsigned* out_live);

/* The device's random numbers (debug/parity): out[s * n_draws + k] = k-th uniform of an engine
 * seeded with seeds[s] -- thrust::default_random_engine + uniform_real_distribution<float>(0,1)
 * as used at path_tracer_kernel.cu:139-142, 492-493.  Host arrays. */
gpt_status gpt_debug_rng_uniforms(int device, const uint32_t* seeds, int n_seeds, int n_draws, float* out);

/* The device's per-path helper functions on caller-given records (debug/parity): fn 0..11 =
 * randhash, positionAlongRay, findGroundPlaneIntersection, FindTriangleIntersect,
 * randomCosineWeightedDirectionInHemisphere, randomDirectionInSphere, computeReflectionDirection,
 * computeTransmissionDirection, computeFresnel, the reflect / transmit / Fresnel chain of a surface
 * hit, computeBackgroundColor, computeTransmission (path_tracer_kernel.cu:75-84, 194-475); fn 12 =
 * the traversal kernel's form of the triangle test on fn 3's records (out: accepted as a hit closer than
 * the initial 123456789, t).  `in` and
 * `out` are HOST arrays of n records of 32-bit words; the record layouts (1/1, 7/3, 7/4, 15/4, 5/3,
 * 2/3, 6/3, 8/3, 14/2, 8/8, 3/3, 4/3 and 15/2 words in / out) are those of
 * oracle/ref_build/helpers_dump.cpp, which runs the reference's own text on the same inputs.
 * Output words the library does not compute are 0xffffffff. */
gpt_status gpt_debug_helpers(int device, int fn, int n, const uint32_t* in, uint32_t* out);

/* gpt_debug_intersect for callers that cannot allocate device memory: tight serving
 * loops, harnesses that forbid hipMalloc.  It has no counterpart in the reference.
 * `d_rays` (n x 6 floats), `d_tri` and `d_t` (n entries each) are DEVICE pointers on
 * the scene's GPU; results are those of gpt_debug_intersect, bit for bit.  The work
 * runs on the scene's own non-blocking stream and the call waits for it
 * (hipStreamSynchronize), so it is NOT for use inside a HIP graph capture; and since
 * that stream does not order itself against the caller's streams, whatever writes
 * d_rays must have FINISHED before the call (results are complete when it returns).
 * The call allocates and frees nothing: its ray
 * queue, hit slots, counters and traversal-stack spill area live in one
 * file-scope device array of 8.6 MiB that every GPU the library runs on holds
 * until the process ends.  Limits (GPT_ERR_TOO_LARGE beyond them): at most
 * GPT_NOALLOC_MAX_RAYS rays per call, kd-tree depth at most 62.  All calls on one
 * GPU share that array: calls that overlap on one GPU race (serialise them);
 * the call returns after its work has finished. */
#define GPT_NOALLOC_MAX_RAYS 16384
gpt_status gpt_intersect_noalloc(gpt_scene* scene, const float* d_rays, int n, int use_kdtree,
                                 int32_t* d_tri, float* d_t);

/* ------------------------------------------------------------------------- */
/* kd-tree (host builder).  Flattened form, all arrays owned by the tree:      */
/*   nodes      : n_nodes x 2 uint32   {w0, w1}                                */
/*      inner   : w0 = (left_child_index << 2) | axis (axis 0..2),             */
/*                w1 = split position (float bits); right child = left + 1     */
/*      leaf    : w0 = (first_ref << 2) | 3, w1 = number of triangle refs      */
/*   refs       : n_refs  int32 triangle indices, leaf by leaf                 */
/*   bounds     : 6 floats, root box min xyz / max xyz                         */
/* ------------------------------------------------------------------------- */
typedef struct gpt_kdtree gpt_kdtree;
gpt_status gpt_kdtree_build(const float* tri_vertices /* n x 9 */, int n_tris,
                            int small_threshold, gpt_kdtree** out_tree);
/* The same tree, word for word, built by kernels on HIP device `device` (what
 * gpt_scene_create uses for meshes of 200 000 triangles and more; the
 * environment variable GPT_KD_BUILD=host|gpu overrides the choice there).
 * Counterpart in the reference: the cukd builder behind constructKDTree
 * (path_tracer_kernel.cu:740-756, cukd/kdtree.cu:81-150), with this library's
 * deterministic rules. */
gpt_status gpt_kdtree_build_gpu(const float* tri_vertices /* n x 9 */, int n_tris,
                                int small_threshold, int device, gpt_kdtree** out_tree);
gpt_status gpt_kdtree_info(const gpt_kdtree* tree, int* n_nodes, int* n_refs,
                           int* max_depth, float bounds[6]);
gpt_status gpt_kdtree_copy(const gpt_kdtree* tree, uint32_t* nodes /* 2*n_nodes */,
                           int32_t* refs /* n_refs */);
void       gpt_kdtree_free(gpt_kdtree* tree);

/* ------------------------------------------------------------------------- */
/* Reference-compatible entry point.                                          */
/* `polys` and `spheres` are DEVICE pointers to arrays in the reference        */
/* layout (as cudaMalloc'ed at path_tracer.cpp:281-284); `pixels` is a HOST    */
/* buffer of numPixels float3, overwritten with this pass's radiance.          */
/* The scene behind (polys, spheres) is uploaded/re-indexed on first use and   */
/* whenever the arrays' contents change.  Depth and the other constants are    */
/* the reference's (30 bounces etc.) unless overridden with                    */
/* gpt_launch_kernel_params().  Errors: message on stderr + exit(1).           */
/* A run of calls with consecutive counters and an unchanged camera (the       */
/* reference's render loop, path_tracer.cpp:94-108) is served from batches of  */
/* up to passes_in_flight passes (32 by default here) traced ahead into two    */
/* device buffers -- one being handed out, the next being traced -- and each    */
/* call still returns exactly the pass `counter` names.  The arrays' contents   */
/* are checksummed on EVERY call (the reference reads them on every call, so a  */
/* caller may edit them in place): passes traced ahead from changed arrays are  */
/* dropped.  gpt_launch_kernel_static_scene(1) switches the re-reading off.     */
/* ------------------------------------------------------------------------- */
#ifndef GPT_NO_POD_TYPES
void launchKernel(int numPolys, Poly* polys, int numSpheres, Sphere* spheres,
                  int numPixels, Color* pixels, int counter, Camera renderCamera);
#endif
/* Parameters used by subsequent launchKernel calls (NULL restores defaults). */
void gpt_launch_kernel_params(const gpt_render_params* params);
/* Non-zero: the caller promises not to change the contents of the arrays behind
 * an unchanged (pointer, count) pair (the reference uploads its scene once,
 * path_tracer.cpp:281-284): launchKernel stops re-reading them. */
void gpt_launch_kernel_static_scene(int is_static);
/* Device-memory helpers for callers of launchKernel that have no HIP runtime
 * of their own (the reference used cudaMalloc/cudaMemcpy/cudaFree). */
gpt_status gpt_device_alloc(void** out_ptr, size_t bytes);
gpt_status gpt_device_upload(void* dst_device, const void* src_host, size_t bytes);
gpt_status gpt_device_download(void* dst_host, const void* src_device, size_t bytes);
gpt_status gpt_device_free(void* ptr);

/* ------------------------------------------------------------------------- */
/* Callers and data formats either side of the path                            */
/* ------------------------------------------------------------------------- */

/* OBJ loading with the reference loader's semantics (objcore/objloader.cpp:18-122)
 * followed by PathTracer::prepMesh's face -> Poly conversion (path_tracer.cpp:52-69):
 * first three vertices of each face; normal = the normal at the face's first
 * normal index (not renormalised).  Faces given without normals get
 * normalize(cross(p1-p0, p2-p0)) (as obj.cpp:87-89 does for its VBOs) instead
 * of the reference's out-of-range read.  Polys get position (0,-0.8,0) and
 * `material` (path_tracer.cpp:275-278).  *out_polys is malloc'ed; free with
 * gpt_free(). bbox (may be NULL): min xyz, max xyz over all `v` lines. */
gpt_status gpt_obj_load(const char* path, const Material* material,
                        Poly** out_polys, int* out_numPolys, float bbox[6]);
void gpt_free(void* p);

/* The reference's default scene: the 14 spheres of path_tracer.cpp:217-271. */
int  gpt_default_scene_sphere_count(void);
void gpt_default_scene(Sphere* out_spheres /* 14 entries */);
/* Named material presets of path_tracer.cpp:124-215 ("red", "green", "orange",
 * "purple", "glass", "greenGlass", "marble", "something", "ketchup", "white",
 * "lightBlue", "gold", "steel", "light").  Returns GPT_ERR_INVALID_ARGUMENT
 * for an unknown name. */
gpt_status gpt_material_preset(const char* name, Material* out);

/* Named scenes.  The reference was shipped as five executables -- "CornellBox", "Diamond-NoDOF",
 * "Monkey", "Speed", "Subsurface" -- each made by editing the scene table, the mesh path and the
 * camera in the source (path_tracer.cpp:54, 116-278); only the names survive, the committed source is
 * the 14-sphere scene ("Default").  A preset restates what its name says with the reference's meshes
 * (`mesh_dir`: the directory holding box.obj, diamond.obj, monkey.obj, coil.obj; may be NULL for
 * "Default" and "Speed"), material presets and sphere table; placements are this library's.
 * *out_spheres / *out_polys are malloc'ed (free with gpt_free; NULL when the scene has none);
 * out_view (may be NULL) receives the orbit-camera parameters for gpt_orbit_camera. */
typedef struct gpt_preset_view {
    float yaw, pitch, radius, apertureRadius, fovx_degrees;
    float center[3];
} gpt_preset_view;
int         gpt_scene_preset_count(void);
const char* gpt_scene_preset_name(int index);
gpt_status  gpt_scene_preset(const char* name, const char* mesh_dir, Sphere** out_spheres, int* out_numSpheres,
                             Poly** out_polys, int* out_numPolys, gpt_preset_view* out_view);

/* InteractiveCamera defaults + buildRenderCamera (interactive_camera.cpp:15-25,
 * 66-87; display.cpp:78-85): orbit camera around `center`. */
void gpt_orbit_camera(float width, float height, float fovx_degrees,
                      float yaw, float pitch, float radius, float apertureRadius,
                      const float center[3], Camera* out);
void gpt_default_camera(float width, float height, Camera* out);

/* floatTo8Bit(sum / passes) per pixel (color.cpp:41-62, display.cpp:195-200):
 * gamma 1/2.2, scale 255, clamp, truncate.  out: numPixels x 3 bytes. */
void gpt_tonemap_8bit(const float* sum_rgb, int numPixels, int passes, unsigned char* out_rgb8);
/* Binary PPM (P6) of the tone-mapped image, row 0 at the top. */
gpt_status gpt_write_ppm(const char* path, const float* sum_rgb, int width, int height, int passes);
/* Little-endian PFM (PF) of sum/passes, raw float radiance. */
gpt_status gpt_write_pfm(const char* path, const float* sum_rgb, int width, int height, int passes);
/* OpenEXR 2 scanline file of sum/passes: uncompressed, 32-bit float channels B, G, R,
 * row 0 at the top (the headless replacement of the PBO display, display.cpp:195-216). */
gpt_status gpt_write_exr(const char* path, const float* sum_rgb, int width, int height, int passes);

#ifdef __cplusplus
}
#endif
#endif /* GPT_H */
