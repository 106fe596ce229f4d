// gpt_api.cpp -- host side of the C-ABI declared in include/gpt.h: scene upload,
// pass orchestration on one HIP stream, statistics, and the reference-compatible
// launchKernel entry point (path_tracer_kernel.cu:760-882).
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/gpt.h"
#include "gpt_device_structs.h"
#include "kd_traverse.h"
#include "kdtree_build.h"
#include "pt_device.h"

using pt::CamPre;
using pt::ShadeParams;
using pt::TriRec;

// kernels (gpt_kernels.hip)
extern "C" {
void gptk_camera(hipStream_t s, int grid, const DevGeometry* g, const CamPre* cam, int numPixels, int numPasses,
                 uint32_t seed0, float4* ray_o, float4* ray_d, float4* thr, float2* hit, float4* pass_rgb,
                 unsigned* counts, uint32_t* masks);
void gptk_intersect(hipStream_t s, int grid, const DevGeometry* g, int use_kd, int stats_on, int depth,
                    const float4* ray_o, const float4* ray_d, float2* hit, unsigned* counts, unsigned* work,
                    unsigned long long* stats, uint32_t* spill, size_t spill_stride, unsigned refill_below,
                    unsigned leaf_min);
int gptk_camera_masks_only(hipStream_t s, const DevGeometry* g, const CamPre* cam, int numPixels, int numPasses, unsigned* counts,
                           uint32_t* masks);
void gptk_shade(hipStream_t s, int grid, const DevGeometry* g, const ShadeParams* sp, const CamPre* cam, const uint32_t* masks,
                int depth, uint32_t seed0, int numPasses, const float4* in_o, const float4* in_d, const float4* in_thr, const float2* hit, float4* out_o,
                float4* out_d, float4* out_thr, float2* out_hit, float4* pass_rgb, unsigned* counts,
                unsigned long long* stats, unsigned char* dbg_events, int* dbg_prims, int dbg_stride);
void gptk_finish(hipStream_t s, int grid, int numPixels, int numPasses, const float4* pass_rgb, float4* image,
                 float* pass_out3);
void gptk_extract_pass(hipStream_t s, int grid, int numPixels, int numPasses, int pass, const float4* pass_rgb, float* out3);
void gptk_gather_records(hipStream_t s, size_t n_refs, const int32_t* refs, const uint32_t* dst, const pt::TriRec* tris,
                         unsigned long long* out);
void gptk_checksum(hipStream_t s, const uint32_t* words, size_t n, unsigned long long* out);
void gptk_collect_wave_iters(hipStream_t s, unsigned long long* stats);
void gptk_rng_uniforms(hipStream_t s, const uint32_t* seeds, int n_seeds, int n_draws, float* out);
int gptk_debug_helpers(hipStream_t s, int fn, int n, const uint32_t* in, uint32_t* out, int* words_in, int* words_out);
}

namespace {

thread_local std::string g_error;

gpt_status fail(gpt_status st, const std::string& msg) {
    g_error = msg;
    return st;
}

#define HIP_TRY(call)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            char buf_[512];                                                                   \
            snprintf(buf_, sizeof(buf_), "HIP error in file '%s' in line %i : %s.", __FILE__, \
                     __LINE__, hipGetErrorString(e_));                                        \
            return fail(GPT_ERR_HIP, buf_);                                                   \
        }                                                                                     \
    } while (0)

// Makes `device` the calling thread's current device for the lifetime of the guard and restores the
// caller's own device afterwards, so that no entry point moves the caller's later allocations and
// launches to another GPU.
struct DeviceGuard {
    int prev = -1;
    bool good = false;
    explicit DeviceGuard(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur == device) { good = true; return; }
        good = hipSetDevice(device) == hipSuccess;
        if (good) prev = cur;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
    bool ok() const { return good; }
};
#define GPT_ON_DEVICE(dev)                                             \
    DeviceGuard device_guard_(dev);                                    \
    if (!device_guard_.ok()) return fail(GPT_ERR_HIP, "hipSetDevice failed")

const int kBlock = 256;
#ifndef GPT_TRACE_GRID
#define GPT_TRACE_GRID 2048
#endif
const int kTraceGrid = GPT_TRACE_GRID;   // persistent grid of the intersect / shade kernels

struct MatKey {
    uint32_t w[15];
    bool operator<(const MatKey& o) const { return memcmp(w, o.w, sizeof(w)) < 0; }
};

MatKey mat_key(const Material& m) {
    MatKey k;
    memset(&k, 0, sizeof(k));
    memcpy(&k.w[0], &m.diffuseColor, 12);
    memcpy(&k.w[3], &m.emittedColor, 12);
    memcpy(&k.w[6], &m.specularColor, 12);
    k.w[9] = m.hasTransmission ? 1u : 0u;
    memcpy(&k.w[10], &m.medium, 20);
    return k;
}

struct MatTable {
    std::map<MatKey, int> index;
    std::vector<float> packed;   // 16 floats per material
    MatKey last_key;
    int last_id = -1;
    int add(const Material& m) {
        MatKey k = mat_key(m);
        if (last_id >= 0 && memcmp(k.w, last_key.w, sizeof(k.w)) == 0) return last_id;   // meshes repeat one material
        auto it = index.find(k);
        if (it != index.end()) { last_key = k; last_id = it->second; return it->second; }
        int id = (int)index.size();
        index[k] = id;
        last_key = k;
        last_id = id;
        const AbsorptionAndScatteringProperties& as = m.medium.absorptionAndScatteringProperties;
        const float rec[16] = {
            m.diffuseColor.x, m.diffuseColor.y, m.diffuseColor.z, m.medium.refractiveIndex,
            m.specularColor.x, m.specularColor.y, m.specularColor.z, m.hasTransmission ? 1.0f : 0.0f,
            m.emittedColor.x, m.emittedColor.y, m.emittedColor.z, as.reducedScatteringCoefficient,
            as.absorptionCoefficient.x, as.absorptionCoefficient.y, as.absorptionCoefficient.z, 0.0f};
        packed.insert(packed.end(), rec, rec + 16);
        return id;
    }
};

template <class T>
gpt_status upload(T** dst, const void* src, size_t bytes) {
    *dst = nullptr;
    HIP_TRY(hipMalloc((void**)dst, bytes > 0 ? bytes : 16));
    if (bytes > 0) HIP_TRY(hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice));
    return GPT_OK;
}

}  // namespace

struct gpt_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    gpt_render_params params;

    // geometry
    float4* d_spheres = nullptr;
    int* d_sphere_mat = nullptr;
    TriRec* d_tris = nullptr;        // per triangle (brute force)
    TriRec* d_leaf_tris = nullptr;   // per leaf reference (kd traversal)
    int32_t* d_refs = nullptr;       // leaf -> triangle list (indexed layout only)
    float4* d_tri_nm = nullptr;
    float4* d_mats = nullptr;
    pt::KdNode* d_nodes = nullptr;
    uint32_t* d_spill = nullptr;
    size_t spill_stride = 0;
    DevGeometry geom;
    int n_nodes = 0, n_refs = 0, tree_depth = 0;
    int kd_small_used = 0;           // the SAH stage threshold the tree was built with

    // per-pixel / per-path buffers
    int numPixels = 0;
    int batch = 1;                           // passes traced together per launch
    int trace_grid = kTraceGrid;             // workgroups of the intersect / shade launches of a batch (launchKernel leaves room for its checksum)
    unsigned refill_below = 28;              // k_intersect_kd: refill a wave when fewer lanes are busy
    unsigned leaf_min = 20;                  // k_intersect_kd: test triangles once this many lanes wait at a leaf
    float4* d_ray_o[2] = {nullptr, nullptr};
    float4* d_ray_d[2] = {nullptr, nullptr};
    float4* d_thr[2] = {nullptr, nullptr};
    float2* d_hit[2] = {nullptr, nullptr};   // per queue: closest hit of each ray (t, primitive)
    float4* d_pass = nullptr;
    uint32_t* d_sphere_mask = nullptr;   // per pixel: the spheres its primary rays can reach (k_camera_masks)
    float4* d_image = nullptr;
    float* d_pass3 = nullptr;
    unsigned* d_counters = nullptr;          // GPT_COUNTER_WORDS
    unsigned long long* d_stats = nullptr;   // GPT_STAT_WORDS
    unsigned long long* d_stats_scratch = nullptr;   // same size: for queries that must not touch the statistics
    uint64_t passes = 0;
    // gpt_debug_render_events only: where k_shade records every bounce's branch code and primitive
    unsigned char* dbg_events = nullptr;
    int* dbg_prims = nullptr;

    // timing
    bool profile = false;
    bool kd_stats = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<size_t, size_t>> trace_pairs;   // indices into ev_pool
    std::vector<std::pair<size_t, size_t>> total_pairs;
    double ms_trace = 0, ms_total = 0;
    uint64_t trace_launches = 0;
};

namespace {

gpt_status free_pixel_buffers(gpt_scene* s) {
    for (int k = 0; k < 2; k++) {
        if (s->d_ray_o[k]) HIP_TRY(hipFree(s->d_ray_o[k]));
        if (s->d_ray_d[k]) HIP_TRY(hipFree(s->d_ray_d[k]));
        if (s->d_thr[k]) HIP_TRY(hipFree(s->d_thr[k]));
        s->d_ray_o[k] = s->d_ray_d[k] = s->d_thr[k] = nullptr;
    }
    for (int i = 0; i < 2; i++) if (s->d_hit[i]) HIP_TRY(hipFree(s->d_hit[i]));
    if (s->d_pass) HIP_TRY(hipFree(s->d_pass));
    if (s->d_image) HIP_TRY(hipFree(s->d_image));
    if (s->d_pass3) HIP_TRY(hipFree(s->d_pass3));
    if (s->d_sphere_mask) HIP_TRY(hipFree(s->d_sphere_mask));
    s->d_hit[0] = s->d_hit[1] = nullptr; s->d_pass = nullptr; s->d_image = nullptr; s->d_pass3 = nullptr;
    s->d_sphere_mask = nullptr;
    s->numPixels = 0;
    return GPT_OK;
}

gpt_status get_event(gpt_scene* s, size_t* idx) {
    if (s->ev_used == s->ev_pool.size()) {
        hipEvent_t e;
        HIP_TRY(hipEventCreate(&e));
        s->ev_pool.push_back(e);
    }
    *idx = s->ev_used++;
    return GPT_OK;
}

// Fold finished event pairs into the running totals.  Requires a synchronised stream.
gpt_status collect_events(gpt_scene* s) {
    for (auto& p : s->trace_pairs) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[p.first], s->ev_pool[p.second]));
        s->ms_trace += ms;
        s->trace_launches++;
    }
    for (auto& p : s->total_pairs) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->ev_pool[p.first], s->ev_pool[p.second]));
        s->ms_total += ms;
    }
    s->trace_pairs.clear();
    s->total_pairs.clear();
    s->ev_used = 0;
    return GPT_OK;
}

CamPre make_cam(const Camera* c, int row_stride_is_width) {
    return pt::camera_precompute(pt::v3(c->position.x, c->position.y, c->position.z),
                                 pt::v3(c->view.x, c->view.y, c->view.z), pt::v3(c->up.x, c->up.y, c->up.z),
                                 c->resolution.x, c->resolution.y, c->fov.x, c->fov.y, c->apertureRadius,
                                 c->focalDistance, row_stride_is_width);
}

int pixel_grid(int numPixels) {
    int g = (numPixels + kBlock - 1) / kBlock;
    return std::max(1, std::min(g, 4096));
}

// Enqueue `P` consecutive passes (counters counter0 .. counter0+P-1) traced together:
// every kernel launch covers the live paths of all P passes.  pass3: optional device
// buffer of P x numPixels x 3 floats receiving every pass on its own as packed float3.
gpt_status enqueue_batch(gpt_scene* s, const Camera* cam, int counter0, int P, float* pass3) {
    const int N = s->numPixels;
    const uint32_t seed0 = (uint32_t)counter0;
    const CamPre cp = make_cam(cam, s->params.row_stride_is_width);
    ShadeParams sp;
    sp.ray_bias = s->params.ray_bias;
    sp.min_ray_weight = s->params.min_ray_weight;
    unsigned* counts = s->d_counters;
    unsigned* work = s->d_counters + GPT_MAX_DEPTH_SLOTS;
    const int use_kd = (s->params.use_kdtree && s->geom.n_tris > 0) ? 1 : 0;
    const long long total = (long long)N * P;

    size_t e0 = 0, e1 = 0;
    if (s->profile) { gpt_status st = get_event(s, &e0); if (st) return st; HIP_TRY(hipEventRecord(s->ev_pool[e0], s->stream)); }

    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(unsigned) * GPT_COUNTER_WORDS, s->stream));
    // every ray arrives with its closest ground / sphere hit: without triangles there is nothing to intersect
    const bool trace = s->geom.n_tris > 0 || s->kd_stats;
    // ... and then nothing but the first shade launch would read the primary rays: it creates them itself
    const bool fused = !trace && s->params.max_depth > 0 && total < (1ll << 32) && !getenv("GPT_NO_FUSED_CAMERA");
    const uint32_t* fused_masks = nullptr;
    if (fused) {
        if (gptk_camera_masks_only(s->stream, &s->geom, &cp, N, P, counts, s->d_sphere_mask)) fused_masks = s->d_sphere_mask;
    } else {
        gptk_camera(s->stream, pixel_grid((int)std::min<long long>(total, 1 << 30)), &s->geom, &cp, N, P, seed0,
                    s->d_ray_o[0], s->d_ray_d[0], s->d_thr[0], s->d_hit[0], s->d_pass, counts, s->d_sphere_mask);
    }
    // (the first shade launch initialises the pass radiance slots; without any bounce nobody does)
    if (s->params.max_depth == 0) HIP_TRY(hipMemsetAsync(s->d_pass, 0, (size_t)total * sizeof(float4), s->stream));
    const int grid = (int)std::max<long long>(1, std::min<long long>(s->trace_grid, (total + kBlock - 1) / kBlock));
    for (int depth = 0; depth < s->params.max_depth; depth++) {
        const int in = depth & 1, out = in ^ 1;
        size_t t0 = 0, t1 = 0;
        if (s->profile) { gpt_status st = get_event(s, &t0); if (st) return st; HIP_TRY(hipEventRecord(s->ev_pool[t0], s->stream)); }
        if (trace)
            gptk_intersect(s->stream, grid, &s->geom, use_kd, s->kd_stats ? 1 : 0, depth, s->d_ray_o[in], s->d_ray_d[in],
                           s->d_hit[in], counts, work, s->d_stats, s->d_spill, s->spill_stride, s->refill_below,
                           s->leaf_min);
        if (s->profile) {
            gpt_status st = get_event(s, &t1); if (st) return st;
            HIP_TRY(hipEventRecord(s->ev_pool[t1], s->stream));
            s->trace_pairs.push_back({t0, t1});
        }
        gptk_shade(s->stream, grid, &s->geom, &sp, (fused && depth == 0) ? &cp : nullptr, fused_masks, depth, seed0, P, s->d_ray_o[in], s->d_ray_d[in], s->d_thr[in],
                   s->d_hit[in], s->d_ray_o[out], s->d_ray_d[out], s->d_thr[out], s->d_hit[out], s->d_pass, counts,
                   s->d_stats, P == 1 ? s->dbg_events : nullptr, P == 1 ? s->dbg_prims : nullptr, s->params.max_depth);
    }
    gptk_finish(s->stream, pixel_grid(N), N, P, s->d_pass, s->d_image, pass3);
    if (s->kd_stats) gptk_collect_wave_iters(s->stream, s->d_stats);
    HIP_TRY(hipGetLastError());
    if (s->profile) {
        gpt_status st = get_event(s, &e1); if (st) return st;
        HIP_TRY(hipEventRecord(s->ev_pool[e1], s->stream));
        s->total_pairs.push_back({e0, e1});
    }
    s->passes += (uint64_t)P;
    return GPT_OK;
}

gpt_status check_render_args(gpt_scene* s, const Camera* cam) {
    if (!s || !cam) return fail(GPT_ERR_INVALID_ARGUMENT, "null scene or camera");
    const long long w = (long long)cam->resolution.x, h = (long long)cam->resolution.y;
    if (w <= 0 || h <= 0 || w * h != (long long)s->numPixels)
        return fail(GPT_ERR_INVALID_ARGUMENT,
                    "camera resolution does not match the pixel count given to gpt_scene_reset");
    return GPT_OK;
}

}  // namespace

extern "C" {

const char* gpt_last_error(void) { return g_error.c_str(); }
void gpt_set_last_error_(const char* msg) { g_error = msg ? msg : ""; }

void gpt_render_params_default(gpt_render_params* p) {
    memset(p, 0, sizeof(*p));
    p->max_depth = 30;
    p->ray_bias = (float)0.0002;
    p->min_ray_weight = 0.00001;
    p->ground_elevation = (float)-0.8;
    p->ground_enabled = 1;
    p->row_stride_is_width = 1;
    p->use_kdtree = 1;
    p->kd_small_threshold = 0;
    p->passes_in_flight = 0;
}

gpt_status gpt_scene_create(const Sphere* spheres, int numSpheres, const Poly* polys, int numPolys,
                            const gpt_render_params* params, int device, gpt_scene** out_scene) {
    if (!out_scene) return fail(GPT_ERR_INVALID_ARGUMENT, "out_scene is null");
    *out_scene = nullptr;
    if (numSpheres < 0 || numPolys < 0 || (numSpheres > 0 && !spheres) || (numPolys > 0 && !polys))
        return fail(GPT_ERR_INVALID_ARGUMENT, "bad sphere/poly arrays");
    if (numSpheres >= 0x40000000 || numPolys >= 0x40000000)
        return fail(GPT_ERR_TOO_LARGE, "too many primitives");
    gpt_render_params rp;
    if (params) rp = *params; else gpt_render_params_default(&rp);
    if (rp.max_depth < 0 || rp.max_depth > GPT_MAX_DEPTH)
        return fail(GPT_ERR_INVALID_ARGUMENT, "max_depth must be within 0..64");
    if (rp.kd_small_threshold < 0) rp.kd_small_threshold = 0;   // 0: chosen per scene, see below
    if (rp.passes_in_flight < 0) rp.passes_in_flight = 0;
    if (rp.passes_in_flight > 512) rp.passes_in_flight = 512;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(GPT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(GPT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    GPT_ON_DEVICE(device);

    gpt_scene* s = new gpt_scene();
    s->device = device;
    s->params = rp;
    if (const char* e = getenv("GPT_REFILL_BELOW")) s->refill_below = (unsigned)std::max(1, std::min(64, atoi(e)));
    if (const char* e = getenv("GPT_LEAF_MIN")) s->leaf_min = (unsigned)std::max(1, std::min(64, atoi(e)));
    gpt_status st = GPT_OK;
    do {
        if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess) {
            st = fail(GPT_ERR_HIP, "hipStreamCreate failed");
            break;
        }
        MatTable mt;
        std::vector<float> sph((size_t)numSpheres * 4);
        std::vector<int> sph_mat((size_t)numSpheres);
        for (int i = 0; i < numSpheres; i++) {
            sph[4 * (size_t)i + 0] = spheres[i].position.x;
            sph[4 * (size_t)i + 1] = spheres[i].position.y;
            sph[4 * (size_t)i + 2] = spheres[i].position.z;
            sph[4 * (size_t)i + 3] = spheres[i].radius * spheres[i].radius;   // the test needs radius^2 only (:255)
            sph_mat[(size_t)i] = mt.add(spheres[i].material);
        }
        std::vector<TriRec> recs((size_t)numPolys);
        std::vector<float> nm((size_t)numPolys * 4);
        std::vector<float> verts((size_t)numPolys * 9);
        for (int i = 0; i < numPolys; i++) {
            const Poly& p = polys[i];
            float* f = recs[(size_t)i].f;
            f[0] = p.p0.x; f[1] = p.p0.y; f[2] = p.p0.z;
            f[3] = p.p1.x - p.p0.x; f[4] = p.p1.y - p.p0.y; f[5] = p.p1.z - p.p0.z;   // edge1 (:230)
            f[6] = p.p2.x - p.p0.x; f[7] = p.p2.y - p.p0.y; f[8] = p.p2.z - p.p0.z;   // edge2 (:231)
            memcpy(&f[9], &i, 4);   // triangle index
            float* v = &verts[9 * (size_t)i];
            v[0] = p.p0.x; v[1] = p.p0.y; v[2] = p.p0.z;
            v[3] = p.p1.x; v[4] = p.p1.y; v[5] = p.p1.z;
            v[6] = p.p2.x; v[7] = p.p2.y; v[8] = p.p2.z;
            nm[4 * (size_t)i + 0] = p.n.x; nm[4 * (size_t)i + 1] = p.n.y; nm[4 * (size_t)i + 2] = p.n.z;
            int id = mt.add(p.material);
            memcpy(&nm[4 * (size_t)i + 3], &id, 4);
        }
        const bool timing = getenv("GPT_TIMING") != nullptr;
        auto tnow = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        double tt0 = tnow();
        gptkd::Tree tree;
        gptkd::BuildParams bp;
        if (const char* e = getenv("GPT_KD_COST")) bp.traversal_cost = (float)atof(e);
        if (const char* e = getenv("GPT_KD_EMPTY_RATIO")) bp.empty_ratio = (float)atof(e);
        // large meshes are indexed on the GPU (the same tree, word for word); GPT_KD_BUILD=host|gpu overrides
        bool on_gpu = numPolys >= 200000;
        if (const char* e = getenv("GPT_KD_BUILD")) on_gpu = strcmp(e, "gpu") == 0;
        auto build_with = [&](int small_threshold, gptkd::Tree& out) -> gpt_status {
            bp.small_threshold = small_threshold;
            if (on_gpu) {
                std::string berr;
                if (!gptkd::build_gpu(verts.data(), numPolys, bp, out, &berr)) {
                    // asked for explicitly: report; chosen by size: the host builder gives the same tree
                    if (getenv("GPT_KD_BUILD")) return fail(GPT_ERR_HIP, berr);
                    fprintf(stderr, "gpt_scene_create: %s; building the kd-tree on the host instead\n", berr.c_str());
                    (void)hipGetLastError();
                    on_gpu = false;
                    gptkd::build(verts.data(), numPolys, bp, out);
                }
            } else {
                gptkd::build(verts.data(), numPolys, bp, out);
            }
            return GPT_OK;
        };
        if (rp.kd_small_threshold > 0) {
            if ((st = build_with(rp.kd_small_threshold, tree))) break;
            s->kd_small_used = rp.kd_small_threshold;
        } else {
            // Automatic: the stage threshold that is right for a surface mesh (median splits with empty-space cuts down
            // to a dozen triangles) costs a volume of scattered triangles 10 % and vice versa.  Both trees are built and
            // the one with the smaller expected traversal work -- surface-area rule, a node step weighted 4 and a
            // triangle test 2.4, the weights that rank the measured kernel times of both kinds of scene -- is kept.
            gptkd::Tree other;
            if ((st = build_with(12, tree))) break;
            if ((st = build_with(96, other))) break;
            const double c12 = gptkd::expected_cost(tree, 4.0, 2.4), c96 = gptkd::expected_cost(other, 4.0, 2.4);
            if (c96 < c12) std::swap(tree, other);
            s->kd_small_used = c96 < c12 ? 96 : 12;
            if (timing) fprintf(stderr, "gpt_scene_create: expected cost %.1f (SAH stage from 12) / %.1f (from 96)\n", c12, c96);
        }
        double tt1 = tnow();
        s->n_refs = (int)tree.refs.size();
        s->tree_depth = tree.max_depth;
        // device storage: nodes clustered per 128-byte line, triangle records in leaf order
        std::vector<gptkd::Node> dev_nodes;
        gptkd::treelet_layout(tree, dev_nodes);
        s->n_nodes = (int)dev_nodes.size();
        std::vector<int32_t> dev_refs;
        if (getenv("GPT_NO_DFS_LEAVES")) dev_refs = tree.refs;
        else gptkd::leaf_order_refs(dev_nodes, tree.refs, dev_refs);
        // device copy only: inner nodes learn which of their children are empty leaves
        if (dev_nodes.size() > (size_t)PT_KD_CHILD_MASK) { st = fail(GPT_ERR_TOO_LARGE, "kd-tree has too many nodes"); break; }
        pt::kd_mark_empty_children(reinterpret_cast<pt::KdNode*>(dev_nodes.data()), dev_nodes.size());
        // Leaf triangles: records duplicated in leaf order (a leaf visit then costs ONE memory
        // latency: the kernel is bound by dependent loads, not by footprint); only a copy beyond
        // 4 GiB falls back to an index list into the per-triangle records.
        bool leaf_order = tree.refs.size() * sizeof(TriRec) <= (size_t)4 << 30;
        if (const char* e = getenv("GPT_LEAF_ORDER")) leaf_order = atoi(e) != 0;
        // In that copy a leaf's `first` is the offset of its run in 8-byte units, and a run starts at
        // the next 128-byte line whenever it would otherwise touch one line more than its length needs
        // (three 40-byte records fit a line): fewer cache lines per leaf visit.
        // (the records themselves are gathered on the device from the per-triangle array: the host only
        // numbers the runs)
        std::vector<uint32_t> ref_dst;      // per leaf-ordered reference: offset of its record, 8-byte units
        size_t leaf_words = 0;
        if (leaf_order) {
            const size_t unit = sizeof(TriRec) / 8, line = 128 / 8;
            const bool fit = getenv("GPT_NO_LINE_FIT") == nullptr;
            std::vector<uint32_t> leaf_at(dev_refs.size() + 1, UINT32_MAX);   // first reference -> leaf node
            for (size_t i = 0; i < dev_nodes.size(); i++)
                if ((dev_nodes[i].w0 & 3u) == 3u && dev_nodes[i].w1) leaf_at[dev_nodes[i].w0 >> 2] = (uint32_t)i;
            ref_dst.resize(dev_refs.size());
            bool ok = true;
            for (size_t f = 0; f < dev_refs.size() && ok;) {      // the runs tile the list
                const uint32_t nd = leaf_at[f];
                if (nd == UINT32_MAX) { ok = false; break; }
                const size_t cnt = dev_nodes[nd].w1, len = cnt * unit;
                size_t at = leaf_words;
                if (fit && (at % line + len + line - 1) / line > (len + line - 1) / line) at = (at + line - 1) / line * line;
                if ((at + len) >> 30) { ok = false; break; }
                for (size_t k = 0; k < cnt; k++) ref_dst[f + k] = (uint32_t)(at + k * unit);
                dev_nodes[nd].w0 = ((uint32_t)at << 2) | 3u;
                leaf_words = at + len;
                f += cnt;
            }
            if (!ok) { st = fail(GPT_ERR_TOO_LARGE, "leaf records exceed the 30-bit offset"); break; }
        }

        double tt2 = tnow();
        if ((st = upload(&s->d_spheres, sph.data(), sph.size() * 4))) break;
        if ((st = upload(&s->d_sphere_mat, sph_mat.data(), sph_mat.size() * 4))) break;
        if ((st = upload(&s->d_tris, recs.data(), recs.size() * sizeof(TriRec)))) break;
        if ((st = upload(&s->d_tri_nm, nm.data(), nm.size() * 4))) break;
        if ((st = upload(&s->d_mats, mt.packed.data(), mt.packed.size() * 4))) break;
        if ((st = upload(&s->d_nodes, dev_nodes.data(), dev_nodes.size() * 8))) break;
        if (leaf_order) {
            int32_t* d_refs_tmp = nullptr;
            uint32_t* d_dst_tmp = nullptr;
            if ((st = upload(&d_refs_tmp, dev_refs.data(), dev_refs.size() * 4))) break;
            if ((st = upload(&d_dst_tmp, ref_dst.data(), ref_dst.size() * 4))) { (void)hipFree(d_refs_tmp); break; }
            hipError_t he = hipMalloc((void**)&s->d_leaf_tris, std::max<size_t>(leaf_words * 8, 16));
            if (he == hipSuccess) he = hipMemset(s->d_leaf_tris, 0, std::max<size_t>(leaf_words * 8, 16));
            if (he == hipSuccess) {
                gptk_gather_records(nullptr, dev_refs.size(), d_refs_tmp, d_dst_tmp, s->d_tris, (unsigned long long*)s->d_leaf_tris);
                he = hipGetLastError();
            }
            if (he == hipSuccess) he = hipDeviceSynchronize();
            (void)hipFree(d_refs_tmp);
            (void)hipFree(d_dst_tmp);
            if (he != hipSuccess) { st = fail(GPT_ERR_HIP, std::string("leaf records: ") + hipGetErrorString(he)); break; }
        } else {
            if ((st = upload(&s->d_refs, dev_refs.data(), dev_refs.size() * 4))) break;
        }

        if (timing) fprintf(stderr, "gpt_scene_create: kd build %.3f s, device layout %.3f s, upload %.3f s\n", tt1 - tt0, tt2 - tt1, tnow() - tt2);
        s->spill_stride = (size_t)kTraceGrid * kBlock;
        const size_t spill_words = 2 * (size_t)(tree.max_depth + 2) * s->spill_stride;
        if (spill_words >> 32) { st = fail(GPT_ERR_TOO_LARGE, "kd-tree too deep for the traversal stack"); break; }
        if (hipMalloc((void**)&s->d_spill, spill_words * 4) != hipSuccess) { st = fail(GPT_ERR_OUT_OF_MEMORY, "spill buffer"); break; }
        if (hipMalloc((void**)&s->d_counters, sizeof(unsigned) * GPT_COUNTER_WORDS) != hipSuccess ||
            hipMalloc((void**)&s->d_stats, sizeof(unsigned long long) * GPT_STAT_WORDS) != hipSuccess ||
            hipMalloc((void**)&s->d_stats_scratch, sizeof(unsigned long long) * GPT_STAT_WORDS) != hipSuccess) {
            st = fail(GPT_ERR_OUT_OF_MEMORY, "counters");
            break;
        }
        if (hipMemset(s->d_stats, 0, sizeof(unsigned long long) * GPT_STAT_WORDS) != hipSuccess) { st = fail(GPT_ERR_HIP, "memset"); break; }

        DevGeometry& g = s->geom;
        g.spheres = s->d_spheres;
        g.sphere_mat = s->d_sphere_mat;
        g.n_spheres = numSpheres;
        g.n_tris = numPolys;
        g.tri_nm = s->d_tri_nm;
        g.mats = s->d_mats;
        g.tris = s->d_tris;
        g.kd.nodes = s->d_nodes;
        g.kd.leaf_tris = leaf_order ? s->d_leaf_tris : s->d_tris;
        g.kd.refs = leaf_order ? nullptr : s->d_refs;
        float ext = 0;
        for (int k = 0; k < 3; k++) {
            g.kd.bmin[k] = tree.bounds[k];
            g.kd.bmax[k] = tree.bounds[3 + k];
            ext = std::max(ext, tree.bounds[3 + k] - tree.bounds[k]);
        }
        g.kd.sigma = 2e-5f * ext;
        g.ground_elevation = rp.ground_elevation;
        g.ground_enabled = rp.ground_enabled;
    } while (0);
    if (st != GPT_OK) {
        std::string keep = g_error;
        gpt_scene_destroy(s);
        g_error = keep;
        return st;
    }
    *out_scene = s;
    return GPT_OK;
}

gpt_status gpt_scene_destroy(gpt_scene* s) {
    if (!s) return GPT_OK;
    DeviceGuard device_guard_(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    free_pixel_buffers(s);
    void* ptrs[] = {s->d_spheres, s->d_sphere_mat, s->d_tris, s->d_leaf_tris, s->d_refs, s->d_tri_nm, s->d_mats, s->d_nodes,
                    s->d_spill, s->d_counters, s->d_stats, s->d_stats_scratch};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (hipEvent_t e : s->ev_pool) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return GPT_OK;
}

gpt_status gpt_scene_reset(gpt_scene* s, int numPixels) {
    if (!s || numPixels <= 0) return fail(GPT_ERR_INVALID_ARGUMENT, "bad scene or pixel count");
    GPT_ON_DEVICE(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    gpt_status st = collect_events(s);
    if (st) return st;
    if (numPixels != s->numPixels) {
        if ((st = free_pixel_buffers(s))) return st;
        int batch = s->params.passes_in_flight;
        if (batch <= 0) {
            // automatic: queues, hit slots and per-pass radiance take 128 bytes per pixel and pass
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            const size_t budget = free_b / 2;
            batch = (int)std::max<size_t>(1, std::min<size_t>(512, budget / (128 * (size_t)numPixels)));
        }
        while (batch > 1 && (long long)batch * numPixels > (1ll << 30)) batch--;
        if ((long long)numPixels > (1ll << 30)) return fail(GPT_ERR_TOO_LARGE, "too many pixels");
        const size_t n = (size_t)numPixels * (size_t)batch;
        for (int k = 0; k < 2; k++) {
            HIP_TRY(hipMalloc((void**)&s->d_ray_o[k], n * sizeof(float4)));
            HIP_TRY(hipMalloc((void**)&s->d_ray_d[k], n * sizeof(float4)));
            HIP_TRY(hipMalloc((void**)&s->d_thr[k], n * sizeof(float4)));
        }
        for (int i = 0; i < 2; i++) HIP_TRY(hipMalloc((void**)&s->d_hit[i], n * sizeof(float2)));
        HIP_TRY(hipMalloc((void**)&s->d_pass, n * sizeof(float4)));
        HIP_TRY(hipMalloc((void**)&s->d_image, (size_t)numPixels * sizeof(float4)));
        HIP_TRY(hipMalloc((void**)&s->d_pass3, (size_t)numPixels * 3 * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&s->d_sphere_mask, (size_t)numPixels * sizeof(uint32_t)));
        s->numPixels = numPixels;
        s->batch = batch;
    }
    HIP_TRY(hipMemsetAsync(s->d_image, 0, (size_t)numPixels * sizeof(float4), s->stream));
    HIP_TRY(hipMemsetAsync(s->d_stats, 0, sizeof(unsigned long long) * GPT_STAT_WORDS, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    s->passes = 0;
    s->ms_trace = s->ms_total = 0;
    s->trace_launches = 0;
    return GPT_OK;
}

gpt_status gpt_render_pass(gpt_scene* s, const Camera* cam, int counter, float* host_pixels) {
    gpt_status st = check_render_args(s, cam);
    if (st) return st;
    GPT_ON_DEVICE(s->device);
    if ((st = enqueue_batch(s, cam, counter, 1, host_pixels ? s->d_pass3 : nullptr))) return st;
    if (host_pixels) {
        HIP_TRY(hipMemcpyAsync(host_pixels, s->d_pass3, (size_t)s->numPixels * 3 * sizeof(float),
                               hipMemcpyDeviceToHost, s->stream));
    }
    HIP_TRY(hipStreamSynchronize(s->stream));
    return collect_events(s);
}

gpt_status gpt_render_range(gpt_scene* s, const Camera* cam, int counter_begin, int counter_end, int sync) {
    gpt_status st = check_render_args(s, cam);
    if (st) return st;
    GPT_ON_DEVICE(s->device);
    for (int c = counter_begin; c < counter_end; c += s->batch) {
        const int P = std::min(s->batch, counter_end - c);
        if ((st = enqueue_batch(s, cam, c, P, nullptr))) return st;
        if (s->profile && s->ev_used > 8192) {
            HIP_TRY(hipStreamSynchronize(s->stream));
            if ((st = collect_events(s))) return st;
        }
    }
    if (sync) {
        HIP_TRY(hipStreamSynchronize(s->stream));
        return collect_events(s);
    }
    return GPT_OK;
}

gpt_status gpt_scene_sync(gpt_scene* s) {
    if (!s) return fail(GPT_ERR_INVALID_ARGUMENT, "null scene");
    GPT_ON_DEVICE(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    return collect_events(s);
}

gpt_status gpt_scene_set_profiling(gpt_scene* s, int time_kernels, int count_kd_work) {
    if (!s) return fail(GPT_ERR_INVALID_ARGUMENT, "null scene");
    gpt_status st = gpt_scene_sync(s);
    if (st) return st;
    s->profile = time_kernels != 0;
    s->kd_stats = count_kd_work != 0;
    return GPT_OK;
}

gpt_status gpt_accum_device_ptr(gpt_scene* s, void** out_ptr, size_t* out_bytes) {
    if (!s || !out_ptr || !s->d_image) return fail(GPT_ERR_INVALID_ARGUMENT, "no image (call gpt_scene_reset first)");
    *out_ptr = s->d_image;
    if (out_bytes) *out_bytes = (size_t)s->numPixels * sizeof(float4);
    return GPT_OK;
}

gpt_status gpt_accum_download(gpt_scene* s, float* host_rgb, int* out_passes) {
    if (!s || !s->d_image || !host_rgb) return fail(GPT_ERR_INVALID_ARGUMENT, "no image or null output");
    GPT_ON_DEVICE(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<float> tmp((size_t)s->numPixels * 4);
    HIP_TRY(hipMemcpy(tmp.data(), s->d_image, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < (size_t)s->numPixels; i++) {
        host_rgb[3 * i + 0] = tmp[4 * i + 0];
        host_rgb[3 * i + 1] = tmp[4 * i + 1];
        host_rgb[3 * i + 2] = tmp[4 * i + 2];
    }
    if (out_passes) *out_passes = s->numPixels > 0 ? (int)tmp[3] : 0;
    return collect_events(s);
}

gpt_status gpt_accum_upload(gpt_scene* s, const float* host_rgb, int passes) {
    if (!s || !s->d_image || !host_rgb) return fail(GPT_ERR_INVALID_ARGUMENT, "no image or null input");
    GPT_ON_DEVICE(s->device);
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<float> tmp((size_t)s->numPixels * 4);
    for (size_t i = 0; i < (size_t)s->numPixels; i++) {
        tmp[4 * i + 0] = host_rgb[3 * i + 0];
        tmp[4 * i + 1] = host_rgb[3 * i + 1];
        tmp[4 * i + 2] = host_rgb[3 * i + 2];
        tmp[4 * i + 3] = (float)passes;
    }
    HIP_TRY(hipMemcpy(s->d_image, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    return GPT_OK;
}

gpt_status gpt_get_stats(gpt_scene* s, gpt_stats* out) {
    if (!s || !out) return fail(GPT_ERR_INVALID_ARGUMENT, "null argument");
    gpt_status st = gpt_scene_sync(s);
    if (st) return st;
    unsigned long long h[GPT_STAT_WORDS];
    HIP_TRY(hipMemcpy(h, s->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    memset(out, 0, sizeof(*out));
    for (int d = 0; d < GPT_MAX_DEPTH; d++) {
        out->live_at_depth[d] = h[GPT_STAT_LIVE0 + d];
        out->path_bounces += h[GPT_STAT_LIVE0 + d];
    }
    out->passes = s->passes;
    out->kd_nodes_visited = h[GPT_STAT_KD_NODES];
    out->kd_tris_tested = h[GPT_STAT_KD_TRIS];
    out->hits = h[GPT_STAT_HITS];
    out->wave_iters_node = h[GPT_STAT_WAVE_ITERS_NODE];
    out->wave_iters_tri = h[GPT_STAT_WAVE_ITERS_TRI];
    out->busy_lane_rounds = h[GPT_STAT_BUSY_LANE_ROUNDS];
    out->wave_rounds = h[GPT_STAT_WAVE_ROUNDS];
    for (int d = 0; d < GPT_MAX_DEPTH; d++) {
        out->kd_nodes_at_depth[d] = h[GPT_STAT_NODES_D0 + d];
        out->kd_tris_at_depth[d] = h[GPT_STAT_TRIS_D0 + d];
    }
    out->device_ms_trace = s->ms_trace;
    out->device_ms_total = s->ms_total;
    out->trace_launches = s->trace_launches;
    return GPT_OK;
}

gpt_status gpt_scene_info(gpt_scene* s, int* n_nodes, int* n_refs, int* tree_depth, int* passes_in_flight) {
    if (!s) return fail(GPT_ERR_INVALID_ARGUMENT, "null scene");
    if (n_nodes) *n_nodes = s->n_nodes;
    if (n_refs) *n_refs = s->n_refs;
    if (tree_depth) *tree_depth = s->tree_depth;
    if (passes_in_flight) *passes_in_flight = s->numPixels > 0 ? s->batch : 0;
    return GPT_OK;
}

gpt_status gpt_scene_kd_small_threshold(gpt_scene* s, int* out) {
    if (!s || !out) return fail(GPT_ERR_INVALID_ARGUMENT, "null argument");
    *out = s->kd_small_used;
    return GPT_OK;
}

gpt_status gpt_debug_primary_hits(gpt_scene* s, const Camera* cam, int counter, int32_t* prim, float* t) {
    gpt_status st = check_render_args(s, cam);
    if (st) return st;
    if (!prim || !t) return fail(GPT_ERR_INVALID_ARGUMENT, "null output");
    GPT_ON_DEVICE(s->device);
    const int N = s->numPixels;
    const CamPre cp = make_cam(cam, s->params.row_stride_is_width);
    unsigned* counts = s->d_counters;
    unsigned* work = s->d_counters + GPT_MAX_DEPTH_SLOTS;
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(unsigned) * GPT_COUNTER_WORDS, s->stream));
    gptk_camera(s->stream, pixel_grid(N), &s->geom, &cp, N, 1, (uint32_t)counter, s->d_ray_o[0], s->d_ray_d[0],
                s->d_thr[0], s->d_hit[0], s->d_pass, counts, s->d_sphere_mask);
    const int grid = std::max(1, std::min(kTraceGrid, (N + kBlock - 1) / kBlock));
    // statistics are not touched by this query: the kernel gets the scene's scratch copy of the stats words
    unsigned long long* scratch_stats = s->d_stats_scratch;
    HIP_TRY(hipMemsetAsync(scratch_stats, 0, sizeof(unsigned long long) * GPT_STAT_WORDS, s->stream));
    const int use_kd = (s->params.use_kdtree && s->geom.n_tris > 0) ? 1 : 0;
    gptk_intersect(s->stream, grid, &s->geom, use_kd, 0, 0, s->d_ray_o[0], s->d_ray_d[0], s->d_hit[0], counts, work,
                   scratch_stats, s->d_spill, s->spill_stride, s->refill_below, s->leaf_min);
    HIP_TRY(hipGetLastError());
    std::vector<float> h((size_t)N * 2), ro((size_t)N * 4);
    HIP_TRY(hipMemcpyAsync(h.data(), s->d_hit[0], h.size() * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipMemcpyAsync(ro.data(), s->d_ray_o[0], ro.size() * 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    // queue slots are not in pixel order: each slot carries its pixel index
    for (int i = 0; i < N; i++) {
        int pixel;
        memcpy(&pixel, &ro[4 * (size_t)i + 3], 4);
        if (pixel < 0 || pixel >= N) return fail(GPT_ERR_HIP, "corrupt path queue");
        t[pixel] = h[2 * (size_t)i];
        memcpy(&prim[pixel], &h[2 * (size_t)i + 1], 4);
    }
    return GPT_OK;
}

// The path queue as it enters depth `depth` of a batch of P passes (experiments on ray order): the first
// `max_rays` rays (origin, direction: 6 floats each) in queue order; *out_n = how many were written,
// *out_live = the queue's length.  Renders the batch up to that depth as a side effect (statistics and image untouched
// beyond what the kernels add to the statistics words).
gpt_status gpt_debug_dump_queue(gpt_scene* s, const Camera* cam, int counter0, int P, int depth, float* rays, int max_rays,
                                int* out_n, unsigned* out_live) {
    gpt_status st = check_render_args(s, cam);
    if (st) return st;
    if (!rays || !out_n || P < 1 || P > s->batch || depth < 0 || depth >= s->params.max_depth) return fail(GPT_ERR_INVALID_ARGUMENT, "bad argument");
    GPT_ON_DEVICE(s->device);
    const int N = s->numPixels;
    const CamPre cp = make_cam(cam, s->params.row_stride_is_width);
    ShadeParams sp;
    sp.ray_bias = s->params.ray_bias;
    sp.min_ray_weight = s->params.min_ray_weight;
    unsigned* counts = s->d_counters;
    unsigned* work = s->d_counters + GPT_MAX_DEPTH_SLOTS;
    const int use_kd = (s->params.use_kdtree && s->geom.n_tris > 0) ? 1 : 0;
    const long long total = (long long)N * P;
    HIP_TRY(hipMemsetAsync(s->d_counters, 0, sizeof(unsigned) * GPT_COUNTER_WORDS, s->stream));
    gptk_camera(s->stream, pixel_grid((int)std::min<long long>(total, 1 << 30)), &s->geom, &cp, N, P, (uint32_t)counter0,
                s->d_ray_o[0], s->d_ray_d[0], s->d_thr[0], s->d_hit[0], s->d_pass, counts, s->d_sphere_mask);
    const int grid = (int)std::max<long long>(1, std::min<long long>(kTraceGrid, (total + kBlock - 1) / kBlock));
    for (int d = 0; d < depth; d++) {
        const int in = d & 1, out = in ^ 1;
        if (s->geom.n_tris > 0)
            gptk_intersect(s->stream, grid, &s->geom, use_kd, 0, d, s->d_ray_o[in], s->d_ray_d[in], s->d_hit[in], counts, work,
                           s->d_stats_scratch, s->d_spill, s->spill_stride, s->refill_below, s->leaf_min);
        gptk_shade(s->stream, grid, &s->geom, &sp, nullptr, nullptr, d, (uint32_t)counter0, P, s->d_ray_o[in], s->d_ray_d[in], s->d_thr[in],
                   s->d_hit[in], s->d_ray_o[out], s->d_ray_d[out], s->d_thr[out], s->d_hit[out], s->d_pass, counts,
                   s->d_stats_scratch, nullptr, nullptr, 0);
    }
    HIP_TRY(hipGetLastError());
    unsigned live = 0;
    HIP_TRY(hipMemcpyAsync(&live, counts + depth, 4, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    const size_t n = std::min<size_t>(live, (size_t)std::max(0, max_rays));
    std::vector<float> o(n * 4), d4(n * 4);
    const int q = depth & 1;
    HIP_TRY(hipMemcpy(o.data(), s->d_ray_o[q], n * 16, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(d4.data(), s->d_ray_d[q], n * 16, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; i++)
        for (int c = 0; c < 3; c++) { rays[6 * i + c] = o[4 * i + c]; rays[6 * i + 3 + c] = d4[4 * i + c]; }
    *out_n = (int)n;
    if (out_live) *out_live = live;
    return GPT_OK;
}

// One pass as gpt_render_pass renders it, plus what every bounce of every path did.
gpt_status gpt_debug_render_events(gpt_scene* s, const Camera* cam, int counter, unsigned char* events, int32_t* prims,
                                   float* host_pixels) {
    gpt_status st = check_render_args(s, cam);
    if (st) return st;
    if (!events || !prims) return fail(GPT_ERR_INVALID_ARGUMENT, "null output");
    GPT_ON_DEVICE(s->device);
    const size_t n = (size_t)s->numPixels * (size_t)std::max(1, s->params.max_depth);
    if (hipMalloc((void**)&s->dbg_events, n) != hipSuccess || hipMalloc((void**)&s->dbg_prims, n * 4) != hipSuccess) {
        st = fail(GPT_ERR_OUT_OF_MEMORY, "event buffers");
    } else if (hipMemsetAsync(s->dbg_events, 0, n, s->stream) != hipSuccess ||
               hipMemsetAsync(s->dbg_prims, 0xff, n * 4, s->stream) != hipSuccess) {
        st = fail(GPT_ERR_HIP, "event buffers: memset");
    } else {
        st = enqueue_batch(s, cam, counter, 1, host_pixels ? s->d_pass3 : nullptr);
        if (st == GPT_OK &&
            ((host_pixels && hipMemcpyAsync(host_pixels, s->d_pass3, (size_t)s->numPixels * 12, hipMemcpyDeviceToHost, s->stream) != hipSuccess) ||
             hipMemcpyAsync(events, s->dbg_events, n, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
             hipMemcpyAsync(prims, s->dbg_prims, n * 4, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
             hipStreamSynchronize(s->stream) != hipSuccess))
            st = fail(GPT_ERR_HIP, "event download");
    }
    (void)hipStreamSynchronize(s->stream);
    (void)hipFree(s->dbg_events);
    (void)hipFree(s->dbg_prims);
    s->dbg_events = nullptr;
    s->dbg_prims = nullptr;
    if (st == GPT_OK) st = collect_events(s);
    return st;
}

// The device's uniform streams for the given seeds (parity of the random numbers themselves).
gpt_status gpt_debug_rng_uniforms(int device, const uint32_t* seeds, int n_seeds, int n_draws, float* out) {
    if (!seeds || !out || n_seeds <= 0 || n_draws <= 0) return fail(GPT_ERR_INVALID_ARGUMENT, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GPT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(GPT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    GPT_ON_DEVICE(device);
    uint32_t* d_seeds = nullptr;
    float* d_out = nullptr;
    const size_t n_out = (size_t)n_seeds * (size_t)n_draws;
    gpt_status st = GPT_OK;
    if (hipMalloc((void**)&d_seeds, (size_t)n_seeds * 4) != hipSuccess || hipMalloc((void**)&d_out, n_out * 4) != hipSuccess) {
        st = fail(GPT_ERR_OUT_OF_MEMORY, "rng buffers");
    } else if (hipMemcpy(d_seeds, seeds, (size_t)n_seeds * 4, hipMemcpyHostToDevice) != hipSuccess) {
        st = fail(GPT_ERR_HIP, "rng upload");
    } else {
        gptk_rng_uniforms(nullptr, d_seeds, n_seeds, n_draws, d_out);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d_out, n_out * 4, hipMemcpyDeviceToHost) != hipSuccess)
            st = fail(GPT_ERR_HIP, "rng kernel");
    }
    (void)hipFree(d_seeds);
    (void)hipFree(d_out);
    return st;
}

// The device's per-path helper functions on caller-given records (parity with the reference's own text).
gpt_status gpt_debug_helpers(int device, int fn, int n, const uint32_t* in, uint32_t* out) {
    if (!in || !out || n <= 0) return fail(GPT_ERR_INVALID_ARGUMENT, "bad argument");
    int wi = 0, wo = 0;
    if (gptk_debug_helpers(nullptr, fn, 0, nullptr, nullptr, &wi, &wo) != 0) return fail(GPT_ERR_INVALID_ARGUMENT, "unknown helper function");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(GPT_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(GPT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    GPT_ON_DEVICE(device);
    uint32_t *d_in = nullptr, *d_out = nullptr;
    gpt_status st = GPT_OK;
    if (hipMalloc((void**)&d_in, (size_t)n * wi * 4) != hipSuccess || hipMalloc((void**)&d_out, (size_t)n * wo * 4) != hipSuccess) {
        st = fail(GPT_ERR_OUT_OF_MEMORY, "helper buffers");
    } else if (hipMemcpy(d_in, in, (size_t)n * wi * 4, hipMemcpyHostToDevice) != hipSuccess) {
        st = fail(GPT_ERR_HIP, "helper upload");
    } else {
        gptk_debug_helpers(nullptr, fn, n, d_in, d_out, nullptr, nullptr);
        if (hipGetLastError() != hipSuccess || hipMemcpy(out, d_out, (size_t)n * wo * 4, hipMemcpyDeviceToHost) != hipSuccess)
            st = fail(GPT_ERR_HIP, "helper kernel");
    }
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return st;
}

// Closest triangle for caller-supplied rays, through the SAME intersect kernels a render pass
// launches: the rays are laid out as a path queue of depth 0 whose hit slots start at "nothing hit".
gpt_status gpt_debug_intersect(gpt_scene* s, const float* rays, int n, int use_kdtree, int32_t* tri, float* t) {
    if (!s || !rays || !tri || !t || n < 0) return fail(GPT_ERR_INVALID_ARGUMENT, "bad argument");
    if (n == 0) return GPT_OK;
    GPT_ON_DEVICE(s->device);
    if (s->geom.n_tris == 0) {
        for (int i = 0; i < n; i++) { tri[i] = -1; t[i] = PT_T_START; }
        return GPT_OK;
    }
    std::vector<float> ho((size_t)n * 4), hd((size_t)n * 4), hh((size_t)n * 2);
    const int miss = PT_PRIM_MISS;
    for (int i = 0; i < n; i++) {
        for (int c = 0; c < 3; c++) {
            ho[4 * (size_t)i + c] = rays[6 * (size_t)i + c];
            hd[4 * (size_t)i + c] = rays[6 * (size_t)i + 3 + c];
        }
        memcpy(&ho[4 * (size_t)i + 3], &i, 4);
        memcpy(&hd[4 * (size_t)i + 3], &miss, 4);
        hh[2 * (size_t)i] = PT_T_START;
        memcpy(&hh[2 * (size_t)i + 1], &miss, 4);
    }
    float4 *d_o = nullptr, *d_d = nullptr;
    float2* d_h = nullptr;
    unsigned* d_cnt = nullptr;
    unsigned long long* d_st = nullptr;
    gpt_status st = GPT_OK;
    do {
        if (hipMalloc((void**)&d_o, (size_t)n * 16) != hipSuccess || hipMalloc((void**)&d_d, (size_t)n * 16) != hipSuccess ||
            hipMalloc((void**)&d_h, (size_t)n * 8) != hipSuccess ||
            hipMalloc((void**)&d_cnt, sizeof(unsigned) * GPT_COUNTER_WORDS) != hipSuccess ||
            hipMalloc((void**)&d_st, sizeof(unsigned long long) * GPT_STAT_WORDS) != hipSuccess) {
            st = fail(GPT_ERR_OUT_OF_MEMORY, "debug intersect buffers");
            break;
        }
        const unsigned un = (unsigned)n;
        if (hipMemcpyAsync(d_o, ho.data(), (size_t)n * 16, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
            hipMemcpyAsync(d_d, hd.data(), (size_t)n * 16, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
            hipMemcpyAsync(d_h, hh.data(), (size_t)n * 8, hipMemcpyHostToDevice, s->stream) != hipSuccess ||
            hipMemsetAsync(d_cnt, 0, sizeof(unsigned) * GPT_COUNTER_WORDS, s->stream) != hipSuccess ||
            hipMemsetAsync(d_st, 0, sizeof(unsigned long long) * GPT_STAT_WORDS, s->stream) != hipSuccess ||
            hipMemcpyAsync(d_cnt, &un, 4, hipMemcpyHostToDevice, s->stream) != hipSuccess) {
            st = fail(GPT_ERR_HIP, "debug intersect upload");
            break;
        }
        const int grid = std::max(1, std::min(kTraceGrid, (n + kBlock - 1) / kBlock));
        gptk_intersect(s->stream, grid, &s->geom, use_kdtree ? 1 : 0, 0, 0, d_o, d_d, d_h, d_cnt, d_cnt + GPT_MAX_DEPTH_SLOTS,
                       d_st, s->d_spill, s->spill_stride, s->refill_below, s->leaf_min);
        if (hipGetLastError() != hipSuccess ||
            hipMemcpyAsync(hh.data(), d_h, (size_t)n * 8, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
            hipStreamSynchronize(s->stream) != hipSuccess) {
            st = fail(GPT_ERR_HIP, "debug intersect launch");
            break;
        }
        for (int i = 0; i < n; i++) {
            t[i] = hh[2 * (size_t)i];
            memcpy(&tri[i], &hh[2 * (size_t)i + 1], 4);
        }
    } while (0);
    (void)hipFree(d_o); (void)hipFree(d_d); (void)hipFree(d_h); (void)hipFree(d_cnt); (void)hipFree(d_st);
    return st;
}

// ---- multi-GPU: one RCCL sum of the accumulated images ---------------------------
// Samples-per-pixel shard across GPUs: each GPU renders its own range of pass counters into its own
// scene, then the float4 (r, g, b, passes) images are summed once over xGMI.  RCCL is loaded on first
// use (the library has no link-time dependency on it; a process that already holds RCCL, such as one
// that imported torch, shares that copy).
namespace {
struct Rccl {
    void* handle = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllReduce)(const void* send, void* recv, size_t count, int datatype, int op, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
    bool ok = false;
};
const int kNcclFloat = 7, kNcclSum = 0;   // rccl.h: ncclFloat32, ncclSum

Rccl* rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.error = std::string("cannot load librccl: ") + dlerror(); return; }
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(r.handle, "ncclCommInitAll");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.handle, "ncclCommDestroy");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.handle, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.handle, "ncclGroupEnd");
        r.AllReduce = (decltype(r.AllReduce))dlsym(r.handle, "ncclAllReduce");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.handle, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.CommDestroy && r.GroupStart && r.GroupEnd && r.AllReduce && r.GetErrorString;
        if (!r.ok) r.error = "librccl lacks an expected symbol";
    });
    return &r;
}

// communicators are kept per device list: creating them costs far more than a reduce
std::mutex g_comm_mu;
std::map<std::vector<int>, std::vector<void*>> g_comms;
}  // namespace

gpt_status gpt_accum_allreduce(gpt_scene* const* scenes, int n) {
    if (!scenes || n <= 0) return fail(GPT_ERR_INVALID_ARGUMENT, "no scenes");
    std::vector<int> devs;
    for (int i = 0; i < n; i++) {
        if (!scenes[i] || !scenes[i]->d_image) return fail(GPT_ERR_INVALID_ARGUMENT, "scene without an image (call gpt_scene_reset first)");
        if (scenes[i]->numPixels != scenes[0]->numPixels) return fail(GPT_ERR_INVALID_ARGUMENT, "scenes differ in their pixel count");
        for (int d : devs)
            if (d == scenes[i]->device) return fail(GPT_ERR_INVALID_ARGUMENT, "two scenes on one device: each GPU takes part once");
        devs.push_back(scenes[i]->device);
    }
    Rccl& R = *rccl();
    if (!R.ok) return fail(GPT_ERR_HIP, R.error);
    int prev = -1;
    (void)hipGetDevice(&prev);
    std::lock_guard<std::mutex> lock(g_comm_mu);
    std::vector<void*>& comms = g_comms[devs];
    gpt_status st = GPT_OK;
    do {
        if (comms.empty()) {
            comms.assign((size_t)n, nullptr);
            const int rc = R.CommInitAll(comms.data(), n, devs.data());
            if (rc != 0) { comms.clear(); st = fail(GPT_ERR_HIP, std::string("ncclCommInitAll: ") + R.GetErrorString(rc)); break; }
        }
        // every scene's reduce is queued on that scene's own stream, i.e. behind its rendering
        int rc = R.GroupStart();
        for (int i = 0; i < n && rc == 0; i++) {
            if (hipSetDevice(devs[(size_t)i]) != hipSuccess) { rc = -1; break; }
            rc = R.AllReduce(scenes[i]->d_image, scenes[i]->d_image, (size_t)scenes[i]->numPixels * 4, kNcclFloat, kNcclSum,
                             comms[(size_t)i], scenes[i]->stream);
        }
        const int rc_end = R.GroupEnd();
        if (rc == 0) rc = rc_end;
        if (rc != 0) { st = fail(GPT_ERR_HIP, std::string("ncclAllReduce: ") + (rc > 0 ? R.GetErrorString(rc) : "hipSetDevice failed")); break; }
        for (int i = 0; i < n; i++) {
            if (hipSetDevice(devs[(size_t)i]) != hipSuccess || hipStreamSynchronize(scenes[i]->stream) != hipSuccess) {
                st = fail(GPT_ERR_HIP, "waiting for the reduce failed");
                break;
            }
        }
    } while (0);
    if (prev >= 0) (void)hipSetDevice(prev);
    return st;
}

// Ends what the library keeps for the process: the RCCL communicators of gpt_accum_allreduce and launchKernel's cached
// scene, side stream and staging buffers.  Scenes the caller created stay the caller's to destroy.
void gpt_shutdown_launch_cache_();
gpt_status gpt_shutdown(void) {
    gpt_shutdown_launch_cache_();
    std::lock_guard<std::mutex> lock(g_comm_mu);
    Rccl& R = *rccl();
    gpt_status st = GPT_OK;
    for (auto& entry : g_comms) {
        for (size_t i = 0; i < entry.second.size(); i++) {
            if (!entry.second[i]) continue;
            DeviceGuard guard(entry.first[i]);
            const int rc = R.ok ? R.CommDestroy(entry.second[i]) : 0;
            if (rc != 0) st = fail(GPT_ERR_HIP, std::string("ncclCommDestroy: ") + R.GetErrorString(rc));
        }
    }
    g_comms.clear();
    return st;
}

// ---- closest-hit query that allocates nothing -----------------------------------
// gpt_intersect_noalloc: gpt_debug_intersect for callers that cannot allocate device memory.
// Everything the query needs beyond its inputs and outputs -- the ray queue in the kernels'
// layout, the hit slots, the queue counters, the statistics words and the spill part of the
// traversal stacks -- lives in ONE file-scope device array.  Every GPU holds its own copy of that
// array at its own address; g_noalloc_ptr[device] is that address, looked up once per device.
#define GPT_NOALLOC_SPILL_LEVELS 64
#define GPT_NOALLOC_THREADS GPT_NOALLOC_MAX_RAYS   /* one traversal thread per ray at most */
#define GPT_NOALLOC_OFF_RAY_O 0ull
#define GPT_NOALLOC_OFF_RAY_D (GPT_NOALLOC_OFF_RAY_O + 16ull * GPT_NOALLOC_MAX_RAYS)
#define GPT_NOALLOC_OFF_HIT (GPT_NOALLOC_OFF_RAY_D + 16ull * GPT_NOALLOC_MAX_RAYS)
#define GPT_NOALLOC_OFF_COUNTERS (GPT_NOALLOC_OFF_HIT + 8ull * GPT_NOALLOC_MAX_RAYS)
#define GPT_NOALLOC_COUNTER_BYTES ((4ull * GPT_COUNTER_WORDS + 255ull) / 256ull * 256ull)
#define GPT_NOALLOC_OFF_STATS (GPT_NOALLOC_OFF_COUNTERS + GPT_NOALLOC_COUNTER_BYTES)
#define GPT_NOALLOC_STAT_BYTES ((8ull * GPT_STAT_WORDS + 255ull) / 256ull * 256ull)
#define GPT_NOALLOC_OFF_SPILL (GPT_NOALLOC_OFF_STATS + GPT_NOALLOC_STAT_BYTES)
#define GPT_NOALLOC_SPILL_BYTES (4ull * 2ull * GPT_NOALLOC_SPILL_LEVELS * GPT_NOALLOC_THREADS)   /* two words per stack entry */
#define GPT_NOALLOC_BYTES (GPT_NOALLOC_OFF_SPILL + GPT_NOALLOC_SPILL_BYTES)

static __device__ __align__(256) unsigned char scratch[GPT_NOALLOC_BYTES];

namespace {
const int kMaxDevices = 64;
std::mutex g_noalloc_mu;
unsigned char* g_noalloc_ptr[kMaxDevices];   // per device: address of ITS copy of `scratch` (null: not looked up yet)

// Caller rays (origin, direction: 6 floats) -> a depth-0 path queue whose hit slots say "nothing hit yet".
__global__ __launch_bounds__(256) void k_noalloc_pack(const float* __restrict__ rays, int n, float4* __restrict__ ray_o,
                                                      float4* __restrict__ ray_d, float2* __restrict__ hit,
                                                      unsigned* __restrict__ counts) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i == 0) counts[0] = (unsigned)n;
    if (i >= n) return;
    const float* r = rays + 6 * (size_t)i;
    ray_o[i] = make_float4(r[0], r[1], r[2], __int_as_float(i));
    ray_d[i] = make_float4(r[3], r[4], r[5], __int_as_float(PT_PRIM_MISS));
    hit[i] = make_float2(PT_T_START, __int_as_float(PT_PRIM_MISS));
}
__global__ __launch_bounds__(256) void k_noalloc_unpack(const float2* __restrict__ hit, int n, int32_t* __restrict__ tri,
                                                        float* __restrict__ t) {
    const int i = (int)(blockIdx.x * 256u + threadIdx.x);
    if (i >= n) return;
    const float2 h = hit[i];
    t[i] = h.x;
    tri[i] = __float_as_int(h.y);
}
}  // namespace

gpt_status gpt_intersect_noalloc(gpt_scene* s, const float* d_rays, int n, int use_kdtree, int32_t* d_tri, float* d_t) {
    if (!s || n < 0 || (n > 0 && (!d_rays || !d_tri || !d_t))) return fail(GPT_ERR_INVALID_ARGUMENT, "bad argument");
    if (n > GPT_NOALLOC_MAX_RAYS)
        return fail(GPT_ERR_TOO_LARGE, "gpt_intersect_noalloc: more than GPT_NOALLOC_MAX_RAYS (16384) rays in one call");
    if (use_kdtree && s->geom.n_tris > 0 && s->tree_depth + 2 > GPT_NOALLOC_SPILL_LEVELS)
        return fail(GPT_ERR_TOO_LARGE, "gpt_intersect_noalloc: the kd-tree is deeper than the fixed traversal stack (62 levels)");
    if (s->device < 0 || s->device >= kMaxDevices) return fail(GPT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (n == 0) return GPT_OK;
    GPT_ON_DEVICE(s->device);
    unsigned char* base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_noalloc_mu);
        if (!g_noalloc_ptr[s->device]) {
            // once per device, with that device current: where THIS GPU keeps its copy of the array
            void* p = nullptr;
            HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_noalloc_ptr[s->device] = (unsigned char*)p;
        }
        base = g_noalloc_ptr[s->device];
    }
    float4* ray_o = (float4*)(base + GPT_NOALLOC_OFF_RAY_O);
    float4* ray_d = (float4*)(base + GPT_NOALLOC_OFF_RAY_D);
    float2* hit = (float2*)(base + GPT_NOALLOC_OFF_HIT);
    unsigned* counts = (unsigned*)(base + GPT_NOALLOC_OFF_COUNTERS);
    unsigned long long* stats = (unsigned long long*)(base + GPT_NOALLOC_OFF_STATS);
    uint32_t* spill = (uint32_t*)(base + GPT_NOALLOC_OFF_SPILL);
    // the array keeps the previous call's contents: queue heads, live counts and statistics start at zero
    HIP_TRY(hipMemsetAsync(counts, 0, GPT_NOALLOC_COUNTER_BYTES + GPT_NOALLOC_STAT_BYTES, s->stream));
    const int grid = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_noalloc_pack, dim3(grid), dim3(kBlock), 0, s->stream, d_rays, n, ray_o, ray_d, hit, counts);
    if (s->geom.n_tris > 0)
        gptk_intersect(s->stream, grid, &s->geom, use_kdtree ? 1 : 0, 0, 0, ray_o, ray_d, hit, counts, counts + GPT_MAX_DEPTH_SLOTS,
                       stats, spill, (size_t)grid * kBlock, s->refill_below, s->leaf_min);
    hipLaunchKernelGGL(k_noalloc_unpack, dim3(grid), dim3(kBlock), 0, s->stream, hit, n, d_tri, d_t);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return GPT_OK;
}

// ---- kd-tree builder through the C-ABI ----------------------------------------
struct gpt_kdtree { gptkd::Tree tree; };

gpt_status gpt_kdtree_build(const float* tri_vertices, int n_tris, int small_threshold, gpt_kdtree** out_tree) {
    if (!out_tree || n_tris < 0 || (n_tris > 0 && !tri_vertices)) return fail(GPT_ERR_INVALID_ARGUMENT, "bad argument");
    gpt_kdtree* t = new gpt_kdtree();
    gptkd::BuildParams bp;
    if (small_threshold > 0) bp.small_threshold = small_threshold;
    gptkd::build(tri_vertices, n_tris, bp, t->tree);
    *out_tree = t;
    return GPT_OK;
}
gpt_status gpt_kdtree_build_gpu(const float* tri_vertices, int n_tris, int small_threshold, int device, gpt_kdtree** out_tree) {
    if (!out_tree || n_tris < 0 || (n_tris > 0 && !tri_vertices)) return fail(GPT_ERR_INVALID_ARGUMENT, "bad argument");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return fail(GPT_ERR_NO_DEVICE, "no HIP device");
    GPT_ON_DEVICE(device);
    gpt_kdtree* t = new gpt_kdtree();
    gptkd::BuildParams bp;
    if (small_threshold > 0) bp.small_threshold = small_threshold;
    std::string err;
    if (!gptkd::build_gpu(tri_vertices, n_tris, bp, t->tree, &err)) { delete t; return fail(GPT_ERR_HIP, err); }
    *out_tree = t;
    return GPT_OK;
}
gpt_status gpt_kdtree_info(const gpt_kdtree* t, int* n_nodes, int* n_refs, int* max_depth, float bounds[6]) {
    if (!t) return fail(GPT_ERR_INVALID_ARGUMENT, "null tree");
    if (n_nodes) *n_nodes = (int)t->tree.nodes.size();
    if (n_refs) *n_refs = (int)t->tree.refs.size();
    if (max_depth) *max_depth = t->tree.max_depth;
    if (bounds) memcpy(bounds, t->tree.bounds, 24);
    return GPT_OK;
}
gpt_status gpt_kdtree_copy(const gpt_kdtree* t, uint32_t* nodes, int32_t* refs) {
    if (!t) return fail(GPT_ERR_INVALID_ARGUMENT, "null tree");
    if (nodes) memcpy(nodes, t->tree.nodes.data(), t->tree.nodes.size() * 8);
    if (refs) memcpy(refs, t->tree.refs.data(), t->tree.refs.size() * 4);
    return GPT_OK;
}
void gpt_kdtree_free(gpt_kdtree* t) { delete t; }

// ---- device memory helpers ---------------------------------------------------
gpt_status gpt_device_alloc(void** out_ptr, size_t bytes) {
    if (!out_ptr) return fail(GPT_ERR_INVALID_ARGUMENT, "null output");
    HIP_TRY(hipMalloc(out_ptr, bytes > 0 ? bytes : 16));
    return GPT_OK;
}
gpt_status gpt_device_upload(void* dst_device, const void* src_host, size_t bytes) {
    if (bytes == 0) return GPT_OK;
    HIP_TRY(hipMemcpy(dst_device, src_host, bytes, hipMemcpyHostToDevice));
    return GPT_OK;
}
gpt_status gpt_device_download(void* dst_host, const void* src_device, size_t bytes) {
    if (bytes == 0) return GPT_OK;
    HIP_TRY(hipMemcpy(dst_host, src_device, bytes, hipMemcpyDeviceToHost));
    return GPT_OK;
}
gpt_status gpt_device_free(void* ptr) {
    if (ptr) HIP_TRY(hipFree(ptr));
    return GPT_OK;
}

// ---- launchKernel ---------------------------------------------------------------
namespace {
// Passes traced ahead of the caller, pass after pass as packed float3 (what a call hands over).
struct LookAhead {
    float* d_img3 = nullptr;      // count x numPixels x 3 floats
    Camera cam;
    int begin = 0, count = 0;     // counters [begin, begin + count); count == 0: empty
    hipEvent_t done = nullptr;    // recorded behind the batch on the scene's stream
};
// One pass on its way to the host: pinned staging buffer + the event behind its copy.
struct Staged {
    float* host = nullptr;        // pinned, numPixels x 3 floats
    hipEvent_t done = nullptr;
    Camera cam;
    int counter = 0;
    int src_slot = -1;            // the look-ahead buffer the copy reads (until `done`)
    bool valid = false;
};
struct LaunchCache {
    std::mutex mu;
    gpt_scene* scene = nullptr;
    const void* polys = nullptr; const void* spheres = nullptr;
    int numPolys = -1, numSpheres = -1;
    unsigned long long checksum = 0;
    gpt_render_params params;
    bool params_set = false;
    bool assume_static = false;
    int device = -1;
    hipStream_t copy = nullptr;               // device-to-host copies (and the checksum of a first call), beside the scene's tracing stream
    hipStream_t check = nullptr;              // the per-call checksum of a known scene: its own stream, never behind a copy
    unsigned long long* d_sum = nullptr;
    unsigned long long* h_sum = nullptr;      // pinned
    LookAhead slot[2];
    Staged staged[2];
    size_t slot_floats = 0;
    int last_counter = 0, streak = 0;
    Camera last_cam;
    bool have_last = false;
} g_launch;

bool same_camera(const Camera& a, const Camera& b) {
    return a.resolution.x == b.resolution.x && a.resolution.y == b.resolution.y && a.position.x == b.position.x &&
           a.position.y == b.position.y && a.position.z == b.position.z && a.view.x == b.view.x && a.view.y == b.view.y &&
           a.view.z == b.view.z && a.up.x == b.up.x && a.up.y == b.up.y && a.up.z == b.up.z && a.fov.x == b.fov.x &&
           a.fov.y == b.fov.y && a.apertureRadius == b.apertureRadius && a.focalDistance == b.focalDistance;
}

[[noreturn]] void die(const char* what) {
    fprintf(stderr, "%s %s\n", what, g_error.c_str());
    exit(EXIT_FAILURE);
}
void hip_or_die(hipError_t e, const char* what) {
    if (e != hipSuccess) { g_error = std::string(what) + ": " + hipGetErrorString(e); die("launchKernel:"); }
}

void drop_lookahead(LaunchCache& L, bool free_buffers) {
    if (L.copy) (void)hipStreamSynchronize(L.copy);
    for (Staged& st : L.staged) {
        st.valid = false; st.src_slot = -1;
        if (free_buffers && st.host) { (void)hipHostFree(st.host); st.host = nullptr; }
    }
    for (LookAhead& a : L.slot) {
        a.count = 0;
        if (free_buffers && a.d_img3) { (void)hipFree(a.d_img3); a.d_img3 = nullptr; }
    }
    if (free_buffers) L.slot_floats = 0;
}

// The side stream, checksum words and events of the device the cache was set up on (none: a no-op).
void release_side_resources(LaunchCache& L) {
    if (L.device < 0) return;
    DeviceGuard guard(L.device);
    if (L.copy) { (void)hipStreamSynchronize(L.copy); (void)hipStreamDestroy(L.copy); L.copy = nullptr; }
    if (L.check) { (void)hipStreamSynchronize(L.check); (void)hipStreamDestroy(L.check); L.check = nullptr; }
    if (L.d_sum) { (void)hipFree(L.d_sum); L.d_sum = nullptr; }
    if (L.h_sum) { (void)hipHostFree(L.h_sum); L.h_sum = nullptr; }
    for (LookAhead& a : L.slot) if (a.done) { (void)hipEventDestroy(a.done); a.done = nullptr; }
    for (Staged& st : L.staged) if (st.done) { (void)hipEventDestroy(st.done); st.done = nullptr; }
    L.device = -1;
}
}  // namespace

void gpt_launch_kernel_params(const gpt_render_params* params) {
    std::lock_guard<std::mutex> lock(g_launch.mu);
    if (params) { g_launch.params = *params; g_launch.params_set = true; }
    else g_launch.params_set = false;
    if (g_launch.scene) {
        DeviceGuard guard(g_launch.scene->device);
        (void)hipStreamSynchronize(g_launch.scene->stream);
        drop_lookahead(g_launch, true);
        gpt_scene_destroy(g_launch.scene);
        g_launch.scene = nullptr;
    }
}

void gpt_shutdown_launch_cache_() {
    std::lock_guard<std::mutex> lock(g_launch.mu);
    LaunchCache& L = g_launch;
    if (L.scene) {
        DeviceGuard guard(L.scene->device);
        (void)hipStreamSynchronize(L.scene->stream);
        drop_lookahead(L, true);
        gpt_scene_destroy(L.scene);
        L.scene = nullptr;
    } else if (L.device >= 0) {
        DeviceGuard guard(L.device);
        drop_lookahead(L, true);
    }
    release_side_resources(L);
    L.have_last = false;
}

void gpt_launch_kernel_static_scene(int is_static) {
    std::lock_guard<std::mutex> lock(g_launch.mu);
    g_launch.assume_static = is_static != 0;
}

}  // extern "C"

namespace {
// One call.  `verify_late`: the checksum of an already known scene is started at once but only waited for AFTER the
// pass has been copied to the caller -- the answer is prepared on the assumption that the arrays are unchanged, under
// the check's latency -- and the function returns false when the arrays turn out to have changed (the caller then
// repeats the call with verify_late = false, which finds the change before it does anything else).
bool launch_kernel_once(int numPolys, Poly* polys, int numSpheres, Sphere* spheres, int numPixels, Color* pixels,
                        int counter, const Camera& renderCamera, bool verify_late) {
    LaunchCache& L = g_launch;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { g_error = "no HIP device"; die("launchKernel:"); }
    if (L.device != dev) {
        // first call (or the caller moved to another GPU): the side stream and the checksum words live on `dev`
        if (L.scene) {
            DeviceGuard guard(L.scene->device);
            (void)hipStreamSynchronize(L.scene->stream);
            drop_lookahead(L, true);
            gpt_scene_destroy(L.scene);
            L.scene = nullptr;
        }
        release_side_resources(L);     // (those of the GPU the caller was on before)
        hip_or_die(hipStreamCreateWithFlags(&L.copy, hipStreamNonBlocking), "stream");
        {   // (the most urgent priority the device offers: the check is 50 us of work that a whole call waits for)
            int least = 0, greatest = 0;
            hip_or_die(hipDeviceGetStreamPriorityRange(&least, &greatest), "stream priorities");
            hip_or_die(hipStreamCreateWithPriority(&L.check, hipStreamNonBlocking, greatest), "stream");
        }
        hip_or_die(hipMalloc((void**)&L.d_sum, 8), "hipMalloc");
        hip_or_die(hipHostMalloc((void**)&L.h_sum, 8, hipHostMallocDefault), "hipHostMalloc");
        for (LookAhead& a : L.slot) hip_or_die(hipEventCreateWithFlags(&a.done, hipEventDisableTiming), "event");
        for (Staged& st : L.staged) hip_or_die(hipEventCreateWithFlags(&st.done, hipEventDisableTiming), "event");
        L.device = dev;
    }
    const bool same_arrays = L.scene && L.polys == polys && L.spheres == spheres && L.numPolys == numPolys && L.numSpheres == numSpheres;
    const bool continues = L.have_last && same_camera(L.last_cam, renderCamera) && counter == L.last_counter + 1;
    // Content checksum of the two device arrays: the scene is re-indexed only when it changes.  The reference reads
    // the arrays on every call (path_tracer_kernel.cu:805), so a caller may edit them in place between two calls: the
    // checksum -- one pass over the arrays on a side stream (the caller's own streams are not touched) and one
    // wait -- runs on EVERY call, unless the caller has declared the scene static.  A mismatch drops every pass traced
    // ahead from the old contents.  For a scene already known the wait comes last (verify_late), so that the check runs
    // under the copy of the pass to the caller instead of in front of it.
    bool same = same_arrays;
    bool check_pending = false;
    if (!same_arrays || !L.assume_static) {
        hipStream_t cs = same_arrays ? L.check : L.copy;
        hip_or_die(hipMemsetAsync(L.d_sum, 0, 8, cs), "hipMemset");
        if (numPolys > 0) gptk_checksum(cs, (const uint32_t*)polys, (size_t)numPolys * sizeof(Poly) / 4, L.d_sum);
        if (numSpheres > 0) gptk_checksum(cs, (const uint32_t*)spheres, (size_t)numSpheres * sizeof(Sphere) / 4, L.d_sum);
        hip_or_die(hipMemcpyAsync(L.h_sum, L.d_sum, 8, hipMemcpyDeviceToHost, cs), "checksum readback");
        if (same_arrays && verify_late) {
            check_pending = true;          // waited for below, after the pass has been copied out
        } else {
            hip_or_die(hipStreamSynchronize(cs), "checksum");
            same = same_arrays && L.checksum == *L.h_sum;
        }
    }
    if (!same) {
        const unsigned long long sum = *L.h_sum;
        if (L.scene) {
            DeviceGuard guard(L.scene->device);
            (void)hipStreamSynchronize(L.scene->stream);
            drop_lookahead(L, true);
            gpt_scene_destroy(L.scene);
            L.scene = nullptr;
        }
        std::vector<Poly> hp((size_t)std::max(numPolys, 0));
        std::vector<Sphere> hs((size_t)std::max(numSpheres, 0));
        if (numPolys > 0) hip_or_die(hipMemcpy(hp.data(), polys, hp.size() * sizeof(Poly), hipMemcpyDeviceToHost), "copying polys from the device");
        if (numSpheres > 0) hip_or_die(hipMemcpy(hs.data(), spheres, hs.size() * sizeof(Sphere), hipMemcpyDeviceToHost), "copying spheres from the device");
        gpt_render_params rp;
        if (L.params_set) rp = L.params;
        else { gpt_render_params_default(&rp); rp.row_stride_is_width = 0; }   // the reference's arithmetic
        // room for the look-ahead below (1 switches it off)
        if (!L.params_set || rp.passes_in_flight <= 0) rp.passes_in_flight = 32;
        if (gpt_scene_create(hs.data(), numSpheres, hp.data(), numPolys, &rp, dev, &L.scene) != GPT_OK)
            die("launchKernel: scene upload failed:");
        L.polys = polys; L.spheres = spheres; L.numPolys = numPolys; L.numSpheres = numSpheres; L.checksum = sum;
        L.have_last = false;
    }
    gpt_scene* s = L.scene;
    // The batches traced ahead fill the GPU with workgroups that stay until their launch is done (eight per CU take
    // every wave slot), and the per-call checksum, a kernel on another stream, would now and then wait for a whole
    // launch to end -- with the caller waiting behind it (every 20th call took 2.3 ms instead of 0.9).  Six workgroups
    // per CU trace 9 % slower and leave the checksum its slots (mean per call 1.03 ms instead of 1.19); a scene declared
    // static needs none.
    s->trace_grid = L.assume_static ? kTraceGrid : kTraceGrid / 8 * 6;
    if (const char* e = getenv("GPT_LAUNCH_GRID")) s->trace_grid = std::max(1, std::min(kTraceGrid, atoi(e)));
    DeviceGuard device_guard_(s->device);
    if (!device_guard_.ok()) { g_error = "hipSetDevice"; die("launchKernel:"); }
    if (s->numPixels != numPixels) {
        hip_or_die(hipStreamSynchronize(s->stream), "sync");
        drop_lookahead(L, true);
        if (gpt_scene_reset(s, numPixels) != GPT_OK) die("launchKernel: buffer allocation failed:");
    }
    if (check_render_args(s, &renderCamera) != GPT_OK) die("launchKernel:");
    const size_t pass_floats = (size_t)numPixels * 3;
    if (L.slot_floats != pass_floats * (size_t)s->batch) {
        drop_lookahead(L, true);
        for (LookAhead& a : L.slot) hip_or_die(hipMalloc((void**)&a.d_img3, pass_floats * (size_t)s->batch * sizeof(float)), "look-ahead buffer");
        for (Staged& st : L.staged) hip_or_die(hipHostMalloc((void**)&st.host, pass_floats * sizeof(float), hipHostMallocDefault), "staging buffer");
        L.slot_floats = pass_floats * (size_t)s->batch;
    }
    // The reference's render loop asks for counter, counter + 1, ... with an unchanged camera until
    // the user moves it (path_tracer.cpp:94-108).  A pass is a pure function of (scene, camera,
    // counter), and tracing passes one at a time pays the tail of the longest paths at every depth,
    // so a run of such calls is answered from batches traced ahead -- 1, 2, 4, ... passes, at most the
    // scene's passes_in_flight -- into one of two buffers: while the caller takes the passes of one,
    // the next batch is already being traced into the other.  Any other call (camera moved, counter
    // jumped) starts over with a single pass.
    L.streak = continues ? L.streak + 1 : 0;
    auto holds = [&](const LookAhead& a, int c) { return a.count > 0 && same_camera(a.cam, renderCamera) && c >= a.begin && c - a.begin < a.count; };
    auto trace_into = [&](LookAhead& a, int begin, int P) {
        if (enqueue_batch(s, &renderCamera, begin, P, a.d_img3) != GPT_OK) die("launchKernel:");
        hip_or_die(hipEventRecord(a.done, s->stream), "event");
        a.cam = renderCamera; a.begin = begin; a.count = P;
    };
    // Before a buffer is traced into again, no copy may still be reading it.
    auto trace_over = [&](int k, int begin, int P) {
        for (Staged& st : L.staged)
            if (st.valid && st.src_slot == k) { hip_or_die(hipEventSynchronize(st.done), "copy"); st.src_slot = -1; }
        trace_into(L.slot[k], begin, P);
    };
    int cur = holds(L.slot[0], counter) ? 0 : (holds(L.slot[1], counter) ? 1 : -1);
    if (cur < 0) {
        int P = 1;
        for (int k = 0; k < L.streak && P < s->batch; k++) P *= 2;
        // (the buffer that does not hold the previous call's batch, so that a run's prefetched batch survives)
        cur = (L.have_last && holds(L.slot[0], L.last_counter)) ? 1 : 0;
        trace_over(cur, counter, std::min(P, s->batch));
    }
    // The pass travels device -> pinned staging buffer (a DMA on the side stream) -> caller's buffer (a host copy):
    // faster than a copy straight into pageable memory, and the DMA of the NEXT pass is started before this call
    // returns, so that a run of calls finds its pass already on the host.
    auto stage = [&](Staged& st, int k, int c) {
        const LookAhead& a = L.slot[k];
        hip_or_die(hipStreamWaitEvent(L.copy, a.done, 0), "wait");
        hip_or_die(hipMemcpyAsync(st.host, a.d_img3 + (size_t)(c - a.begin) * pass_floats, pass_floats * sizeof(float),
                                  hipMemcpyDeviceToHost, L.copy), "device-to-host copy");
        hip_or_die(hipEventRecord(st.done, L.copy), "event");
        st.valid = true; st.cam = renderCamera; st.counter = c; st.src_slot = k;
    };
    int have = -1;
    for (int k = 0; k < 2; k++)
        if (L.staged[k].valid && L.staged[k].counter == counter && same_camera(L.staged[k].cam, renderCamera)) have = k;
    if (have < 0) { have = 0; stage(L.staged[0], cur, counter); }
    // the batch after this one, once the run is established: traced while the caller consumes this one
    const int next_begin = L.slot[cur].begin + L.slot[cur].count;
    if (L.streak >= 2 && s->batch > 1 && !holds(L.slot[cur ^ 1], next_begin))
        trace_over(cur ^ 1, next_begin, std::min(s->batch, std::max(2 * L.slot[cur].count, 2)));
    // the next pass of the run starts its way to the host now
    if (L.streak >= 1) {
        const int nk = holds(L.slot[cur], counter + 1) ? cur : (holds(L.slot[cur ^ 1], counter + 1) ? (cur ^ 1) : -1);
        if (nk >= 0) stage(L.staged[have ^ 1], nk, counter + 1);
    }
    hip_or_die(hipEventSynchronize(L.staged[have].done), "device-to-host copy");
    memcpy(pixels, L.staged[have].host, pass_floats * sizeof(float));
    L.staged[have].valid = false;
    L.last_counter = counter;
    L.last_cam = renderCamera;
    L.have_last = true;
    if (check_pending) {
        hip_or_die(hipStreamSynchronize(L.check), "checksum");
        if (L.checksum != *L.h_sum) return false;      // the arrays were edited in place: what was handed out is stale
    }
    return true;
}
}  // namespace

extern "C" {

void launchKernel(int numPolys, Poly* polys, int numSpheres, Sphere* spheres, int numPixels, Color* pixels,
                  int counter, Camera renderCamera) {
    std::lock_guard<std::mutex> lock(g_launch.mu);
    if (!launch_kernel_once(numPolys, polys, numSpheres, spheres, numPixels, pixels, counter, renderCamera, true))
        launch_kernel_once(numPolys, polys, numSpheres, spheres, numPixels, pixels, counter, renderCamera, false);
}

}  // extern "C"
