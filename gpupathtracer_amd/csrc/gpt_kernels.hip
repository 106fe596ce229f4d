// gpt_kernels.hip -- gfx950 kernels of one sample-per-pixel pass.
//
// Wavefront formulation of the reference's launchKernel
// (path_tracer_kernel.cu:760-882):
//
//   k_camera        raycastFromCameraKernel + initializeThings (:86-102,125-191)
//   per depth:
//     k_intersect   closest hit of traceRayKernel (:495-557): ground, spheres,
//                   then triangles by kd-tree traversal or by the reference's
//                   brute-force loop
//     k_shade       the rest of traceRayKernel (:560-716) fused with the stream
//                   compaction that replaces thrust::remove_if (:811-813): the
//                   surviving paths are written, whole state, to the next
//                   depth's queue at positions handed out by a wave64 ballot +
//                   one atomic per wave
//   k_finish        image += pass (PathTracer::render, path_tracer.cpp:100-103)
//
// Path state lives in compacted, structure-of-arrays queues (three 16-byte
// words per path), double-buffered by depth parity; nothing returns to the host
// between depths: the live count of depth d is a device counter that the
// kernels of depth d read.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gpt_device_structs.h"
#include "kd_traverse.h"
#include "pt_device.h"

using namespace pt;

#define GPT_BLOCK 256

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned lane_id() {
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` below this lane
__device__ __forceinline__ unsigned prefix_count(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
// The wave takes the next `chunk` items of a global work queue.
__device__ __forceinline__ unsigned wave_fetch(unsigned* head, unsigned chunk) {
    unsigned base = 0;
    if (lane_id() == 0) base = atomicAdd(head, chunk);
    return (unsigned)__builtin_amdgcn_readfirstlane((int)base);
}

__device__ __forceinline__ Mat load_material(const float4* __restrict__ mats, int index) {
    const float4 m0 = mats[4 * index + 0];
    const float4 m1 = mats[4 * index + 1];
    const float4 m2 = mats[4 * index + 2];
    const float4 m3 = mats[4 * index + 3];
    Mat m;
    m.diffuse = v3(m0.x, m0.y, m0.z);
    m.ior = m0.w;
    m.specular = v3(m1.x, m1.y, m1.z);
    m.transmits = (m1.w != 0.0f) ? 1 : 0;
    m.emitted = v3(m2.x, m2.y, m2.z);
    m.sigma_s = m2.w;
    m.sigma_a = v3(m3.x, m3.y, m3.z);
    return m;
}

// ---------------------------------------------------------------------------
// k_camera: one primary ray per pixel; queue slot i holds pixel i.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(GPT_BLOCK) void k_camera(CamPre cam, int numPixels, uint32_t seed,
                                                      float4* __restrict__ ray_o,
                                                      float4* __restrict__ ray_d,
                                                      float4* __restrict__ thr,
                                                      float4* __restrict__ pass_rgb,
                                                      unsigned* __restrict__ counts) {
    const int stride = gridDim.x * GPT_BLOCK;
    for (int i = blockIdx.x * GPT_BLOCK + threadIdx.x; i < numPixels; i += stride) {
        V3 o, d;
        camera_ray(cam, i, seed, o, d);
        ray_o[i] = make_float4(o.x, o.y, o.z, __int_as_float(i));
        ray_d[i] = make_float4(d.x, d.y, d.z, __int_as_float(-1));   // medium: air
        thr[i] = make_float4(1.0f, 1.0f, 1.0f, 0.0f);
        pass_rgb[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = (unsigned)numPixels;
}

// ---------------------------------------------------------------------------
// Traversal stack: the top KD_LDS_N entries of every lane live in LDS
// ([entry][thread] so a wave's accesses never conflict), older entries spill to
// a per-thread column of a global buffer and come back when the LDS part runs
// empty.  Depth never exceeds the tree's depth, which sizes the spill buffer.
// ---------------------------------------------------------------------------
#define KD_LDS_N 8

struct DevStack {
    uint32_t* lds;          // this thread's column: lds[(3*e + c) * GPT_BLOCK]
    uint32_t* spill;        // this thread's column: spill[(3*e + c) * spill_stride]
    size_t spill_stride;
    int sp, lo;             // entries [lo, sp) are in LDS, [0, lo) in the spill buffer

    __device__ __forceinline__ void reset() { sp = 0; lo = 0; }
    __device__ __forceinline__ void push(uint32_t node, float tmin, float tmax) {
        if (sp - lo == KD_LDS_N) {
            const int slot = lo & (KD_LDS_N - 1);
            spill[(size_t)(3 * lo + 0) * spill_stride] = lds[(3 * slot + 0) * GPT_BLOCK];
            spill[(size_t)(3 * lo + 1) * spill_stride] = lds[(3 * slot + 1) * GPT_BLOCK];
            spill[(size_t)(3 * lo + 2) * spill_stride] = lds[(3 * slot + 2) * GPT_BLOCK];
            lo++;
        }
        const int slot = sp & (KD_LDS_N - 1);
        lds[(3 * slot + 0) * GPT_BLOCK] = node;
        lds[(3 * slot + 1) * GPT_BLOCK] = __float_as_uint(tmin);
        lds[(3 * slot + 2) * GPT_BLOCK] = __float_as_uint(tmax);
        sp++;
    }
    __device__ __forceinline__ bool pop(uint32_t& node, float& tmin, float& tmax) {
        if (sp == 0) return false;
        sp--;
        if (sp < lo) {
            node = spill[(size_t)(3 * sp + 0) * spill_stride];
            tmin = __uint_as_float(spill[(size_t)(3 * sp + 1) * spill_stride]);
            tmax = __uint_as_float(spill[(size_t)(3 * sp + 2) * spill_stride]);
            lo = sp;
        } else {
            const int slot = sp & (KD_LDS_N - 1);
            node = lds[(3 * slot + 0) * GPT_BLOCK];
            tmin = __uint_as_float(lds[(3 * slot + 1) * GPT_BLOCK]);
            tmax = __uint_as_float(lds[(3 * slot + 2) * GPT_BLOCK]);
        }
        return true;
    }
};

// ---------------------------------------------------------------------------
// k_intersect: closest hit for every live path of this depth.
// Waves pull 64 queue slots at a time from a global work counter.
// ---------------------------------------------------------------------------
template <bool USE_KD, bool STATS>
__global__ __launch_bounds__(GPT_BLOCK) void k_intersect(DevGeometry g, int depth,
                                                         const float4* __restrict__ ray_o,
                                                         const float4* __restrict__ ray_d,
                                                         float2* __restrict__ hit,
                                                         unsigned* __restrict__ counts,
                                                         unsigned* __restrict__ work,
                                                         unsigned long long* __restrict__ stats,
                                                         uint32_t* __restrict__ spill, size_t spill_stride) {
    __shared__ uint32_t s_stack[3 * KD_LDS_N * GPT_BLOCK];
    const unsigned n = counts[depth];
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[GPT_STAT_LIVE0 + depth] += n;

    DevStack st;
    st.lds = s_stack + threadIdx.x;
    st.spill = spill + ((size_t)blockIdx.x * GPT_BLOCK + threadIdx.x);
    st.spill_stride = spill_stride;
    st.reset();

    uint32_t cNodes = 0, cTris = 0;
    const unsigned lane = lane_id();
    for (;;) {
        const unsigned base = wave_fetch(&work[depth], 64u);
        if (base >= n) break;
        const unsigned i = base + lane;
        if (i < n) {
            const float4 ro = ray_o[i];
            const float4 rd = ray_d[i];
            const V3 o = v3(ro.x, ro.y, ro.z);
            const V3 d = v3(rd.x, rd.y, rd.z);

            float bestT = PT_T_START;
            int prim = PT_PRIM_MISS;
            if (g.ground_enabled) {
                const float t = ground_t(g.ground_elevation, o, d);
                if (t > 0) { bestT = t; prim = PT_PRIM_GROUND; }
            }
            for (int s = 0; s < g.n_spheres; s++) {
                const float4 sp = g.spheres[s];
                const float t = sphere_t(v3(sp.x, sp.y, sp.z), sp.w, o, d);
                if (t > 0 && t < bestT) { bestT = t; prim = PT_PRIM_SPHERE | s; }
            }
            if (g.n_tris > 0) {
                float bt = bestT;
                int bi = -1;
                if (USE_KD) kd_closest_hit<DevStack, STATS>(g.kd, o, d, bt, bi, st, cNodes, cTris);
                else brute_closest_hit<STATS>(g.kd.tris, g.n_tris, o, d, bt, bi, cTris);
                if (bi >= 0) { bestT = bt; prim = bi; }
            }
            hit[i] = make_float2(bestT, __int_as_float(prim));
        }
    }
    if (STATS) {
        atomicAdd(&stats[GPT_STAT_KD_NODES], (unsigned long long)cNodes);
        atomicAdd(&stats[GPT_STAT_KD_TRIS], (unsigned long long)cTris);
    }
}

// ---------------------------------------------------------------------------
// k_shade: medium + BSDF for every live path, compaction into the next queue.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(GPT_BLOCK) void k_shade(DevGeometry g, ShadeParams sp, int depth, uint32_t seed,
                                                     const float4* __restrict__ in_o,
                                                     const float4* __restrict__ in_d,
                                                     const float4* __restrict__ in_thr,
                                                     const float2* __restrict__ hit,
                                                     float4* __restrict__ out_o,
                                                     float4* __restrict__ out_d,
                                                     float4* __restrict__ out_thr,
                                                     float4* __restrict__ pass_rgb,
                                                     unsigned* __restrict__ counts,
                                                     unsigned* __restrict__ work) {
    const unsigned n = counts[depth];
    const unsigned lane = lane_id();
    unsigned* shade_head = work + GPT_MAX_DEPTH_SLOTS + depth;
    for (;;) {
        const unsigned base = wave_fetch(shade_head, 64u);
        if (base >= n) break;
        const unsigned i = base + lane;
        bool alive = false;
        V3 o = v3(0, 0, 0), d = v3(0, 0, 0), thr = v3(0, 0, 0);
        int pixel = 0, medium = -1;
        if (i < n) {
            const float4 ro = in_o[i];
            const float4 rd = in_d[i];
            const float4 th = in_thr[i];
            const float2 h = hit[i];
            o = v3(ro.x, ro.y, ro.z);
            d = v3(rd.x, rd.y, rd.z);
            thr = v3(th.x, th.y, th.z);
            pixel = __float_as_int(ro.w);
            medium = __float_as_int(rd.w);
            const float bestT = h.x;
            const int prim = __float_as_int(h.y);

            V3 nrm = v3(0, 1, 0);
            Mat m = ground_material();
            int matIndex = -1;
            if (prim >= 0) {
                if (prim & PT_PRIM_SPHERE) {
                    const int s = prim & ~PT_PRIM_SPHERE;
                    const float4 sph = g.spheres[s];
                    nrm = sphere_normal(v3(sph.x, sph.y, sph.z), o, d, bestT);
                    matIndex = g.sphere_mat[s];
                } else {
                    const float4 nm = g.tri_nm[prim];
                    nrm = v3(nm.x, nm.y, nm.z);
                    matIndex = __float_as_int(nm.w);
                }
                m = load_material(g.mats, matIndex);
            }
            V3 sa = v3(0, 0, 0);
            float ss = 0.0f;
            if (medium >= 0) {
                const float4 m2 = g.mats[4 * medium + 2];
                const float4 m3 = g.mats[4 * medium + 3];
                ss = m2.w;
                sa = v3(m3.x, m3.y, m3.z);
            }
            V3 add;
            bool has_add;
            alive = shade_bounce(o, d, thr, medium, sa, ss, prim != PT_PRIM_MISS, bestT, nrm, m, matIndex,
                                 pixel, depth, seed, sp, add, has_add);
            if (has_add) {
                float4 acc = pass_rgb[pixel];
                acc.x += add.x; acc.y += add.y; acc.z += add.z;
                pass_rgb[pixel] = acc;
            }
        }
        // ---- wave64 compaction: ballot, prefix count, one atomic per wave ----
        const unsigned long long mask = __ballot(alive);
        if (mask != 0ull) {
            unsigned wbase = 0;
            if (lane == 0) wbase = atomicAdd(&counts[depth + 1], (unsigned)__popcll(mask));
            wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)wbase);
            if (alive) {
                const unsigned dst = wbase + prefix_count(mask);
                out_o[dst] = make_float4(o.x, o.y, o.z, __int_as_float(pixel));
                out_d[dst] = make_float4(d.x, d.y, d.z, __int_as_float(medium));
                out_thr[dst] = make_float4(thr.x, thr.y, thr.z, 0.0f);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// k_finish: image += pass; optionally the pass alone as packed float3 for the
// host (the reference's D2H copy, path_tracer_kernel.cu:869).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(GPT_BLOCK) void k_finish(int numPixels, const float4* __restrict__ pass_rgb,
                                                      float4* __restrict__ image, float* __restrict__ pass_out3) {
    const int stride = gridDim.x * GPT_BLOCK;
    for (int i = blockIdx.x * GPT_BLOCK + threadIdx.x; i < numPixels; i += stride) {
        const float4 p = pass_rgb[i];
        float4 a = image[i];
        a.x += p.x; a.y += p.y; a.z += p.z; a.w += 1.0f;
        image[i] = a;
        if (pass_out3) {
            pass_out3[3 * (size_t)i + 0] = p.x;
            pass_out3[3 * (size_t)i + 1] = p.y;
            pass_out3[3 * (size_t)i + 2] = p.z;
        }
    }
}

// ---------------------------------------------------------------------------
// Debug / parity kernels
// ---------------------------------------------------------------------------
// Closest triangle for caller-supplied rays (no ground, no spheres).
template <bool USE_KD>
__global__ __launch_bounds__(GPT_BLOCK) void k_debug_intersect(DevGeometry g, const float* __restrict__ rays, int n,
                                                               int32_t* __restrict__ tri, float* __restrict__ t_out,
                                                               uint32_t* __restrict__ spill, size_t spill_stride) {
    __shared__ uint32_t s_stack[3 * KD_LDS_N * GPT_BLOCK];
    DevStack st;
    st.lds = s_stack + threadIdx.x;
    st.spill = spill + ((size_t)blockIdx.x * GPT_BLOCK + threadIdx.x);
    st.spill_stride = spill_stride;
    st.reset();
    const int stride = gridDim.x * GPT_BLOCK;
    uint32_t c0 = 0, c1 = 0;
    for (int i = blockIdx.x * GPT_BLOCK + threadIdx.x; i < n; i += stride) {
        const V3 o = v3(rays[6 * (size_t)i], rays[6 * (size_t)i + 1], rays[6 * (size_t)i + 2]);
        const V3 d = v3(rays[6 * (size_t)i + 3], rays[6 * (size_t)i + 4], rays[6 * (size_t)i + 5]);
        float bt = PT_T_START;
        int bi = -1;
        if (g.n_tris > 0) {
            if (USE_KD) kd_closest_hit<DevStack, false>(g.kd, o, d, bt, bi, st, c0, c1);
            else brute_closest_hit<false>(g.kd.tris, g.n_tris, o, d, bt, bi, c1);
        }
        tri[i] = bi;
        t_out[i] = bt;
    }
}

// Order-independent checksum of a device buffer (launchKernel's scene cache key).
__global__ __launch_bounds__(GPT_BLOCK) void k_checksum(const uint32_t* __restrict__ words, size_t n,
                                                        unsigned long long* __restrict__ out) {
    unsigned long long acc = 0;
    const size_t stride = (size_t)gridDim.x * GPT_BLOCK;
    for (size_t i = (size_t)blockIdx.x * GPT_BLOCK + threadIdx.x; i < n; i += stride) {
        unsigned long long x = (unsigned long long)words[i] + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
        acc += x;
    }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off);
    if (lane_id() == 0) atomicAdd(out, acc);
}

// ---------------------------------------------------------------------------
// Launchers (called from gpt_api.cpp)
// ---------------------------------------------------------------------------
extern "C" {

void gptk_camera(hipStream_t s, int grid, const CamPre* cam, int numPixels, uint32_t seed, float4* ray_o,
                 float4* ray_d, float4* thr, float4* pass_rgb, unsigned* counts) {
    hipLaunchKernelGGL(k_camera, dim3(grid), dim3(GPT_BLOCK), 0, s, *cam, numPixels, seed, ray_o, ray_d, thr,
                       pass_rgb, counts);
}

void gptk_intersect(hipStream_t s, int grid, const DevGeometry* g, int use_kd, int stats_on, int depth,
                    const float4* ray_o, const float4* ray_d, float2* hit, unsigned* counts, unsigned* work,
                    unsigned long long* stats, uint32_t* spill, size_t spill_stride) {
#define GPT_LAUNCH_INTERSECT(KD, ST)                                                                        \
    hipLaunchKernelGGL((k_intersect<KD, ST>), dim3(grid), dim3(GPT_BLOCK), 0, s, *g, depth, ray_o, ray_d,   \
                       hit, counts, work, stats, spill, spill_stride)
    if (use_kd) { if (stats_on) GPT_LAUNCH_INTERSECT(true, true); else GPT_LAUNCH_INTERSECT(true, false); }
    else { if (stats_on) GPT_LAUNCH_INTERSECT(false, true); else GPT_LAUNCH_INTERSECT(false, false); }
#undef GPT_LAUNCH_INTERSECT
}

void gptk_shade(hipStream_t s, int grid, const DevGeometry* g, const ShadeParams* sp, int depth, uint32_t seed,
                const float4* in_o, const float4* in_d, const float4* in_thr, const float2* hit, float4* out_o,
                float4* out_d, float4* out_thr, float4* pass_rgb, unsigned* counts, unsigned* work) {
    hipLaunchKernelGGL(k_shade, dim3(grid), dim3(GPT_BLOCK), 0, s, *g, *sp, depth, seed, in_o, in_d, in_thr, hit,
                       out_o, out_d, out_thr, pass_rgb, counts, work);
}

void gptk_finish(hipStream_t s, int grid, int numPixels, const float4* pass_rgb, float4* image, float* pass_out3) {
    hipLaunchKernelGGL(k_finish, dim3(grid), dim3(GPT_BLOCK), 0, s, numPixels, pass_rgb, image, pass_out3);
}

void gptk_debug_intersect(hipStream_t s, int grid, const DevGeometry* g, int use_kd, const float* rays, int n,
                          int32_t* tri, float* t_out, uint32_t* spill, size_t spill_stride) {
    if (use_kd)
        hipLaunchKernelGGL((k_debug_intersect<true>), dim3(grid), dim3(GPT_BLOCK), 0, s, *g, rays, n, tri, t_out,
                           spill, spill_stride);
    else
        hipLaunchKernelGGL((k_debug_intersect<false>), dim3(grid), dim3(GPT_BLOCK), 0, s, *g, rays, n, tri, t_out,
                           spill, spill_stride);
}

void gptk_checksum(hipStream_t s, const uint32_t* words, size_t n, unsigned long long* out) {
    int grid = (int)((n + GPT_BLOCK - 1) / GPT_BLOCK);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_checksum, dim3(grid), dim3(GPT_BLOCK), 0, s, words, n, out);
}

}  // extern "C"
