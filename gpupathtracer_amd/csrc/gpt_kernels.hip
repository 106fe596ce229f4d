// gpt_kernels.hip -- gfx950 kernels of one sample-per-pixel pass.
//
// Wavefront formulation of the reference's launchKernel
// (path_tracer_kernel.cu:760-882):
//
//   k_camera        raycastFromCameraKernel + initializeThings (:86-102,125-191),
//                   and the new ray's closest ground / sphere hit (:509-533)
//   per depth:
//     k_intersect   the triangle part of traceRayKernel's closest hit
//                   (:536-557), by kd-tree traversal or by the reference's
//                   brute-force loop, starting from the ground / sphere hit
//                   that arrived with the ray
//     k_shade       the rest of traceRayKernel (:560-716) fused with the stream
//                   compaction that replaces thrust::remove_if (:811-813): the
//                   surviving paths are staged per block in LDS (wave64 ballot
//                   + prefix counts) and written, whole state plus the new
//                   ray's ground / sphere hit, to the next depth's queue, 512
//                   slots per atomic
//   k_finish        image += pass (PathTracer::render, path_tracer.cpp:100-103)
//
// Path state lives in compacted, structure-of-arrays queues (three 16-byte
// words and an 8-byte hit slot per path), double-buffered by depth parity;
// nothing returns to the host between depths: the live count of depth d is a
// device counter that the kernels of depth d read.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// wave-iteration counters of the instrumented traversal (first active lane counts)
__device__ unsigned long long g_wave_iters[4];
#if defined(__HIP_DEVICE_COMPILE__)
#define PT_COUNT_WAVE_ITERATION(slot)                                                          \
    do {                                                                                       \
        const unsigned long long m_ = __ballot(1);                                             \
        if (__builtin_amdgcn_mbcnt_hi((unsigned)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_, 0u)) == 0u) \
            atomicAdd(&g_wave_iters[slot], 1ull);                                              \
    } while (0)
#endif
#include "gpt_device_structs.h"
#include "kd_traverse.h"
#include "pt_device.h"
#include "pt_helper_eval.h"

using namespace pt;

#define GPT_BLOCK 256

// ---------------------------------------------------------------------------
// Streaming accesses: the path queues are written once and read once per depth, GBs per
// launch.  Marking them non-temporal keeps them from displacing the kd-tree and triangle
// records in L2 / Infinity Cache, which every ray re-reads.
// ---------------------------------------------------------------------------
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld_stream(const float4* p) {
    nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float2 ld_stream(const float2* p) {
    nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2*>(p));
    return make_float2(v.x, v.y);
}
__device__ __forceinline__ void st_stream(float4* p, float4 v) {
    nt_f4 w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<nt_f4*>(p));
}
__device__ __forceinline__ void st_stream(float2* p, float2 v) {
    nt_f2 w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<nt_f2*>(p));
}

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

// Inclusive wave64 scans in registers (DPP row shifts, then the two row broadcasts; every lane
// of the wave must be active).  Lanes without a source take the identity 0.
#define GPT_DPP_STEP(op, v, ctrl, row_mask) \
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, row_mask, 0xf, false))
__device__ __forceinline__ unsigned dpp_add(unsigned a, unsigned b) { return a + b; }
__device__ __forceinline__ unsigned dpp_max(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned dpp_or(unsigned a, unsigned b) { return a | b; }
// ---------------------------------------------------------------------------
// Closest hit among the ground plane and the spheres (traceRayKernel :509-533).
// The kernel that CREATES a ray (k_camera, k_shade) runs this for it, with all
// lanes busy, and stores the result in the ray's `hit` slot; the intersect
// kernels start from that slot and only have triangles left to test.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void analytic_hits(const DevGeometry& g, V3 o, V3 d, float& bestT_out, int& prim_out) {
    // (locals, not the caller's references: the loop bound and the running best stay in registers)
    float bestT = PT_T_START;
    int prim = PT_PRIM_MISS;
    if (g.ground_enabled) {
        const float t = ground_t(g.ground_elevation, o, d);
        if (t > 0) { bestT = t; prim = PT_PRIM_GROUND; }
    }
    // The sphere table is read through the constant address space: it is written before the launch and
    // never by a kernel, and only that tells the compiler that the stores around this loop cannot
    // change it -- the loads become scalar loads (one per wave, their own wait counter) instead of
    // vector loads whose wait also drains every store the lane has in flight.
    typedef float cf4 __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(4))) cf4* sph = (const __attribute__((address_space(4))) cf4*)(unsigned long long)g.spheres;
#ifdef ABL_NO_SPHERES   // experiment (wrong pictures): what do the sphere tests of new rays cost?
    const int n = 0;
#else
    const int n = g.n_spheres;
#endif
    for (int s = 0; s < n; s++) {
        const cf4 sp = sph[s];
        if (sphere_closer(v3(sp.x, sp.y, sp.z), sp.w /* radius^2 */, o, d, bestT)) prim = PT_PRIM_SPHERE | s;
    }
    bestT_out = bestT;
    prim_out = prim;
}

// ---------------------------------------------------------------------------
// k_camera: one primary ray per (pass, pixel).  A launch covers `numPasses`
// consecutive pass counters at once.  Queue order is pixel-major: slot i holds
// pass i % numPasses of pixel i / numPasses, so the lanes of a wave trace the
// samples of one or two pixels -- near-identical primary rays, and secondary
// rays that start from neighbouring surface points -- which keeps a wave's kd
// traversal in the same nodes and leaves.
// ---------------------------------------------------------------------------
// analytic_hits with the spheres named by `mask` only (bit s = sphere s; at most 32 spheres), in ascending order like the
// full loop.  `mask` must be wave-uniform: the loop then runs on the scalar unit and the sphere table is read by scalar loads.
__device__ __forceinline__ void analytic_hits_masked(const DevGeometry& g, V3 o, V3 d, uint32_t mask, float& bestT_out,
                                                     int& prim_out) {
    float bestT = PT_T_START;
    int prim = PT_PRIM_MISS;
    if (g.ground_enabled) {
        const float t = ground_t(g.ground_elevation, o, d);
        if (t > 0) { bestT = t; prim = PT_PRIM_GROUND; }
    }
    typedef float cf4 __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(4))) cf4* sph = (const __attribute__((address_space(4))) cf4*)(unsigned long long)g.spheres;
    for (uint32_t m = mask; m != 0u; m &= m - 1u) {
        const int s = __builtin_ctz(m);
        const cf4 sp = sph[s];
        if (sphere_closer(v3(sp.x, sp.y, sp.z), sp.w /* radius^2 */, o, d, bestT)) prim = PT_PRIM_SPHERE | s;
    }
    bestT_out = bestT;
    prim_out = prim;
}

// ---------------------------------------------------------------------------
// k_camera_masks: for every pixel, the spheres a primary ray of that pixel can possibly hit (bit s of masks[pixel]),
// once per batch -- the camera is the same for all its passes -- so that k_camera tests one or two spheres per ray
// instead of all of them.  Conservative: a sphere left out is one no ray of the pixel reaches, so
// the rays' hits are those of the full loop.
// The rays of a pixel: origin within `aperture` of the camera position, through a point of the focal plane within
// rho = focal * (|horizontal| / (resx - 1) + |vertical| / (resy - 1)) of the pixel centre's (the jitter is half a pixel
// either way).  At parameter s along the central ray (0: camera, 1: focal plane) such a ray is at most
// |1 - s| * aperture + s * rho away from it.  A ray that meets the sphere (centre C, radius R) does so at a point whose
// central-ray counterpart is within C's distance + R + that deviation of the camera, which bounds s; the sphere is kept
// when its centre is no further from the central LINE than R + the deviation at that s (1 % and a little added).
// R here is not the sphere's radius but the radius the reference's float arithmetic gives it: its test forms
// (v . d)^2 - (v . v - r^2) from terms of the size of dist^2, whose roundings add up to at most ~7e-7 dist^2 -- for a
// sphere far away that is more than r^2 itself, and rays passing the sphere at up to sqrt(r^2 + 7e-7 dist^2) from its
// centre can pass the test (measured at 3840 x 2160: 2e-7 dist^2; a sphere of a third of a pixel 50 units away is "hit"
// from five pixels away).  The masks must keep what the loop hits, so they use sqrt(r^2 + 2e-6 dist^2).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(GPT_BLOCK) void k_camera_masks(DevGeometry g, CamPre cam, int numPixels,
                                                            uint32_t* __restrict__ masks) {
    const int pixel = (int)(blockIdx.x * GPT_BLOCK + threadIdx.x);
    if (pixel >= numPixels) return;
    const int y = (int)(pixel / cam.stride);
    const int x = (int)(pixel - (y * cam.stride));
    const float sx = (float)x / cam.resx_m1, sy = (float)y / cam.resy_m1;
    const V3 onePlane = cam.middle + (((2 * sx) - 1) * cam.horizontal) + (((2 * sy) - 1) * cam.vertical);
    const V3 dirc = onePlane - cam.position;
    const float len1 = length(dirc);
    const float Lc = cam.focal * len1;                      // camera -> the pixel centre's point of the focal plane
    const float rho = 1.01f * fabsf(cam.focal) * (length(cam.horizontal) / cam.resx_m1 + length(cam.vertical) / cam.resy_m1);
    const float a = cam.lens ? 1.01f * cam.aperture : 0.0f;
    const float denom = Lc - rho - a;
    uint32_t mask = 0;
    const int n = g.n_spheres < 32 ? g.n_spheres : 32;
    for (int s = 0; s < n; s++) {
        const float4 sp = g.spheres[s];
        const V3 w = v3(sp.x, sp.y, sp.z) - cam.position;
        const float dist = length(w);
        const float dmax = dist + a;                            // no origin of the pixel's rays is further from the centre
        const float R = sqrtf(sp.w + 2.0e-6f * dmax * dmax);    // the radius of the reference's float test (see above)
        bool keep = true;
        if (denom > 0.05f * Lc && len1 > 0.0f) {
            const float smax = (dist + R + a) / denom;
            const float reach = 1.01f * (R + a + smax * (rho + a)) + 1e-4f * (dist + R);
            // distance of the centre from the central line, as |w x dirc| / |dirc|: the cross product is exact to a few
            // ulps of dist (dist^2 - proj^2 would cancel to ~1e-7 dist^2, more than a sub-pixel sphere's reach^2)
            const float perp = length(cross(w, dirc)) / len1;
            keep = !(perp > reach);                         // (a NaN keeps the sphere)
        }
        if (keep) mask |= 1u << s;
    }
    masks[pixel] = mask;
}

__global__ __launch_bounds__(GPT_BLOCK) void k_camera(DevGeometry g, CamPre cam, int numPixels, int numPasses,
                                                      uint32_t seed0,
                                                      float4* __restrict__ ray_o,
                                                      float4* __restrict__ ray_d,
                                                      float4* __restrict__ thr,
                                                      float2* __restrict__ hit,
                                                      float4* __restrict__ pass_rgb,
                                                      unsigned* __restrict__ counts,
                                                      const uint32_t* __restrict__ masks) {
    const unsigned total = (unsigned)numPixels * (unsigned)numPasses;
    const unsigned stride = gridDim.x * GPT_BLOCK;
    // (the loop is block-uniform: the union of a wave's sphere masks is formed with all its lanes)
    for (unsigned base = blockIdx.x * GPT_BLOCK; base < total; base += stride) {
        const unsigned i = base + threadIdx.x;
        const bool in = i < total;
        const int pixel = (int)((in ? i : total - 1u) / (unsigned)numPasses);
        const unsigned pass = (in ? i : total - 1u) - (unsigned)pixel * (unsigned)numPasses;
        V3 o, d;
        camera_ray(cam, pixel, seed0 + pass, o, d);
        float bestT;
        int prim;
        if (masks) {
            // the spheres any ray of this wave can reach: one or two pixels' worth with many passes per launch,
            // 64 neighbouring pixels' with one
            unsigned m = in ? masks[pixel] : 0u;
            GPT_DPP_STEP(dpp_or, m, 0x111, 0xf);
            GPT_DPP_STEP(dpp_or, m, 0x112, 0xf);
            GPT_DPP_STEP(dpp_or, m, 0x114, 0xf);
            GPT_DPP_STEP(dpp_or, m, 0x118, 0xf);
            GPT_DPP_STEP(dpp_or, m, 0x142, 0xa);
            GPT_DPP_STEP(dpp_or, m, 0x143, 0xc);
            analytic_hits_masked(g, o, d, (uint32_t)__builtin_amdgcn_readlane((int)m, 63), bestT, prim);
        } else {
            analytic_hits(g, o, d, bestT, prim);
        }
        if (in) {
            st_stream(&ray_o[i], make_float4(o.x, o.y, o.z, __int_as_float(pixel)));
            st_stream(&ray_d[i], make_float4(d.x, d.y, d.z, __int_as_float(-1)));   // medium: air
            // (no throughput is stored for a primary ray, and its pass radiance slot is not cleared here:
            // k_shade<true> handles every slot of the uncompacted queue exactly once and starts the slot with
            // what the first bounce adds, or zero)
            st_stream(&hit[i], make_float2(bestT, __int_as_float(prim)));
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[0] = total;
}

// ---------------------------------------------------------------------------
// Traversal stack: the top KD_LDS_N entries of every lane live in LDS
// ([word][thread] so a wave's accesses never conflict), older entries spill to
// a per-thread column of a global buffer and come back when the LDS part runs
// empty.  Depth never exceeds the tree's depth, which sizes the spill buffer.
// ---------------------------------------------------------------------------
#ifndef KD_LDS_N
#define KD_LDS_N 8
#endif
#ifndef KD_FETCH_CHUNK
#define KD_FETCH_CHUNK 512u   // rays a wave takes from the global queue per atomic
#endif
#ifndef KD_WAVES_PER_EU
#define KD_WAVES_PER_EU 7
#endif

// v_min / v_max as they are: the other operand for a NaN one, and none of the canonicalising extra instructions the
// compiler wraps around fminf / fmaxf.
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

struct DevStack {
    // An entry is two words.  Word 1 is tmin with its low 11 bits cleared (rounded DOWN by at most 2.4e-4 of
    // itself: the stored interval starts a little early); the interval's LENGTH tmax - tmin, rounded UP to its upper 16
    // float bits (15 without the sign), takes those 11 bits and the 4 bits a 28-bit node index leaves free in word 0.
    // The stored interval always contains the exact one (traversal stays conservative); the length being off by
    // < 1 % of itself costs next to nothing -- rounding tmax itself would misplace the end by 1 % of the DISTANCE,
    // many leaves' worth far from the origin.  Two words instead of three mean one LDS instruction per push or pop
    // (an LDS write costs a SIMD 18 - 27 cycles: tools/microbench/valu_issue.hip) and room for one more wave per SIMD.
    //
    // The stored tmin never increases towards the top of the stack: a new entry's tmin is lowered to the tmin of the
    // entry it is pushed onto if it lies above it (front-to-back traversal pushes in that order anyway; the thickened
    // planes break it by a slack now and then, and starting an interval early is always allowed).  So when the closest
    // hit lies in front of the TOP entry it lies in front of every entry, and a traversal ends with one look at the
    // stack instead of one pop per entry left -- that loop was 128 M wave iterations per 64 passes of the benchmark
    // frame, next to 224 M walk iterations.  `top` is the top entry's stored tmin (+infinity: stack empty).
    uint32_t* lds;          // this thread's column of words: lds[(2*e + c) * GPT_BLOCK]
    uint32_t* spill;        // this thread's column: spill[(2*e + c) * spill_stride]
    unsigned spill_stride;  // (2 * depth * stride fits 32 bits: checked by the host)
    int sp, lo;             // entries [lo, sp) are in LDS, [0, lo) in the spill buffer
    float top;

    static __device__ __forceinline__ void pack(uint32_t node, float tmin, float tmax, uint32_t& w0, uint32_t& w1) {
        const uint32_t t0 = __float_as_uint(tmin) & ~0x7ffu;
        const uint32_t len = (__float_as_uint(fmaxf(tmax - __uint_as_float(t0), 0.0f)) + 0xffffu) >> 16;
        w0 = node | ((len >> 11) << 28);
        w1 = t0 | (len & 0x7ffu);
    }
    static __device__ __forceinline__ void unpack(uint32_t w0, uint32_t w1, uint32_t& node, float& tmin, float& tmax) {
        node = w0 & PT_KD_CHILD_MASK;
        tmin = __uint_as_float(w1 & ~0x7ffu);
        const uint32_t len = ((w0 >> 28) << 11) | (w1 & 0x7ffu);
        // (the product covers the rounding of the subtraction in pack() and of this addition)
        tmax = (tmin + __uint_as_float(len << 16)) * 1.000001f;
    }

    // The spill path uses non-temporal accesses: written once, read once -- and a different
    // kind of access than the LDS path's, so the compiler cannot merge the two into generic
    // flat accesses.
    __device__ __forceinline__ void reset() { sp = 0; lo = 0; top = __uint_as_float(0x7f800000u); }
    __device__ __forceinline__ void push(uint32_t node, float tmin, float tmax) {
        const int slot = (int)((unsigned)sp % KD_LDS_N);
        // (asked of the whole wave first: a spill is rare, and a uniform branch around it costs one scalar instruction
        // where the divergent region alone costs five)
        if (__builtin_amdgcn_ballot_w64(sp - lo == KD_LDS_N) != 0ull) {
            asm volatile("");
            if (sp - lo == KD_LDS_N) {   // the oldest LDS entry occupies the slot about to be reused
                __builtin_nontemporal_store(lds[(2 * slot + 0) * GPT_BLOCK], spill + (unsigned)(2 * lo + 0) * spill_stride);
                __builtin_nontemporal_store(lds[(2 * slot + 1) * GPT_BLOCK], spill + (unsigned)(2 * lo + 1) * spill_stride);
                lo++;
            }
        }
        uint32_t w0, w1;
        pack(node, vmin(tmin, top), tmax, w0, w1);
        lds[(2 * slot + 0) * GPT_BLOCK] = w0;
        lds[(2 * slot + 1) * GPT_BLOCK] = w1;
        top = __uint_as_float(w1 & ~0x7ffu);
        sp++;
    }
    // pop() for the lanes in `take` (each of which has an entry), executed by the whole wave: the other lanes read
    // their top slot to no effect.  The spilled case is asked of the wave first, as in push().
    __device__ __forceinline__ void pop_if(KdPred take, uint32_t& node, float& tmin, float& tmax) {
        sp -= take.on() ? 1 : 0;
        // the LDS slot is read unconditionally (it is always a valid address) and overridden in the
        // rare spilled case; otherwise the compiler selects between the two ADDRESSES and emits
        // generic flat loads
        const int slot = (int)((unsigned)sp % KD_LDS_N);
        uint32_t w0 = lds[(2 * slot + 0) * GPT_BLOCK], w1 = lds[(2 * slot + 1) * GPT_BLOCK];
        // ... and the tmin word of the entry below, the new top
        const int below = (int)((unsigned)(sp - 1) % KD_LDS_N);
        uint32_t wb = lds[(2 * below + 1) * GPT_BLOCK];
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(wb));   // the values, not the addresses, meet below
        // the popped entry or the one below it lives in the spill buffer (lo > 0: only after a spill; a stack that merely
        // runs empty does not come here)
        const KdPred spilled = take & KdPred::of(sp <= lo && lo > 0);
        if (spilled.bits() != 0ull) {
            asm volatile("");
            if (spilled.on()) {
                if (sp < lo) {
                    w0 = __builtin_nontemporal_load(spill + (unsigned)(2 * sp + 0) * spill_stride);
                    w1 = __builtin_nontemporal_load(spill + (unsigned)(2 * sp + 1) * spill_stride);
                    lo = sp;
                }
                if (sp > 0) wb = __builtin_nontemporal_load(spill + (unsigned)(2 * (sp - 1) + 1) * spill_stride);
                asm volatile("" : "+v"(w0), "+v"(w1), "+v"(wb));   // wait here, in the rare branch, not at a later join
            }
        }
        unpack(w0, w1, node, tmin, tmax);
        const float below_tmin = sp > 0 ? __uint_as_float(wb & ~0x7ffu) : __uint_as_float(0x7f800000u);
        top = take.on() ? below_tmin : top;
    }
};

// ---------------------------------------------------------------------------
// k_intersect_brute: closest hit with the reference's loop over every triangle
// (:536-557), starting from the ray's closest ground / sphere hit.
// ---------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(GPT_BLOCK) void k_intersect_brute(DevGeometry g, int depth,
                                                               const float4* __restrict__ ray_o,
                                                               const float4* __restrict__ ray_d,
                                                               float2* __restrict__ hit,
                                                               unsigned* __restrict__ counts,
                                                               unsigned* __restrict__ work,
                                                               unsigned long long* __restrict__ stats) {
    const unsigned n = counts[depth];
    uint32_t cTris = 0, cHits = 0;
    // every ray costs the same here, so the queue is split statically (no work counter)
    const unsigned stride = gridDim.x * GPT_BLOCK;
    for (unsigned i = blockIdx.x * GPT_BLOCK + threadIdx.x; i < n; i += stride) {
        {
            const float4 ro = ld_stream(&ray_o[i]);
            const float4 rd = ld_stream(&ray_d[i]);
            const V3 o = v3(ro.x, ro.y, ro.z);
            const V3 d = v3(rd.x, rd.y, rd.z);
            const float2 h = hit[i];            // closest among ground and spheres
            int prim = __float_as_int(h.y);
            if (g.n_tris > 0) {
                float bt = h.x;
                int bi = -1;
                brute_closest_hit<STATS>(g.tris, g.n_tris, o, d, bt, bi, cTris);
                if (bi >= 0) {
                    prim = bi;
                    hit[i] = make_float2(bt, __int_as_float(bi));
                }
            }
            if (STATS && prim != PT_PRIM_MISS) cHits++;
        }
    }
    if (STATS) {
        atomicAdd(&stats[GPT_STAT_HITS], (unsigned long long)cHits);
        atomicAdd(&stats[GPT_STAT_KD_TRIS], (unsigned long long)cTris);
    }
}

// ---------------------------------------------------------------------------
// k_intersect_kd: closest hit with kd-tree traversal.
//
// Rays differ widely in traversal length (most miss the mesh's box at once, a
// few walk hundreds of nodes), and the kernel is bound by instruction issue: every
// wave instruction costs the same whether 64 lanes or 5 take part.  So a wave
// neither owns a fixed set of 64 rays nor lets each lane loop over its own
// leaf.  Each lane holds a resumable traversal cursor; the wave alternates
// between three phases:
//   refill:   idle lanes take the next rays of the global work queue (one
//             atomic per wave per 512 rays, ballot + prefix count per round)
//             together with their closest ground / sphere hit, and either
//             finish at once (ray misses the root box) or start a traversal;
//   walk:     lanes that stand at an inner node take one step each per
//             iteration; a lane that reaches a leaf (empty or not) parks there.
//             The phase ends when no lane walks any more or `leaf_min` lanes
//             are parked;
//   test:     the (ray, triangle) pairs of all parked lanes are enumerated
//             wave-wide (prefix sum of the leaves' counts) and dealt out 64 at
//             a time to ALL lanes: lane j tests pair base+j, fetching the
//             owner's ray through ds_bpermute, and folds a hit into the owner's
//             best with an LDS 64-bit atomic min on (t bits, index + 1) -- which
//             is exactly the reference's rule: smallest t, ties to the lowest
//             triangle index, and a triangle never displaces a ground / sphere
//             hit at the same t (their key has 0 in the low word).  Then the
//             parked lanes move on to their next deferred interval or finish.
// Once the queue is drained the remaining lanes simply run to completion.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long hit_key(float t, int tri) {
    return ((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(unsigned)(tri + 1);
}

__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v) {
    GPT_DPP_STEP(dpp_add, v, 0x111, 0xf);   // row_shr:1
    GPT_DPP_STEP(dpp_add, v, 0x112, 0xf);   // row_shr:2
    GPT_DPP_STEP(dpp_add, v, 0x114, 0xf);   // row_shr:4
    GPT_DPP_STEP(dpp_add, v, 0x118, 0xf);   // row_shr:8
    GPT_DPP_STEP(dpp_add, v, 0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    GPT_DPP_STEP(dpp_add, v, 0x143, 0xc);   // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ unsigned wave_inclusive_max(unsigned v) {
    GPT_DPP_STEP(dpp_max, v, 0x111, 0xf);
    GPT_DPP_STEP(dpp_max, v, 0x112, 0xf);
    GPT_DPP_STEP(dpp_max, v, 0x114, 0xf);
    GPT_DPP_STEP(dpp_max, v, 0x118, 0xf);
    GPT_DPP_STEP(dpp_max, v, 0x142, 0xa);
    GPT_DPP_STEP(dpp_max, v, 0x143, 0xc);
    return v;
}

__device__ __forceinline__ int bperm(unsigned byte_addr, int v) {
    return __builtin_amdgcn_ds_bpermute((int)byte_addr, v);
}
__device__ __forceinline__ float bperm(unsigned byte_addr, float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)byte_addr, __float_as_int(v)));
}


// kd_root_interval without a branch (the per-axis special case of a zero direction component costs the wave three
// divergent regions there).  The reciprocals are clamped to +-1e30 -- as good as infinite for any scene -- so that
// 0 * inf never arises: a ray parallel to a pair of planes gets [-huge, +huge] between them and an empty interval
// outside, which is the special case's answer.  A NaN or an infinity anywhere in the ray makes `probe` a NaN or an
// infinity, 0 * probe a NaN, and the final comparison false: such a ray hits nothing in the reference's loop.
__device__ __forceinline__ bool kd_root_interval_flat(const KdView& kd, float sigma, V3 o, V3 d, float t_limit, V3& inv,
                                                      float& tmin, float& tmax) {
    inv = v3(__builtin_amdgcn_fmed3f(kd_rcp(d.x), -1.0e30f, 1.0e30f), __builtin_amdgcn_fmed3f(kd_rcp(d.y), -1.0e30f, 1.0e30f),
             __builtin_amdgcn_fmed3f(kd_rcp(d.z), -1.0e30f, 1.0e30f));
    const float oc[3] = {o.x, o.y, o.z};
    const float ic[3] = {inv.x, inv.y, inv.z};
    float t0 = 0.0f, t1 = t_limit;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float ta = ((kd.bmin[a] - sigma) - oc[a]) * ic[a], tb = ((kd.bmax[a] + sigma) - oc[a]) * ic[a];
        const float tn = vmin(ta, tb), tf = vmax(ta, tb);
        t0 = vmax(t0, __builtin_fmaf(-PT_KD_REL_SLACK, fabsf(tn), tn));
        t1 = vmin(t1, __builtin_fmaf(PT_KD_REL_SLACK, fabsf(tf), tf));
    }
    const float probe = ((o.x + o.y) + (o.z + d.x)) + (d.y + d.z);
    tmin = t0;
    tmax = t1;
    return t0 <= __builtin_fmaf(probe, 0.0f, t1);
}

// FindTriangleIntersect's three results without its early return: with det == 0 the quotients are infinities or NaNs,
// and no finite t comes out (the caller's rule "a hit must beat the best t so far" then rejects what the early return
// rejects).  Same operations in the same order as pt::triangle_t otherwise.
__device__ __forceinline__ void triangle_uvt(V3 a, V3 edge1, V3 edge2, V3 o, V3 d, float& bary1, float& bary2, float& t) {
    const V3 pvec = cross(d, edge2);
    const float det = dot(edge1, pvec);
    const float inv_det = 1.0f / det;
    const V3 tvec = o - a;
    bary1 = dot(tvec, pvec) * inv_det;
    const V3 qvec = cross(tvec, edge1);
    bary2 = dot(d, qvec) * inv_det;
    t = dot(edge2, qvec) * inv_det;
}

// PACKED: the leaves hold the offsets of their runs of triangle records (leaf-ordered layout); otherwise they hold
// positions in the index list g.kd.refs.
template <bool STATS, bool PACKED>
__global__ __launch_bounds__(GPT_BLOCK, KD_WAVES_PER_EU) void k_intersect_kd(DevGeometry g, int depth,
                                                            const float4* __restrict__ ray_o,
                                                            const float4* __restrict__ ray_d,
                                                            float2* __restrict__ hit,
                                                            unsigned* __restrict__ counts,
                                                            unsigned* __restrict__ work,
                                                            unsigned long long* __restrict__ stats,
                                                            uint32_t* __restrict__ spill, size_t spill_stride,
                                                            unsigned refill_below, unsigned leaf_min) {
    __shared__ uint32_t s_stack[2 * KD_LDS_N * GPT_BLOCK];
    __shared__ unsigned long long s_best[GPT_BLOCK];   // per lane: hit_key of the best hit so far
    __shared__ unsigned char s_owner[GPT_BLOCK];       // per wave: 64 marks of the test phase's owner search
    const unsigned n = counts[depth];

    // The scalar unit is this kernel's scarcest resource: a CU issues about one scalar instruction per cycle for its
    // four SIMDs together (4.3 cycles per instruction and SIMD, against 2.4 for a plain vector instruction and 3.7 for a
    // compare or a select: tools/microbench/valu_issue.hip), and eight more scalar instructions per walk iteration cost
    // 7 - 9 % of the kernel's time, eight more vector instructions 1.5 %.  Hence the shape of everything below: which
    // lanes hold a ray, wait at a leaf or look for their next interval is kept in wave masks (KdPred) that the code
    // updates itself, not in per-lane bools (whose merging at every join the compiler spells out in scalar
    // instructions); loops run on wave-uniform conditions; and decisions are made numerically where that saves masks.
    KdCursor<DevStack, STATS> cur;
    cur.st.lds = s_stack + threadIdx.x;
    cur.st.spill = spill + ((size_t)blockIdx.x * GPT_BLOCK + threadIdx.x);
    cur.st.spill_stride = (unsigned)spill_stride;
    cur.st.reset();
    const unsigned lane = lane_id();
    const unsigned wave_base = threadIdx.x & ~63u;

    uint32_t cNodes = 0, cTris = 0, cHits = 0, cBusy = 0, cRounds = 0;
    KdPred haveM = KdPred::none();     // lanes that hold an unfinished traversal
    bool drained = false;       // wave-uniform: the work queue is exhausted
    unsigned slot = 0;          // queue slot of the ray this lane is tracing
    int prim0 = PT_PRIM_MISS;   // best among ground / spheres for that ray
    unsigned leafFirst = 0, leafCount = 0;

    // The wave takes rays from the global queue KD_FETCH_CHUNK at a time (one atomic on
    // the shared counter per chunk) and hands them to idle lanes from that local range.
    // The queue is cut into 8 contiguous parts with a head word each; a workgroup starts in the
    // part of its XCD (blocks b and b + 8 share one, and with it an L2: rays that are neighbours
    // in the queue -- neighbouring pixels -- then warm the same L2) and moves on to the other
    // parts when its own is exhausted.
    unsigned part = blockIdx.x & (GPT_WORK_PARTS - 1u), parts_done = 0;
    unsigned local_next = 0, local_end = 0;
    // A short queue (one pass per launch, a caller's ray set) in chunks of KD_FETCH_CHUNK would occupy a fraction of the
    // grid's waves, each working through its 512 rays 30 at a time: the chunk shrinks until every wave gets two.
    unsigned chunk = KD_FETCH_CHUNK;
#ifndef KD_FIXED_CHUNK
    {
        const unsigned per_wave = n / (gridDim.x * (GPT_BLOCK / 64u) * 2u);
        if (per_wave < KD_FETCH_CHUNK) chunk = per_wave < 64u ? 64u : (per_wave & ~63u);
    }
#endif
    for (;;) {
        // ---- refill ----
        while (!drained) {
            const unsigned long long idle = ~haveM.bits();
            const unsigned cnt = (unsigned)__popcll(idle);
            if (cnt == 0u || (64u - cnt) >= refill_below) break;
            if (local_next >= local_end) {
                // next chunk of the current part of the queue; an exhausted part is left for good
                const unsigned part_lo = (unsigned)((unsigned long long)n * part / GPT_WORK_PARTS);
                const unsigned part_hi = (unsigned)((unsigned long long)n * (part + 1u) / GPT_WORK_PARTS);
                const unsigned base = part_lo + wave_fetch(&work[(depth * GPT_WORK_PARTS + part) * GPT_WORK_STRIDE],
                                                           chunk);
                if (base >= part_hi) {
                    part = (part + 1u) & (GPT_WORK_PARTS - 1u);
                    if (++parts_done == GPT_WORK_PARTS) drained = true;
                    continue;
                }
                local_next = base;
                local_end = (base + chunk < part_hi) ? base + chunk : part_hi;
            }
            const unsigned avail = local_end - local_next;
            const unsigned take = cnt < avail ? cnt : avail;
            bool started = false;
            if (prefix_count(idle) < take && !haveM.on()) {
                const unsigned i = local_next + prefix_count(idle);
                const float4 ro = ld_stream(&ray_o[i]);
                const float4 rd = ld_stream(&ray_d[i]);
                float2 h = hit[i];          // closest among ground and spheres
                // (consumed here: a load still pending when the walk resumes would make every
                // later wait a wait for ALL outstanding loads)
                asm volatile("" : "+v"(h.x), "+v"(h.y));
                // (every lane of the region takes the ray; one that misses the root box just is not counted into
                // haveM and is overwritten by the next refill)
                cur.o = v3(ro.x, ro.y, ro.z);
                cur.d = v3(rd.x, rd.y, rd.z);
                cur.sigma = kd_ray_sigma(g.kd, cur.o);
                started = kd_root_interval_flat(g.kd, cur.sigma, cur.o, cur.d, h.x, cur.inv, cur.tmin, cur.tmax);
                cur.node = 0;
                cur.st.reset();
                slot = i;
                prim0 = __float_as_int(h.y);
                s_best[threadIdx.x] = hit_key(h.x, -1);
                if (STATS && !started && prim0 != PT_PRIM_MISS) cHits++;
            }
            haveM = haveM | KdPred::of(started);
            local_next += take;
        }
        const unsigned nHave = haveM.count();
        if (nHave == 0u) {
            if (drained) break;
            continue;
        }
        // ---- walk ----
        // One tree level per iteration for the lanes in walkM (everything that happens once per leaf -- next interval,
        // end of traversal -- is left to the phases after it).  The step is KdCursor::node_step's, decided numerically:
        // it runs with the walking lanes as the execution mask (every comparison's lane mask is then already confined
        // to them); each child gets its interval, and a child is visited iff its interval is not empty -- two
        // comparisons.  Which child is the near one, whether a child is an empty leaf and whether the plane lies
        // behind the origin all enter as SIGN BITS or NaNs folded into the operands of those two comparisons instead
        // of as masks of their own:
        //   near child  [tmin, min(hi, tmax)], or [tmin, tmax] when lo < 0 (the crossing may lie behind the origin);
        //   far child   [max(lo, tmin), tmax], empty when hi < 0;
        // where [lo, hi] is the thickened plane's range of ray parameters.  With the reciprocals clamped to +-1e30 no
        // infinities or NaNs arise, and a ray parallel to the plane gets a crossing so far out or so far back that the
        // same rule decides it as node_step does.  Exactly on the plane both children are visited whole, in either order.
        KdPred walkM = haveM;                                   // (no lane waits at a leaf when the phase starts)
        const unsigned parkGoal = nHave < leaf_min ? nHave : leaf_min;
        const unsigned walkStop = (unsigned)__builtin_amdgcn_readfirstlane((int)(nHave - parkGoal));   // leave once no more lanes than this still walk
        do {
#ifdef ABL_WALK_SPAD  // experiment: ABL_WALK_SPAD extra SCALAR instructions per walk iteration (is the loop bound by scalar issue?)
            {
#pragma unroll
                for (int q_ = 0; q_ < ABL_WALK_SPAD; q_++) asm volatile("s_add_u32 s100, s100, 1" : : : "s100", "scc");
            }
#endif
#ifdef ABL_WALK_PAD   // experiment: ABL_WALK_PAD extra vector instructions per walk iteration (is the loop VALU-bound?)
            {
                unsigned pad_ = lane;
#pragma unroll
                for (int q_ = 0; q_ < ABL_WALK_PAD; q_++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pad_));
            }
#endif
            bool cont = false;
            if (walkM.on()) {
                if (STATS) { PT_COUNT_WAVE_ITERATION(0); cNodes++; }
                const KdNode w = g.kd.nodes[cur.node];
                const uint32_t w0 = w.w0, SIGN = 0x80000000u;
                const unsigned axis = w0 & 3u;
                const KdPred innerM = KdPred::of(axis != 3u);
                const float oa = sel3((int)axis, cur.o.x, cur.o.y, cur.o.z);
                const float ia = sel3((int)axis, cur.inv.x, cur.inv.y, cur.inv.z);
                const float key = __uint_as_float(w.w1) - oa;       // > 0: the origin lies left of the plane
                const uint32_t s = __float_as_uint(key) >> 31;      // 1: the right child is the near one
                const uint32_t sx = s ^ 1u;
                const uint32_t left = (w0 >> 2) & PT_KD_CHILD_MASK;
                const uint32_t nearChild = left + s, farChild = left + sx;
                const uint32_t wN = w0 << sx, wF = w0 << s;         // bit 31: the near / far child is an empty leaf
                const float tp = key * ia;
                const float slack = __builtin_fmaf(PT_KD_REL_SLACK, fabsf(tp), cur.sigma * fabsf(ia));
                const float lo = tp - slack, hi = tp + slack;
                // The interval ends below are compared as signed INTEGERS: for the non-negative floats involved that is
                // the float order, and an end with the sign bit set lies below every start, +0 included (as a float, an
                // interval [0, -0] would not be empty -- and a ray whose search ends at t = 0 would wander through half the tree).
                // far child: the end's sign bit is set when hi < 0 or the child is an empty leaf
                const float farTmin = vmax(lo, cur.tmin);
                const int32_t farEnd = (int32_t)(((__float_as_uint(hi) | wF) & SIGN) | __float_as_uint(cur.tmax));
                const KdPred farM = KdPred::of(__float_as_int(farTmin) <= farEnd);
                // near child: hi becomes a NaN when lo < 0 (the minimum then returns tmax); the end's sign as above
                const uint32_t behind = (uint32_t)((int32_t)__float_as_uint(lo) >> 31) & 0x7fc00000u;
                const float nearTmax = vmin(__uint_as_float(__float_as_uint(hi) | behind), cur.tmax);
                const int32_t nearEnd = (int32_t)((wN & SIGN) | __float_as_uint(nearTmax));
                const KdPred nearM = KdPred::of(nearEnd >= __float_as_int(cur.tmin));
                if ((innerM & nearM & farM).on()) cur.st.push(farChild, farTmin, cur.tmax);
                // (a lane at a leaf computes garbage here: it takes node, tmin and tmax from its next stack entry)
                cur.node = nearM.on() ? nearChild : farChild;
                cur.tmin = nearM.on() ? cur.tmin : farTmin;
                cur.tmax = nearM.on() ? nearTmax : cur.tmax;
                // a lane stops at a leaf, or where neither child is worth a visit: nothing to test there
                leafFirst = w0 >> 2;
                leafCount = innerM.on() ? 0u : w.w1;                // may be 0 at a leaf, too: an empty one
                cont = (innerM & (nearM | farM)).on();
            }
            walkM = KdPred::of(cont);
            if (STATS) { if (haveM.on()) cBusy++; if (lane == 0) cRounds++; }
        } while (walkM.count() > walkStop);
        const KdPred parkedM = haveM & ~walkM;    // lanes that stand at a leaf whose triangles are tested now
        // ---- test ----
        {
            // `first`: number of the lane's first (ray, triangle) pair; none for a lane without triangles
            const unsigned cnt = parkedM.on() ? leafCount : 0u;
            const unsigned unit = PACKED ? (unsigned)(sizeof(TriRec) / 8) : 1u;
            const unsigned incl = wave_inclusive_sum(cnt);
            const unsigned excl = incl - cnt;
            const unsigned first = cnt != 0u ? excl : 0xffffffffu;
            const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            for (unsigned base = 0; base < total; base += 64) {
                if (STATS) PT_COUNT_WAVE_ITERATION(1);
                const unsigned p = base + lane;
                // owner of pair p = last lane with triangles whose first pair number is <= p: the lanes
                // whose run starts inside this chunk mark the start in a 64-byte LDS row, a running
                // maximum spreads the marks, and the run reaching in from the previous chunk belongs to
                // the last lane that starts at or before `base`
                s_owner[threadIdx.x] = 0;
                if (first - base < 64u) s_owner[wave_base + (first - base)] = (unsigned char)lane;
                __builtin_amdgcn_wave_barrier();
                const unsigned mark = s_owner[threadIdx.x];
                const unsigned long long before = KdPred::of(first <= base).bits();
                const unsigned carry = 63u - (unsigned)__builtin_clzll(before);
                const unsigned L = dpp_max(wave_inclusive_max(mark), carry);
                const unsigned la = L << 2;
                // owner's first record + (p - owner's first pair); in 8-byte units when the leaves hold
                // the offsets of their packed runs, in entries of the index list otherwise
                const unsigned r0u = unit * p + (unsigned)bperm(la, (int)(leafFirst - unit * excl));
                const V3 o = v3(bperm(la, cur.o.x), bperm(la, cur.o.y), bperm(la, cur.o.z));
                const V3 d = v3(bperm(la, cur.d.x), bperm(la, cur.d.y), bperm(la, cur.d.z));
#ifdef ABL_BPERM_X2   // experiment: the six ray ds_bpermutes of a chunk issued a second time (what do they cost?)
                {
                    float x0 = bperm(la, cur.o.y), x1 = bperm(la, cur.o.z), x2 = bperm(la, cur.o.x);
                    float x3 = bperm(la, cur.d.y), x4 = bperm(la, cur.d.z), x5 = bperm(la, cur.d.x);
                    asm volatile("" : : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5));
                }
#endif
#ifdef ABL_TRI_PAD
                {
                    unsigned pad_ = lane;
#pragma unroll
                    for (int q_ = 0; q_ < ABL_TRI_PAD; q_++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pad_));
                }
#endif
                if (p < total) {
                    const float* rp = PACKED ? reinterpret_cast<const float*>(g.kd.leaf_tris) + 2 * (size_t)r0u
                                             : g.kd.leaf_tris[g.kd.refs[r0u]].f;
                    // the whole record at once (left alone, the compiler fetches `a` only after the
                    // determinant test: a second memory latency in every chunk); 8-byte aligned
                    typedef float f4a8 __attribute__((ext_vector_type(4), aligned(8)));
                    typedef float f2a8 __attribute__((ext_vector_type(2), aligned(8)));
#ifdef KD_TRI_NT   // experiment: triangle records fetched with the non-temporal hint (they are 20x the nodes and reused less)
                    const f4a8 q0 = __builtin_nontemporal_load(reinterpret_cast<const f4a8*>(rp)), q1 = __builtin_nontemporal_load(reinterpret_cast<const f4a8*>(rp + 4));
                    const f2a8 q2 = __builtin_nontemporal_load(reinterpret_cast<const f2a8*>(rp + 8));
#else
                    const f4a8 q0 = *reinterpret_cast<const f4a8*>(rp), q1 = *reinterpret_cast<const f4a8*>(rp + 4);
                    const f2a8 q2 = *reinterpret_cast<const f2a8*>(rp + 8);
#endif
                    float4 r0 = make_float4(q0.x, q0.y, q0.z, q0.w), r1 = make_float4(q1.x, q1.y, q1.z, q1.w);
                    float4 r2 = make_float4(q2.x, q2.y, 0.0f, 0.0f);
                    asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z),
                                 "+v"(r1.w), "+v"(r2.x), "+v"(r2.y));
                    const V3 a = v3(r0.x, r0.y, r0.z), e1 = v3(r0.w, r1.x, r1.y), e2 = v3(r1.z, r1.w, r2.x);
                    const int idx = __float_as_int(r2.y);
                    if (STATS) cTris++;
                    // FindTriangleIntersect's acceptance (bary1 >= 0, bary2 >= 0, bary1 + bary2 <= 1) and the caller's t > 0,
                    // in two comparisons: 1 - (bary1 + bary2) has the sign of the third condition exactly, and the
                    // smallest of the three is >= 0 iff all are (-0 included).
                    float b1, b2, t;
                    triangle_uvt(a, e1, e2, o, d, b1, b2, t);
                    const float least = vmin(vmin(b1, b2), 1.0f - (b1 + b2));
                    if (least >= 0.0f && t > 0.0f) atomicMin(&s_best[wave_base + L], hit_key(t, idx));
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- next interval ----
        // The lanes that waited at a leaf take their next deferred interval if it can still hold a closer (or equal) hit;
        // a lane that has none left, or whose next one starts behind its closest hit, is done.
        {
            const unsigned long long key = __atomic_load_n(&s_best[threadIdx.x], __ATOMIC_RELAXED);
            const float bestT = __uint_as_float((unsigned)(key >> 32));
            // (the stack's tmin values do not increase towards the top -- DevStack --, so a top entry that starts behind
            // the closest hit ends the traversal: every other entry starts behind it, too)
            const KdPred takeM = parkedM & KdPred::of(cur.st.sp != 0);
            uint32_t node_;
            float tmin_, tmax_;
            cur.st.pop_if(takeM, node_, tmin_, tmax_);
            const KdPred acceptM = takeM & ~KdPred::of(bestT < tmin_);
            cur.node = acceptM.on() ? node_ : cur.node;
            cur.tmin = acceptM.on() ? tmin_ : cur.tmin;
            cur.tmax = acceptM.on() ? vmin(tmax_, bestT) : cur.tmax;
            const KdPred doneM = parkedM & ~acceptM;
            // The traversal of these lanes is over: a triangle that beat the ground / sphere hit replaces it in the ray's hit slot.
            const unsigned tri1 = (unsigned)key;     // triangle index + 1, or 0
            if (doneM.on() && tri1 != 0u) hit[slot] = make_float2(bestT, __int_as_float((int)(tri1 - 1u)));
            if (STATS && doneM.on() && (tri1 != 0u || prim0 != PT_PRIM_MISS)) cHits++;
            haveM = haveM & ~doneM;
        }
    }
    if (STATS) {
        atomicAdd(&stats[GPT_STAT_KD_NODES], (unsigned long long)cNodes);
        atomicAdd(&stats[GPT_STAT_KD_TRIS], (unsigned long long)cTris);
        atomicAdd(&stats[GPT_STAT_HITS], (unsigned long long)cHits);
        atomicAdd(&stats[GPT_STAT_NODES_D0 + depth], (unsigned long long)cNodes);
        atomicAdd(&stats[GPT_STAT_BUSY_LANE_ROUNDS], (unsigned long long)cBusy);
        atomicAdd(&stats[GPT_STAT_WAVE_ROUNDS], (unsigned long long)cRounds);
        atomicAdd(&stats[GPT_STAT_TRIS_D0 + depth], (unsigned long long)cTris);
    }
}

// ---------------------------------------------------------------------------
// k_shade: medium + BSDF for every live path, compaction into the next queue.
//
// Work is split statically (block-strided tiles of 256 paths).  Survivors are
// not appended to the global queue one wave at a time -- a single hot counter
// serialises at roughly 12 ns per atomic on MI355X -- but staged in LDS per
// block: wave64 ballots + prefix counts place each survivor in the block's
// staging area, and whenever 512 have gathered the block reserves 512 queue
// slots with ONE atomic and writes them out as full coalesced rows.
// ---------------------------------------------------------------------------
#ifndef SHADE_FLUSH
#define SHADE_FLUSH 512
#endif
#ifndef SHADE_WAVES_PER_EU
#define SHADE_WAVES_PER_EU 4
#endif
#ifndef SHADE_BLOCK
#define SHADE_BLOCK 256   // threads per workgroup of k_shade
#endif
#define SHADE_STAGE (SHADE_FLUSH + SHADE_BLOCK)
// (one flush per tile: what a tile can add -- SHADE_BLOCK survivors -- must not exceed a flush, or the staging area overflows)
static_assert(SHADE_FLUSH >= SHADE_BLOCK && SHADE_FLUSH % SHADE_BLOCK == 0, "SHADE_FLUSH must be a multiple of SHADE_BLOCK");

// FIRST: the launch that shades the primary rays.  k_camera stores no throughput for them: it is (1, 1, 1),
// and slot i of the still uncompacted queue belongs to pass i - pixel * numPasses.
// DEBUG: every bounce's branch code (PT_EV_*) and primitive go to dbg_events / dbg_prims at
// [pixel * dbg_stride + depth] (single-pass launches only; parity checks read them).
// FUSED (with FIRST): a scene without triangles launches no intersect kernel, so nothing but this launch would read the
// primary rays k_camera stores -- it creates them itself (k_camera's own functions: camera_ray, the per-pixel sphere
// masks, the ground / sphere tests) and k_camera is not launched: 40 bytes written and 40 read per primary path less.
template <bool FIRST, bool DEBUG, bool FUSED>
__global__ __launch_bounds__(SHADE_BLOCK, SHADE_WAVES_PER_EU) void k_shade(DevGeometry g, ShadeParams sp, CamPre cam,
                                                     const uint32_t* __restrict__ masks, int depth, uint32_t seed0,
                                                     int numPasses,
                                                     const float4* __restrict__ in_o,
                                                     const float4* __restrict__ in_d,
                                                     const float4* __restrict__ in_thr,
                                                     const float2* __restrict__ hit,
                                                     float4* __restrict__ out_o,
                                                     float4* __restrict__ out_d,
                                                     float4* __restrict__ out_thr,
                                                     float2* __restrict__ out_hit,
                                                     float4* __restrict__ pass_rgb,
                                                     unsigned* __restrict__ counts,
                                                     unsigned long long* __restrict__ stats,
                                                     unsigned char* __restrict__ dbg_events,
                                                     int* __restrict__ dbg_prims, int dbg_stride) {
    __shared__ float4 s_o[SHADE_STAGE];
    __shared__ float4 s_d[SHADE_STAGE];
    __shared__ float4 s_t[SHADE_STAGE];
    __shared__ unsigned s_wave_cnt[SHADE_BLOCK / 64];
    __shared__ unsigned s_base;

    const unsigned n = counts[depth];
    if (blockIdx.x == 0 && threadIdx.x == 0) stats[GPT_STAT_LIVE0 + depth] += n;
    const unsigned lane = lane_id();
    const unsigned wave = threadIdx.x >> 6;
    static_assert(!FUSED || FIRST, "only the launch of depth 0 can create the primary rays");
    // a surviving path leaves with its new ray and that ray's closest ground / sphere hit
    auto emit = [&](unsigned pos, unsigned k) {
        const float4 ro = s_o[k], rd = s_d[k];
        st_stream(&out_o[pos], ro);
        st_stream(&out_d[pos], rd);
        st_stream(&out_thr[pos], s_t[k]);
        float bestT;
        int prim;
        analytic_hits(g, v3(ro.x, ro.y, ro.z), v3(rd.x, rd.y, rd.z), bestT, prim);
        st_stream(&out_hit[pos], make_float2(bestT, __int_as_float(prim)));
    };
    unsigned staged = 0;                        // block-uniform: survivors waiting in LDS
    const unsigned stride = gridDim.x * SHADE_BLOCK;
    for (unsigned tile = blockIdx.x * SHADE_BLOCK; tile < n; tile += stride) {
        const unsigned i = tile + threadIdx.x;
        bool alive = false;
        V3 o = v3(0, 0, 0), d = v3(0, 0, 0), thr = v3(0, 0, 0);
        int pixel = 0, medium = -1;
        unsigned pass = 0;
        float4 ro = make_float4(0, 0, 0, 0), rd = ro;
        float2 h = make_float2(0, 0);
        if (FUSED) {
            // as k_camera does it (all lanes of the wave take part: the union of the wave's sphere masks is formed with DPP)
            const bool in = i < n;
            const int px = (int)((in ? i : n - 1u) / (unsigned)numPasses);
            V3 co, cd;
            camera_ray(cam, px, seed0 + ((in ? i : n - 1u) - (unsigned)px * (unsigned)numPasses), co, cd);
            float bt;
            int pr;
            if (masks) {
                unsigned m = in ? masks[px] : 0u;
                GPT_DPP_STEP(dpp_or, m, 0x111, 0xf);
                GPT_DPP_STEP(dpp_or, m, 0x112, 0xf);
                GPT_DPP_STEP(dpp_or, m, 0x114, 0xf);
                GPT_DPP_STEP(dpp_or, m, 0x118, 0xf);
                GPT_DPP_STEP(dpp_or, m, 0x142, 0xa);
                GPT_DPP_STEP(dpp_or, m, 0x143, 0xc);
                analytic_hits_masked(g, co, cd, (uint32_t)__builtin_amdgcn_readlane((int)m, 63), bt, pr);
            } else {
                analytic_hits(g, co, cd, bt, pr);
            }
            ro = make_float4(co.x, co.y, co.z, __int_as_float(px));
            rd = make_float4(cd.x, cd.y, cd.z, __int_as_float(-1));
            h = make_float2(bt, __int_as_float(pr));
        }
        if (i < n) {
            if (!FUSED) {
                ro = ld_stream(&in_o[i]);
                rd = ld_stream(&in_d[i]);
                h = ld_stream(&hit[i]);
            }
            const float4 th = FIRST ? make_float4(1.0f, 1.0f, 1.0f, __uint_as_float(i - (unsigned)__float_as_int(ro.w) * (unsigned)numPasses))
                                    : ld_stream(&in_thr[i]);
            o = v3(ro.x, ro.y, ro.z);
            d = v3(rd.x, rd.y, rd.z);
            thr = v3(th.x, th.y, th.z);
            pixel = __float_as_int(ro.w);
            medium = __float_as_int(rd.w);
            // (bit 31 of the pass word: the path has added to its radiance slot before -- see the add below)
            pass = __float_as_uint(th.w) & 0x7fffffffu;
            bool dirty = (__float_as_uint(th.w) >> 31) != 0u;
            const float bestT = h.x;
            const int prim = __float_as_int(h.y);

            // Everything that depends on the inputs alone is requested at once (the medium's
            // coefficients, the sphere or the triangle's normal and material index), then the material;
            // the sphere normal is worked out while the material is on its way.
            const bool isSphere = prim >= 0 && (prim & PT_PRIM_SPHERE) != 0;
            const bool isTri = prim >= 0 && !isSphere;
            V3 sa = v3(0, 0, 0);
            float ss = 0.0f;
            float4 m2 = make_float4(0, 0, 0, 0), m3 = m2, geo = m2;
            int matIndex = -1;
            if (medium >= 0) {
                m2 = g.mats[4 * medium + 2];
                m3 = g.mats[4 * medium + 3];
            }
            if (isSphere) {
                const int s = prim & ~PT_PRIM_SPHERE;
                geo = g.spheres[s];
                matIndex = g.sphere_mat[s];
            } else if (isTri) {
                geo = g.tri_nm[prim];
                matIndex = __float_as_int(geo.w);
            }
            Mat m = ground_material();
            if (prim >= 0) m = load_material(g.mats, matIndex);
            V3 nrm = v3(0, 1, 0);
            if (isSphere) nrm = sphere_normal(v3(geo.x, geo.y, geo.z), o, d, bestT);
            else if (isTri) nrm = v3(geo.x, geo.y, geo.z);
            if (medium >= 0) {
                ss = m2.w;
                sa = v3(m3.x, m3.y, m3.z);
            }
            V3 add;
            bool has_add;
#ifdef ABL_SHADE_PAD   // experiment: extra vector instructions per shaded path (is k_shade VALU-bound?)
            {
                unsigned pad_ = lane;
#pragma unroll
                for (int q_ = 0; q_ < ABL_SHADE_PAD; q_++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(pad_));
            }
#endif
            int event = 0;
            alive = shade_bounce(o, d, thr, medium, sa, ss, prim != PT_PRIM_MISS, bestT, nrm, m, matIndex,
                                 pixel, depth, seed0 + pass, sp, add, has_add, event);
            if (DEBUG) {
                dbg_events[(size_t)pixel * (size_t)dbg_stride + (size_t)depth] = (unsigned char)event;
                dbg_prims[(size_t)pixel * (size_t)dbg_stride + (size_t)depth] = prim;
            }
            if (FIRST) {
                // slot i of the uncompacted queue IS (pixel, pass): the first bounce stores 0 + add (the sum spelled
                // out: it turns a -0 into the +0 the reference's accumulator would hold) instead of read-modify-write
                st_stream(&pass_rgb[i], has_add ? make_float4(0.0f + add.x, 0.0f + add.y, 0.0f + add.z, 0.0f)
                                                : make_float4(0.0f, 0.0f, 0.0f, 0.0f));
                dirty = has_add;
            } else if (has_add) {
                // A slot nothing was added to since the first bounce holds +0 (stored above, by the launch that shaded
                // the primary rays): the sum is 0 + add and needs no read -- for the usual path, whose only contribution
                // is the sky it ends in, that halves the scattered 16-byte accesses of this kernel.
                const size_t slot = (size_t)pixel * (size_t)numPasses + (size_t)pass;
                float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (dirty) acc = pass_rgb[slot];
                acc.x += add.x; acc.y += add.y; acc.z += add.z;
                pass_rgb[slot] = acc;
                dirty = true;
            }
            pass |= dirty ? 0x80000000u : 0u;
        }
#ifdef ABL_SHADE_SYNC   // experiment: extra block barriers per tile (what do the two of the compaction cost?)
#pragma unroll
        for (int q_ = 0; q_ < ABL_SHADE_SYNC; q_++) __syncthreads();
#endif
        // ---- block compaction into the LDS staging area ----
        const unsigned long long mask = __ballot(alive);
        if (lane == 0) s_wave_cnt[wave] = (unsigned)__popcll(mask);
        __syncthreads();
        unsigned before = 0, total = 0;
#pragma unroll
        for (unsigned w = 0; w < SHADE_BLOCK / 64; w++) {
            const unsigned c = s_wave_cnt[w];
            if (w < wave) before += c;
            total += c;
        }
        if (alive) {
            const unsigned pos = staged + before + prefix_count(mask);
            s_o[pos] = make_float4(o.x, o.y, o.z, __int_as_float(pixel));
            s_d[pos] = make_float4(d.x, d.y, d.z, __int_as_float(medium));
            s_t[pos] = make_float4(thr.x, thr.y, thr.z, __uint_as_float(pass));
        }
        staged += total;
        __syncthreads();
        if (staged >= SHADE_FLUSH) {
            if (threadIdx.x == 0) s_base = atomicAdd(&counts[depth + 1], (unsigned)SHADE_FLUSH);
            __syncthreads();
            const unsigned base = s_base;
#pragma unroll
            for (unsigned k = threadIdx.x; k < SHADE_FLUSH; k += SHADE_BLOCK) emit(base + k, k);
            // move the remainder (fewer entries than the block has threads) to the front
            const unsigned rem = staged - SHADE_FLUSH;
            float4 ko = make_float4(0, 0, 0, 0), kd_ = ko, kt = ko;
            if (threadIdx.x < rem) {
                ko = s_o[SHADE_FLUSH + threadIdx.x];
                kd_ = s_d[SHADE_FLUSH + threadIdx.x];
                kt = s_t[SHADE_FLUSH + threadIdx.x];
            }
            __syncthreads();
            if (threadIdx.x < rem) {
                s_o[threadIdx.x] = ko;
                s_d[threadIdx.x] = kd_;
                s_t[threadIdx.x] = kt;
            }
            staged = rem;
            __syncthreads();
        }
    }
    // ---- flush what is left ----
    if (staged > 0) {
        if (threadIdx.x == 0) s_base = atomicAdd(&counts[depth + 1], staged);
        __syncthreads();
        const unsigned base = s_base;
        for (unsigned k = threadIdx.x; k < staged; k += SHADE_BLOCK) emit(base + k, k);
    }
}

// ---------------------------------------------------------------------------
// k_finish: image += pass for every pass of the batch, in counter order, so a
// pixel's float sum is the one PathTracer::render builds (path_tracer.cpp:100-103).
// Optionally every pass of the batch on its own as packed float3, pass after pass
// (pass k at pass_out3 + k * 3 * numPixels): what the reference's D2H copy hands
// to the host per pass (path_tracer_kernel.cu:869).
// ---------------------------------------------------------------------------
// A pixel's P pass values lie next to each other (16 P bytes), so a thread that walked its own
// pixel's row would fetch one cache line per lane and load.  Instead the block stages tiles of
// 64 pixels x 32 passes through LDS: half a wave reads the 512 contiguous bytes of one pixel's
// 32 passes, and lane j of the first wave then adds row j in pass order.
#define FIN_PIXELS 64
#define FIN_PASSES 32
__global__ __launch_bounds__(GPT_BLOCK) void k_finish(int numPixels, int numPasses, const float4* __restrict__ pass_rgb,
                                                      float4* __restrict__ image, float* __restrict__ pass_out3) {
    __shared__ float4 s_tile[FIN_PIXELS][FIN_PASSES + 1];
    const int t = (int)threadIdx.x;
    for (int pix0 = (int)blockIdx.x * FIN_PIXELS; pix0 < numPixels; pix0 += (int)gridDim.x * FIN_PIXELS) {
        const int i = pix0 + t;
        const bool owner = t < FIN_PIXELS && i < numPixels;
        float4 a = make_float4(0.0f, 0.0f, 0.0f, 0.0f), p = a;
        if (owner) a = image[i];
        for (int k0 = 0; k0 < numPasses; k0 += FIN_PASSES) {
            const int kn = numPasses - k0 < FIN_PASSES ? numPasses - k0 : FIN_PASSES;
            const int c = t & (FIN_PASSES - 1);
#pragma unroll
            for (int r = t / FIN_PASSES; r < FIN_PIXELS; r += GPT_BLOCK / FIN_PASSES)
                if (c < kn && pix0 + r < numPixels)
                    s_tile[r][c] = ld_stream(&pass_rgb[(size_t)(pix0 + r) * (size_t)numPasses + (size_t)(k0 + c)]);
            __syncthreads();
            if (owner)
                for (int k = 0; k < kn; k++) {
                    p = s_tile[t][k];
                    a.x += p.x; a.y += p.y; a.z += p.z; a.w += 1.0f;
                }
            if (pass_out3) {
                // all 256 threads: pass k of the tile's 64 pixels is 192 consecutive floats of the output
                for (int e = t; e < kn * FIN_PIXELS * 3; e += GPT_BLOCK) {
                    const int k = e / (FIN_PIXELS * 3), w = e - k * (FIN_PIXELS * 3), r = w / 3, ch = w - 3 * r;
                    if (pix0 + r < numPixels) {
                        const float4 v = s_tile[r][k];
                        pass_out3[(size_t)(k0 + k) * 3 * (size_t)numPixels + 3 * (size_t)(pix0 + r) + (size_t)ch] =
                            ch == 0 ? v.x : (ch == 1 ? v.y : v.z);
                    }
                }
            }
            __syncthreads();
        }
        if (owner) image[i] = a;
    }
}

// Leaf-ordered copy of the triangle records: reference r of the leaf-ordered list goes to the
// 8-byte word offset dst[r] of the packed array (scene set-up, once).
__global__ __launch_bounds__(GPT_BLOCK) void k_gather_records(size_t n_refs, const int32_t* __restrict__ refs,
                                                              const uint32_t* __restrict__ dst, const TriRec* __restrict__ tris,
                                                              unsigned long long* __restrict__ out) {
    const size_t r = (size_t)blockIdx.x * GPT_BLOCK + threadIdx.x;
    if (r >= n_refs) return;
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(tris + refs[r]);
    unsigned long long* o = out + dst[r];
#pragma unroll
    for (int k = 0; k < (int)(sizeof(TriRec) / 8); k++) o[k] = src[k];
}

// One pass of the last batch as packed float3 (what launchKernel hands to its caller).
__global__ __launch_bounds__(GPT_BLOCK) void k_extract_pass(int numPixels, int numPasses, int pass,
                                                            const float4* __restrict__ pass_rgb, float* __restrict__ out3) {
    const int stride = gridDim.x * GPT_BLOCK;
    for (int i = blockIdx.x * GPT_BLOCK + threadIdx.x; i < numPixels; i += stride) {
        const float4 p = pass_rgb[(size_t)i * (size_t)numPasses + (size_t)pass];
        out3[3 * (size_t)i + 0] = p.x;
        out3[3 * (size_t)i + 1] = p.y;
        out3[3 * (size_t)i + 2] = p.z;
    }
}

// The device's random-number stream, for parity checks: out[s * n_draws + k] = k-th uniform of an
// engine seeded with seeds[s] (thrust::default_random_engine + uniform_real_distribution<float>(0,1)
// as the reference uses them, path_tracer_kernel.cu:139-142, 492-493).
__global__ __launch_bounds__(GPT_BLOCK) void k_rng_uniforms(const uint32_t* __restrict__ seeds, int n_seeds, int n_draws,
                                                            float* __restrict__ out) {
    const int s = (int)(blockIdx.x * GPT_BLOCK + threadIdx.x);
    if (s >= n_seeds) return;
    Rng rng;
    rng.seed(seeds[s]);
    for (int k = 0; k < n_draws; k++) out[(size_t)s * (size_t)n_draws + (size_t)k] = rng.uniform();
}

// The per-path helper functions on caller-given records, for parity checks against outputs of the reference's own
// text (record layouts: pt_helper_eval.h).  fn == HELPER_COUNT: the traversal kernel's form of the triangle test
// (triangle_uvt, its two-comparison acceptance, and the hit key having to beat the key of "nothing hit yet", t = PT_T_START)
// on the HELPER_TRIANGLE records: out = (accepted, t bits).
__global__ __launch_bounds__(GPT_BLOCK) void k_debug_helpers(int fn, int n, int words_in, int words_out,
                                                             const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    const int k = (int)(blockIdx.x * GPT_BLOCK + threadIdx.x);
    if (k >= n) return;
    const uint32_t* r = in + (size_t)k * (size_t)words_in;
    uint32_t* o = out + (size_t)k * (size_t)words_out;
    if (fn == HELPER_COUNT) {
        const V3 a = helper_v3(r), b = helper_v3(r + 3), c = helper_v3(r + 6);
        float b1, b2, t;
        triangle_uvt(a, b - a, c - a, helper_v3(r + 9), helper_v3(r + 12), b1, b2, t);
        const float least = vmin(vmin(b1, b2), 1.0f - (b1 + b2));
        o[0] = (least >= 0.0f && t > 0.0f && hit_key(t, 0) < hit_key(PT_T_START, -1)) ? 1u : 0u;
        o[1] = __float_as_uint(t);
        return;
    }
    uint32_t rec[16], res[8];
    for (int w = 0; w < words_in; w++) rec[w] = r[w];
    helper_eval(fn, rec, res);
    for (int w = 0; w < words_out; w++) o[w] = res[w];
}

// ---------------------------------------------------------------------------
// Instrumentation and the scene checksum
// ---------------------------------------------------------------------------
__global__ void k_collect_wave_iters(unsigned long long* stats) {
    if (threadIdx.x < 2) {
        stats[GPT_STAT_WAVE_ITERS_NODE + threadIdx.x] += g_wave_iters[threadIdx.x];
        g_wave_iters[threadIdx.x] = 0;
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

void gptk_camera(hipStream_t s, int grid, const DevGeometry* g, const CamPre* cam, int numPixels, int numPasses,
                 uint32_t seed0, float4* ray_o, float4* ray_d, float4* thr, float2* hit, float4* pass_rgb,
                 unsigned* counts, uint32_t* masks) {
    // masks: numPixels words of scratch for the per-pixel sphere masks, or null (then, or with more than 32 spheres,
    // every primary ray tests every sphere)
    if (masks && (g->n_spheres < 1 || g->n_spheres > 32)) masks = nullptr;
    if (masks)
        hipLaunchKernelGGL(k_camera_masks, dim3((numPixels + GPT_BLOCK - 1) / GPT_BLOCK), dim3(GPT_BLOCK), 0, s, *g, *cam,
                           numPixels, masks);
    hipLaunchKernelGGL(k_camera, dim3(grid), dim3(GPT_BLOCK), 0, s, *g, *cam, numPixels, numPasses, seed0, ray_o, ray_d,
                       thr, hit, pass_rgb, counts, masks);
}

// What is left of gptk_camera when the first shade launch creates the primary rays: the per-pixel sphere masks (if the
// scene's spheres allow them: returns whether it wrote them) and the length of the primary queue.
__global__ void k_set_count(unsigned* counts, unsigned total) { counts[0] = total; }
int gptk_camera_masks_only(hipStream_t s, const DevGeometry* g, const CamPre* cam, int numPixels, int numPasses, unsigned* counts,
                           uint32_t* masks) {
    if (masks && (g->n_spheres < 1 || g->n_spheres > 32)) masks = nullptr;
    if (masks)
        hipLaunchKernelGGL(k_camera_masks, dim3((numPixels + GPT_BLOCK - 1) / GPT_BLOCK), dim3(GPT_BLOCK), 0, s, *g, *cam,
                           numPixels, masks);
    hipLaunchKernelGGL(k_set_count, dim3(1), dim3(1), 0, s, counts, (unsigned)numPixels * (unsigned)numPasses);
    return masks != nullptr;
}

void gptk_intersect(hipStream_t s, int grid, const DevGeometry* g, int use_kd, int stats_on, int depth,
                    const float4* ray_o, const float4* ray_d, float2* hit, unsigned* counts, unsigned* work,
                    unsigned long long* stats, uint32_t* spill, size_t spill_stride, unsigned refill_below,
                    unsigned leaf_min) {
    if (use_kd) {
#define KD_LAUNCH(STATS_)                                                                                               \
    do {                                                                                                                \
        if (g->kd.refs == nullptr)                                                                                      \
            hipLaunchKernelGGL((k_intersect_kd<STATS_, true>), dim3(grid), dim3(GPT_BLOCK), 0, s, *g, depth, ray_o, ray_d, \
                               hit, counts, work, stats, spill, spill_stride, refill_below, leaf_min);                   \
        else                                                                                                            \
            hipLaunchKernelGGL((k_intersect_kd<STATS_, false>), dim3(grid), dim3(GPT_BLOCK), 0, s, *g, depth, ray_o, ray_d, \
                               hit, counts, work, stats, spill, spill_stride, refill_below, leaf_min);                  \
    } while (0)
        if (stats_on)
            KD_LAUNCH(true);
        else
            KD_LAUNCH(false);
#undef KD_LAUNCH
    } else {
        if (stats_on)
            hipLaunchKernelGGL((k_intersect_brute<true>), dim3(grid), dim3(GPT_BLOCK), 0, s, *g, depth, ray_o, ray_d,
                               hit, counts, work, stats);
        else
            hipLaunchKernelGGL((k_intersect_brute<false>), dim3(grid), dim3(GPT_BLOCK), 0, s, *g, depth, ray_o, ray_d,
                               hit, counts, work, stats);
    }
}

// cam (with depth == 0): the launch creates the primary rays itself (k_camera was not launched; counts[0] holds their number),
// with the per-pixel sphere masks `masks` or, without them, testing every sphere
void gptk_shade(hipStream_t s, int grid, const DevGeometry* g, const ShadeParams* sp, const CamPre* cam, const uint32_t* masks,
                int depth, uint32_t seed0, int numPasses, const float4* in_o, const float4* in_d, const float4* in_thr, const float2* hit, float4* out_o,
                float4* out_d, float4* out_thr, float2* out_hit, float4* pass_rgb, unsigned* counts,
                unsigned long long* stats, unsigned char* dbg_events, int* dbg_prims, int dbg_stride) {
#define GPT_LAUNCH_SHADE(FIRST, DEBUG, FUSED)                                                                                \
    hipLaunchKernelGGL((k_shade<FIRST, DEBUG, FUSED>), dim3(blocks), dim3(SHADE_BLOCK), 0, s, *g, *sp, camv, masks, depth, seed0, numPasses, in_o, in_d, \
                       in_thr, hit, out_o, out_d, out_thr, out_hit, pass_rgb, counts, stats, dbg_events, dbg_prims, dbg_stride)
    const bool debug = dbg_events != nullptr && dbg_prims != nullptr;
    CamPre camv = {};
    if (cam) camv = *cam;
    if (masks && (g->n_spheres < 1 || g->n_spheres > 32)) masks = nullptr;
    const int blocks = grid * GPT_BLOCK / SHADE_BLOCK > 0 ? grid * GPT_BLOCK / SHADE_BLOCK : 1;   // `grid` counts 256-thread blocks
    if (depth == 0 && cam) { if (debug) GPT_LAUNCH_SHADE(true, true, true); else GPT_LAUNCH_SHADE(true, false, true); }
    else if (depth == 0) { if (debug) GPT_LAUNCH_SHADE(true, true, false); else GPT_LAUNCH_SHADE(true, false, false); }
    else { if (debug) GPT_LAUNCH_SHADE(false, true, false); else GPT_LAUNCH_SHADE(false, false, false); }
#undef GPT_LAUNCH_SHADE
}

void gptk_finish(hipStream_t s, int grid, int numPixels, int numPasses, const float4* pass_rgb, float4* image,
                 float* pass_out3) {
    (void)grid;   // sized here: one block per 64 pixels, at most 16 per CU's worth
    const int blocks = (numPixels + FIN_PIXELS - 1) / FIN_PIXELS;
    hipLaunchKernelGGL(k_finish, dim3(blocks < 4096 ? blocks : 4096), dim3(GPT_BLOCK), 0, s, numPixels, numPasses, pass_rgb, image,
                       pass_out3);
}

void gptk_extract_pass(hipStream_t s, int grid, int numPixels, int numPasses, int pass, const float4* pass_rgb, float* out3) {
    hipLaunchKernelGGL(k_extract_pass, dim3(grid), dim3(GPT_BLOCK), 0, s, numPixels, numPasses, pass, pass_rgb, out3);
}

void gptk_gather_records(hipStream_t s, size_t n_refs, const int32_t* refs, const uint32_t* dst, const TriRec* tris,
                         unsigned long long* out) {
    if (n_refs == 0) return;
    hipLaunchKernelGGL(k_gather_records, dim3((unsigned)((n_refs + GPT_BLOCK - 1) / GPT_BLOCK)), dim3(GPT_BLOCK), 0, s, n_refs, refs,
                       dst, tris, out);
}

void gptk_rng_uniforms(hipStream_t s, const uint32_t* seeds, int n_seeds, int n_draws, float* out) {
    hipLaunchKernelGGL(k_rng_uniforms, dim3((unsigned)((n_seeds + GPT_BLOCK - 1) / GPT_BLOCK)), dim3(GPT_BLOCK), 0, s, seeds, n_seeds,
                       n_draws, out);
}

int gptk_debug_helpers(hipStream_t s, int fn, int n, const uint32_t* in, uint32_t* out, int* words_in, int* words_out) {
    const int wi = fn == HELPER_COUNT ? helper_words_in(HELPER_TRIANGLE) : helper_words_in(fn);
    const int wo = fn == HELPER_COUNT ? 2 : helper_words_out(fn);
    if (words_in) *words_in = wi;
    if (words_out) *words_out = wo;
    if (wi == 0) return -1;
    if (in && out && n > 0)
        hipLaunchKernelGGL(k_debug_helpers, dim3((unsigned)((n + GPT_BLOCK - 1) / GPT_BLOCK)), dim3(GPT_BLOCK), 0, s, fn, n, wi, wo, in, out);
    return 0;
}

void gptk_collect_wave_iters(hipStream_t s, unsigned long long* stats) {
    hipLaunchKernelGGL(k_collect_wave_iters, dim3(1), dim3(64), 0, s, stats);
}

void gptk_checksum(hipStream_t s, const uint32_t* words, size_t n, unsigned long long* out) {
    int grid = (int)((n + GPT_BLOCK - 1) / GPT_BLOCK);
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k_checksum, dim3(grid), dim3(GPT_BLOCK), 0, s, words, n, out);
}

}  // extern "C"
