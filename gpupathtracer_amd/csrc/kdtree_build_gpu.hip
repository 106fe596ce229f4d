// kdtree_build_gpu.hip -- the kd-tree builder of kdtree_build.cpp on the GPU.
//
// Same rules, same arithmetic, same numbering: the tree that comes out is the host
// builder's word for word (tests compare them), only faster for large meshes.  Like the
// host builder it works breadth-first, one tree level per round:
//   * every queue entry of the level (a node to fill in + its triangles' clipped boxes,
//     stored contiguously in queue order) is decided independently -- large entries:
//     tight box, empty-space cuts, spatial median, straddlers clipped (design source
//     cukd/detail/node_chunk_array_kernels.cu:133-291, kdtree_kernels.cu:196-281); small
//     entries: surface-area heuristic over the box faces (small_node_array_kernels.cu:23-91);
//   * node numbers, leaf reference runs and the next level's entries / items are then
//     handed out in queue order by prefix sums, which is the order the host builder's
//     commit loop uses.
// "First wins" reductions of the host code (std::min / std::max keep the earlier of two
// equal values, which decides the sign of a zero) are reproduced with 64-bit atomic keys
// (value, item position).
#include <hip/hip_runtime.h>
#include <atomic>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "kdtree_build.h"

namespace gptkd {
namespace {

const int kMaxDepthGpu = 56;      // kMaxDepth of kdtree_build.cpp
const int kB = 256;

struct Items {            // SoA, one element per (entry, triangle) pair of a level
    int* tri;
    float* mn[3];
    float* mx[3];
};
struct Entries {
    unsigned* node;
    int* depth;
    int* small;
    float* box;           // 6 per entry: mn xyz, mx xyz
    int* begin;
    int* count;
};
// Per-level decisions.
struct Level {
    unsigned long long* kmin;   // 3 per entry: (ordered value, position) keys of the items' boxes
    unsigned long long* kmax;   // 3 per entry
    float* tb;                  // 6 per entry: tight box (large) or bounds of the items (entering the SAH stage)
    int* kind;                  // 1: large-node stage, 0: SAH stage
    int* ncuts;
    int* cut_axis;              // 6 per entry
    float* cut_pos;             // 6 per entry
    int* cut_empty_right;       // 6 per entry
    int* depth2;                // depth after the cuts
    int* split;
    int* axis;
    float* pos;
    float* cbox;                // 12 per entry: L box, R box
    int* child_small;
    int* nL;
    int* nR;
    // emission counts and their exclusive scans (n_entries + 1 elements)
    int* c_nodes; int* c_refs; int* c_items; int* c_split;
    int* s_nodes; int* s_refs; int* s_items; int* s_split;
};
struct ItemTmp {
    int* ent;                   // entry of the item
    int* cL; int* cR;           // goes to the left / right child
    int* sL; int* sR;           // exclusive scans (n_items + 1 elements)
    float* lb;                  // 6 per item: box in the left child
    float* rb;                  // 6 per item: box in the right child
};

__device__ __forceinline__ unsigned ord_key(float f) {
    unsigned u = __float_as_uint(f);
    if ((u << 1) == 0u) u = 0u;                       // -0 and +0 compare equal
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
// std::min / std::max of <algorithm>: the first argument stays unless the second is strictly better.
__device__ __forceinline__ float min_keep(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float max_keep(float a, float b) { return (a < b) ? b : a; }
__device__ __forceinline__ double dmin_keep(double a, double b) { return (b < a) ? b : a; }
__device__ __forceinline__ double dmax_keep(double a, double b) { return (a < b) ? b : a; }

__device__ __forceinline__ float box_area(const float mn[3], const float mx[3]) {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.f * (dx * dy + dx * dz + dy * dz);
}

__device__ __forceinline__ float next_down(float f) {     // nextafterf(f, -inf) for finite f
    unsigned u = __float_as_uint(f);
    if ((u << 1) == 0u) return __uint_as_float(0x80000001u);
    return __uint_as_float((u & 0x80000000u) ? u + 1u : u - 1u);
}
__device__ __forceinline__ float next_up(float f) {       // nextafterf(f, +inf) for finite f
    unsigned u = __float_as_uint(f);
    if ((u << 1) == 0u) return __uint_as_float(0x00000001u);
    return __uint_as_float((u & 0x80000000u) ? u - 1u : u + 1u);
}

// clipped_bounds of kdtree_build.cpp: Sutherland-Hodgman in double against the box padded by
// `pad`, bounds rounded outward to float.
__device__ bool clipped_bounds(const float* v9, const float bmn[3], const float bmx[3], double pad, float out_mn[3],
                               float out_mx[3]) {
    double poly[16][3], tmp[16][3];
    int n = 3;
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) poly[i][k] = v9[3 * i + k];
    for (int axis = 0; axis < 3 && n > 0; axis++) {
        for (int side = 0; side < 2 && n > 0; side++) {
            const double pos = side == 0 ? (double)bmn[axis] - pad : (double)bmx[axis] + pad;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const double* a = poly[i];
                const double* c = poly[(i + 1) % n];
                const bool ain = side == 0 ? (a[axis] >= pos) : (a[axis] <= pos);
                const bool cin = side == 0 ? (c[axis] >= pos) : (c[axis] <= pos);
                if (ain) { tmp[m][0] = a[0]; tmp[m][1] = a[1]; tmp[m][2] = a[2]; m++; }
                if (ain != cin) {
                    const double t = (pos - a[axis]) / (c[axis] - a[axis]);
                    for (int k = 0; k < 3; k++) tmp[m][k] = a[k] + (c[k] - a[k]) * t;
                    tmp[m][axis] = pos;
                    m++;
                }
            }
            n = m;
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) poly[i][k] = tmp[i][k];
        }
    }
    if (n == 0) return false;
    for (int k = 0; k < 3; k++) {
        double lo = poly[0][k], hi = poly[0][k];
        for (int i = 1; i < n; i++) { lo = dmin_keep(lo, poly[i][k]); hi = dmax_keep(hi, poly[i][k]); }
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo > lo) flo = next_down(flo);
        if ((double)fhi < hi) fhi = next_up(fhi);
        out_mn[k] = flo;
        out_mx[k] = fhi;
    }
    return true;
}

// ---- wave helpers ------------------------------------------------------------
__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m) {
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = (unsigned)__shfl_xor((int)lo, m, 64);
    hi = (unsigned)__shfl_xor((int)hi, m, 64);
    return ((unsigned long long)hi << 32) | lo;
}

// Folds one (min key, max key) pair per lane into target[0] / target[1]; when the whole wave
// feeds the same target the keys are reduced in registers first (one atomic pair per wave).
__device__ __forceinline__ void fold_keys(bool active, bool uniform, unsigned long long kmin, unsigned long long kmax,
                                          unsigned long long* tmin, unsigned long long* tmax) {
    if (uniform) {
        if (!active) { kmin = ~0ull; kmax = 0ull; }
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long a = shfl_xor_u64(kmin, m), b = shfl_xor_u64(kmax, m);
            kmin = a < kmin ? a : kmin;
            kmax = b > kmax ? b : kmax;
        }
        if ((threadIdx.x & 63u) == 0u) { atomicMin(tmin, kmin); atomicMax(tmax, kmax); }
    } else if (active) {
        atomicMin(tmin, kmin);
        atomicMax(tmax, kmax);
    }
}

// ---- kernels -----------------------------------------------------------------
__global__ void k_init_items(const float* __restrict__ verts, int n, Items it, unsigned long long* root_keys) {
    const int i = blockIdx.x * kB + threadIdx.x;
    const bool active = i < n;
    float mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    if (active) {
        const float* v = verts + 9 * (size_t)i;
        it.tri[i] = i;
        for (int k = 0; k < 3; k++) {
            mn[k] = min_keep(v[k], min_keep(v[3 + k], v[6 + k]));
            mx[k] = max_keep(v[k], max_keep(v[3 + k], v[6 + k]));
            it.mn[k][i] = mn[k];
            it.mx[k][i] = mx[k];
        }
    }
    for (int k = 0; k < 3; k++)
        fold_keys(active, true, ((unsigned long long)ord_key(mn[k]) << 32) | (unsigned)i,
                  ((unsigned long long)ord_key(mx[k]) << 32) | (0xffffffffu - (unsigned)i), &root_keys[k], &root_keys[3 + k]);
}

__global__ void k_resolve_root(const unsigned long long* keys, Items it, float* box) {
    const int k = threadIdx.x;
    if (k < 3) {
        box[k] = it.mn[k][(unsigned)keys[k]];
        box[3 + k] = it.mx[k][0xffffffffu - (unsigned)keys[3 + k]];
    }
}

// entry of every item: the last entry whose `begin` is <= the item's position (entries with
// no items share their begin with the next one and are skipped by taking the LAST such entry
// that still has count > 0 -- found by searching on begin + count instead)
__global__ void k_item_entry(int n_items, int n_entries, const int* __restrict__ begin, const int* __restrict__ count,
                             int* __restrict__ ent) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n_items) return;
    int lo = 0, hi = n_entries - 1;          // first entry with begin + count > i
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (begin[mid] + count[mid] > i) hi = mid; else lo = mid + 1;
    }
    ent[i] = lo;
}

__global__ void k_level_init(int n_entries, Level lv) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e >= n_entries) return;
    for (int k = 0; k < 3; k++) { lv.kmin[3 * e + k] = ~0ull; lv.kmax[3 * e + k] = 0ull; }
}

// bounds of the items of every entry that has not reached the SAH stage yet
__global__ void k_bounds(int n_items, Items it, const int* __restrict__ ent, Entries en, Level lv) {
    const int i = blockIdx.x * kB + threadIdx.x;
    const bool in = i < n_items;
    const int e = in ? ent[i] : -1;
    const int e0 = __shfl(e, 0, 64);
    const bool uniform = __all(e == e0 || !in) && e0 >= 0;
    const bool active = in && en.small[e] == 0;
    if (uniform && !__any(active)) return;
    const int et = uniform ? e0 : (e >= 0 ? e : 0);
    const unsigned pos = in ? (unsigned)(i - en.begin[et >= 0 ? et : 0]) : 0u;
    for (int k = 0; k < 3; k++) {
        const float a = in ? it.mn[k][i] : 0.0f, b = in ? it.mx[k][i] : 0.0f;
        fold_keys(active, uniform, ((unsigned long long)ord_key(a) << 32) | pos,
                  ((unsigned long long)ord_key(b) << 32) | (0xffffffffu - pos), &lv.kmin[3 * et + k], &lv.kmax[3 * et + k]);
    }
}

// large-node stage, per entry: tight box, empty-space cuts, spatial median
__global__ void k_decide_large(int n_entries, Entries en, Items it, Level lv, BuildParams bp) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e >= n_entries) return;
    const int n = en.count[e];
    lv.ncuts[e] = 0;
    lv.split[e] = 0;
    lv.kind[e] = 0;
    lv.depth2[e] = en.depth[e];
    lv.nL[e] = lv.nR[e] = 0;
    if (n == 0) return;
    const float* box = en.box + 6 * (size_t)e;
    float* tb = lv.tb + 6 * (size_t)e;
    if (en.small[e] == 0) {
        const int b = en.begin[e];
        for (int k = 0; k < 3; k++) {
            tb[k] = it.mn[k][b + (int)(unsigned)lv.kmin[3 * e + k]];
            tb[3 + k] = it.mx[k][b + (int)(0xffffffffu - (unsigned)lv.kmax[3 * e + k])];
        }
    }
    if (!(en.small[e] == 0 && n > bp.small_threshold && en.depth[e] < kMaxDepthGpu - 8)) return;
    lv.kind[e] = 1;
    int nc = 0, depth = en.depth[e];
    for (int a = 0; a < 3; a++) {
        const float thr = bp.empty_ratio * (box[3 + a] - box[a]);
        if ((box[3 + a] - tb[3 + a]) - thr > 0) {
            lv.cut_axis[6 * e + nc] = a; lv.cut_pos[6 * e + nc] = tb[3 + a]; lv.cut_empty_right[6 * e + nc] = 1; nc++; depth++;
        }
        if ((tb[a] - box[a]) - thr > 0) {
            lv.cut_axis[6 * e + nc] = a; lv.cut_pos[6 * e + nc] = tb[a]; lv.cut_empty_right[6 * e + nc] = 0; nc++; depth++;
        }
    }
    lv.ncuts[e] = nc;
    lv.depth2[e] = depth;
    const float dx = tb[3] - tb[0], dy = tb[4] - tb[1], dz = tb[5] - tb[2];
    int axis;
    float pos;
    if (dx > dy) {
        if (dx > dz) { axis = 0; pos = 0.5f * dx + tb[0]; }
        else { axis = 2; pos = 0.5f * dz + tb[2]; }
    } else {
        if (dy > dz) { axis = 1; pos = 0.5f * dy + tb[1]; }
        else { axis = 2; pos = 0.5f * dz + tb[2]; }
    }
    lv.axis[e] = axis;
    lv.pos[e] = pos;
    float* cb = lv.cbox + 12 * (size_t)e;
    for (int k = 0; k < 6; k++) { cb[k] = tb[k]; cb[6 + k] = tb[k]; }
    cb[3 + axis] = pos;        // L.box.mx[axis]
    cb[6 + axis] = pos;        // R.box.mn[axis]
}

// large-node stage, per item: side(s) of the median, clipped boxes of straddlers
__global__ void k_classify(int n_items, Items it, ItemTmp tm, Level lv, const float* __restrict__ verts, double clip_pad) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n_items) return;
    const int e = tm.ent[i];
    int cl = 0, cr = 0;
    float mn[3], mx[3];
    for (int k = 0; k < 3; k++) { mn[k] = it.mn[k][i]; mx[k] = it.mx[k][i]; }
    float* lb = tm.lb + 6 * (size_t)i;
    float* rb = tm.rb + 6 * (size_t)i;
    for (int k = 0; k < 3; k++) { lb[k] = rb[k] = mn[k]; lb[3 + k] = rb[3 + k] = mx[k]; }
    if (lv.kind[e] == 1) {
        const int axis = lv.axis[e];
        const float pos = lv.pos[e];
        const bool toL = (mn[axis] < pos) || (mx[axis] <= pos);
        const bool toR = (mx[axis] > pos) || (mn[axis] >= pos);
        if (toL && toR) {
            const float* v = verts + 9 * (size_t)it.tri[i];
            const float* cb = lv.cbox + 12 * (size_t)e;
            for (int side = 0; side < 2; side++) {
                const float* bmn = cb + 6 * side;
                const float* bmx = bmn + 3;
                float cmn[3], cmx[3];
                if (!clipped_bounds(v, bmn, bmx, clip_pad, cmn, cmx)) continue;
                bool empty = false;
                float* out = side == 0 ? lb : rb;
                for (int k = 0; k < 3; k++) {
                    const float a = max_keep(mn[k], max_keep(cmn[k], bmn[k]));
                    const float b = min_keep(mx[k], min_keep(cmx[k], bmx[k]));
                    out[k] = a;
                    out[3 + k] = b;
                    if (a > b) empty = true;
                }
                if (empty) continue;
                if (side == 0) cl = 1; else cr = 1;
            }
        } else if (toL) {
            cl = 1;
        } else {
            cr = 1;
        }
    }
    tm.cL[i] = cl;
    tm.cR[i] = cr;
}

// large entries whose median separates nothing fall through to the SAH stage with their tight box
__global__ void k_post_large(int n_entries, Entries en, Level lv, const int* __restrict__ sL, const int* __restrict__ sR) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e >= n_entries) return;
    if (lv.kind[e] != 1) return;
    const int b = en.begin[e], n = en.count[e];
    const int nl = sL[b + n] - sL[b], nr = sR[b + n] - sR[b];
    if (!(nl == n && nr == n)) {
        lv.split[e] = 1;
        lv.child_small[e] = 0;
        return;
    }
    lv.kind[e] = 2;      // SAH stage on the tight box, as an entry that is already "small"
}

// SAH stage: one wave per entry
__global__ void k_sah(int n_entries, Entries en, Items it, ItemTmp tm, Level lv, BuildParams bp) {
    const int e = blockIdx.x * (kB / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (e >= n_entries) return;
    const int kind = lv.kind[e];
    const int n = en.count[e];
    if (kind == 1 || n == 0) return;
    const int b = en.begin[e];
    float nmn[3], nmx[3];
    const float* box = en.box + 6 * (size_t)e;
    const float* tb = lv.tb + 6 * (size_t)e;
    for (int k = 0; k < 3; k++) {
        if (kind == 2) { nmn[k] = tb[k]; nmx[k] = tb[3 + k]; }
        else if (en.small[e] == 0) { nmn[k] = max_keep(box[k], tb[k]); nmx[k] = min_keep(box[3 + k], tb[3 + k]); }
        else { nmn[k] = box[k]; nmx[k] = box[3 + k]; }
    }
    const int depth = lv.depth2[e];
    const float area = box_area(nmn, nmx);
    int best_c = 0x7fffffff;
    float best_cost = INFINITY, best_pos = 0.0f;
    if (area > 0 && depth < kMaxDepthGpu) {
        const float inv_area = 1.f / area;
        for (int c = lane; c < 6 * n; c += 64) {
            const int a = c / (2 * n), side = (c / n) & 1, j = c % n;
            const float pos = side == 0 ? it.mn[a][b + j] : it.mx[a][b + j];
            if (!(pos > nmn[a] && pos < nmx[a])) continue;
            int nl = 0, nr = 0;
            for (int k = 0; k < n; k++) {
                nl += (it.mn[a][b + k] < pos) ? 1 : 0;
                nr += (it.mx[a][b + k] >= pos) ? 1 : 0;
            }
            if (nl == 0 || nr == 0) continue;
            float lmx[3] = {nmx[0], nmx[1], nmx[2]};
            float rmn[3] = {nmn[0], nmn[1], nmn[2]};
            lmx[a] = pos;
            rmn[a] = pos;
            const float al = box_area(nmn, lmx);
            const float ar = box_area(rmn, nmx);
            const float cost = inv_area * ((float)nl * al + (float)nr * ar) + bp.traversal_cost;
            if (cost < best_cost) { best_cost = cost; best_c = c; best_pos = pos; }    // c ascends per lane: first wins
        }
    }
    // smallest cost, ties to the lowest candidate number
    for (int m = 32; m >= 1; m >>= 1) {
        const float oc = __shfl_xor(best_cost, m, 64);
        const int oi = __shfl_xor(best_c, m, 64);
        const float op = __shfl_xor(best_pos, m, 64);
        if (oc < best_cost || (oc == best_cost && oi < best_c)) { best_cost = oc; best_c = oi; best_pos = op; }
    }
    const bool split = best_c != 0x7fffffff && !(best_cost >= (float)n);
    const int axis = split ? best_c / (2 * n) : 0;
    if (lane == 0) {
        lv.split[e] = split ? 1 : 0;
        if (split) {
            lv.axis[e] = axis;
            lv.pos[e] = best_pos;
            lv.child_small[e] = 1;
            float* cb = lv.cbox + 12 * (size_t)e;
            for (int k = 0; k < 3; k++) { cb[k] = nmn[k]; cb[3 + k] = nmx[k]; cb[6 + k] = nmn[k]; cb[9 + k] = nmx[k]; }
            cb[3 + axis] = best_pos;
            cb[6 + axis] = best_pos;
        }
    }
    for (int k = lane; k < n; k += 64) {
        int cl = 0, cr = 0;
        if (split) {
            cl = it.mn[axis][b + k] < best_pos ? 1 : 0;
            cr = it.mx[axis][b + k] >= best_pos ? 1 : 0;
        }
        tm.cL[b + k] = cl;
        tm.cR[b + k] = cr;
        if (kind == 2) {      // the boxes k_classify clipped for the median that was not taken
            float* lb = tm.lb + 6 * (size_t)(b + k);
            float* rb = tm.rb + 6 * (size_t)(b + k);
            for (int q = 0; q < 3; q++) { lb[q] = rb[q] = it.mn[q][b + k]; lb[3 + q] = rb[3 + q] = it.mx[q][b + k]; }
        }
    }
}

__global__ void k_emit_counts(int n_entries, Entries en, Level lv, const int* __restrict__ sL, const int* __restrict__ sR) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e > n_entries) return;
    if (e == n_entries) { lv.c_nodes[e] = lv.c_refs[e] = lv.c_items[e] = lv.c_split[e] = 0; return; }
    const int b = en.begin[e], n = en.count[e];
    const int split = lv.split[e];
    const int nl = split ? sL[b + n] - sL[b] : 0, nr = split ? sR[b + n] - sR[b] : 0;
    lv.nL[e] = nl;
    lv.nR[e] = nr;
    lv.c_nodes[e] = 2 * lv.ncuts[e] + (split ? 2 : 0);
    lv.c_refs[e] = split ? 0 : n;
    lv.c_items[e] = nl + nr;
    lv.c_split[e] = split;
}

// nodes of the level and the entries of the next one, numbered in queue order
__global__ void k_commit(int n_entries, Entries en, Level lv, Entries nx, Node* __restrict__ nodes, unsigned node_base,
                         unsigned ref_base, int* __restrict__ max_depth, int* __restrict__ leaf_counts) {
    const int e = blockIdx.x * kB + threadIdx.x;
    if (e >= n_entries) return;
    unsigned node = en.node[e];
    int depth = en.depth[e];
    unsigned next_node = node_base + (unsigned)lv.s_nodes[e];
    const unsigned refs_here = ref_base + (unsigned)lv.s_refs[e];
    int local_max = 0, leaves = 0, empty_leaves = 0;
    for (int c = 0; c < lv.ncuts[e]; c++) {
        const unsigned left = next_node;
        next_node += 2;
        nodes[node].w0 = (left << 2) | (unsigned)lv.cut_axis[6 * e + c];
        nodes[node].w1 = __float_as_uint(lv.cut_pos[6 * e + c]);
        const unsigned empty = lv.cut_empty_right[6 * e + c] ? left + 1 : left;
        nodes[empty].w0 = (refs_here << 2) | 3u;
        nodes[empty].w1 = 0u;
        node = lv.cut_empty_right[6 * e + c] ? left : left + 1;
        depth++;
        local_max = depth > local_max ? depth : local_max;
        leaves++;
        empty_leaves++;
    }
    if (!lv.split[e]) {
        nodes[node].w0 = (refs_here << 2) | 3u;
        nodes[node].w1 = (unsigned)en.count[e];
        local_max = depth > local_max ? depth : local_max;
        leaves++;
        if (en.count[e] == 0) empty_leaves++;
    } else {
        const unsigned left = next_node;
        nodes[node].w0 = (left << 2) | (unsigned)lv.axis[e];
        nodes[node].w1 = __float_as_uint(lv.pos[e]);
        const int q = 2 * lv.s_split[e];
        const int ib = lv.s_items[e];
        for (int side = 0; side < 2; side++) {
            nx.node[q + side] = left + (unsigned)side;
            nx.depth[q + side] = depth + 1;
            nx.small[q + side] = lv.child_small[e];
            for (int k = 0; k < 6; k++) nx.box[6 * (size_t)(q + side) + k] = lv.cbox[12 * (size_t)e + 6 * side + k];
            nx.begin[q + side] = side == 0 ? ib : ib + lv.nL[e];
            nx.count[q + side] = side == 0 ? lv.nL[e] : lv.nR[e];
        }
    }
    if (leaves) {
        atomicMax(max_depth, local_max);
        atomicAdd(&leaf_counts[0], leaves);
        if (empty_leaves) atomicAdd(&leaf_counts[1], empty_leaves);
    }
}

// leaf entries: triangle references; split entries: the items of their children
__global__ void k_scatter(int n_items, Items it, ItemTmp tm, Entries en, Level lv, Items nx, int* __restrict__ refs,
                          unsigned ref_base) {
    const int i = blockIdx.x * kB + threadIdx.x;
    if (i >= n_items) return;
    const int e = tm.ent[i];
    const int b = en.begin[e];
    if (!lv.split[e]) {
        refs[ref_base + (unsigned)lv.s_refs[e] + (unsigned)(i - b)] = it.tri[i];
        return;
    }
    const int ib = lv.s_items[e];
    if (tm.cL[i]) {
        const int d = ib + (tm.sL[i] - tm.sL[b]);
        nx.tri[d] = it.tri[i];
        for (int k = 0; k < 3; k++) { nx.mn[k][d] = tm.lb[6 * (size_t)i + k]; nx.mx[k][d] = tm.lb[6 * (size_t)i + 3 + k]; }
    }
    if (tm.cR[i]) {
        const int d = ib + lv.nL[e] + (tm.sR[i] - tm.sR[b]);
        nx.tri[d] = it.tri[i];
        for (int k = 0; k < 3; k++) { nx.mn[k][d] = tm.rb[6 * (size_t)i + k]; nx.mx[k][d] = tm.rb[6 * (size_t)i + 3 + k]; }
    }
}

__global__ void k_set_last(int* a, int idx, int v) { a[idx] = v; }

// ---- host side ---------------------------------------------------------------
struct DevPool {
    std::vector<void*> ptrs;
    bool ok = true;
    std::string err;
    template <class T>
    T* get(size_t n) {
        void* p = nullptr;
        if (!ok) return nullptr;
        hipError_t e = hipMalloc(&p, (n ? n : 1) * sizeof(T));
        if (e != hipSuccess) { ok = false; err = hipGetErrorString(e); return nullptr; }
        ptrs.push_back(p);
        return (T*)p;
    }
    void release(void* p) {
        for (auto& q : ptrs) if (q == p) { (void)hipFree(q); q = nullptr; }
    }
    ~DevPool() { for (void* p : ptrs) if (p) (void)hipFree(p); }
};

Items alloc_items(DevPool& pool, size_t cap) {
    Items it;
    it.tri = pool.get<int>(cap);
    for (int k = 0; k < 3; k++) { it.mn[k] = pool.get<float>(cap); it.mx[k] = pool.get<float>(cap); }
    return it;
}
void free_items(DevPool& pool, Items& it) {
    pool.release(it.tri);
    for (int k = 0; k < 3; k++) { pool.release(it.mn[k]); pool.release(it.mx[k]); }
}
Entries alloc_entries(DevPool& pool, size_t cap) {
    Entries en;
    en.node = pool.get<unsigned>(cap); en.depth = pool.get<int>(cap); en.small = pool.get<int>(cap);
    en.box = pool.get<float>(6 * cap); en.begin = pool.get<int>(cap); en.count = pool.get<int>(cap);
    return en;
}
void free_entries(DevPool& pool, Entries& en) {
    pool.release(en.node); pool.release(en.depth); pool.release(en.small); pool.release(en.box); pool.release(en.begin);
    pool.release(en.count);
}
Level alloc_level(DevPool& pool, size_t cap) {
    Level lv;
    lv.kmin = pool.get<unsigned long long>(3 * cap); lv.kmax = pool.get<unsigned long long>(3 * cap);
    lv.tb = pool.get<float>(6 * cap); lv.kind = pool.get<int>(cap); lv.ncuts = pool.get<int>(cap);
    lv.cut_axis = pool.get<int>(6 * cap); lv.cut_pos = pool.get<float>(6 * cap); lv.cut_empty_right = pool.get<int>(6 * cap);
    lv.depth2 = pool.get<int>(cap); lv.split = pool.get<int>(cap); lv.axis = pool.get<int>(cap); lv.pos = pool.get<float>(cap);
    lv.cbox = pool.get<float>(12 * cap); lv.child_small = pool.get<int>(cap); lv.nL = pool.get<int>(cap); lv.nR = pool.get<int>(cap);
    lv.c_nodes = pool.get<int>(cap + 1); lv.c_refs = pool.get<int>(cap + 1); lv.c_items = pool.get<int>(cap + 1);
    lv.c_split = pool.get<int>(cap + 1);
    lv.s_nodes = pool.get<int>(cap + 1); lv.s_refs = pool.get<int>(cap + 1); lv.s_items = pool.get<int>(cap + 1);
    lv.s_split = pool.get<int>(cap + 1);
    return lv;
}
void free_level(DevPool& pool, Level& lv) {
    void* all[] = {lv.kmin, lv.kmax, lv.tb, lv.kind, lv.ncuts, lv.cut_axis, lv.cut_pos, lv.cut_empty_right, lv.depth2, lv.split,
                   lv.axis, lv.pos, lv.cbox, lv.child_small, lv.nL, lv.nR, lv.c_nodes, lv.c_refs, lv.c_items, lv.c_split,
                   lv.s_nodes, lv.s_refs, lv.s_items, lv.s_split};
    for (void* p : all) pool.release(p);
}
ItemTmp alloc_tmp(DevPool& pool, size_t cap) {
    ItemTmp tm;
    tm.ent = pool.get<int>(cap); tm.cL = pool.get<int>(cap + 1); tm.cR = pool.get<int>(cap + 1);
    tm.sL = pool.get<int>(cap + 1); tm.sR = pool.get<int>(cap + 1);
    tm.lb = pool.get<float>(6 * cap); tm.rb = pool.get<float>(6 * cap);
    return tm;
}
void free_tmp(DevPool& pool, ItemTmp& tm) {
    pool.release(tm.ent); pool.release(tm.cL); pool.release(tm.cR); pool.release(tm.sL); pool.release(tm.sR);
    pool.release(tm.lb); pool.release(tm.rb);
}

inline int blocks(size_t n) { return (int)((n + kB - 1) / kB); }

#define GK_TRY(call)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) { if (err) *err = std::string("kd build on the GPU: ") + hipGetErrorString(e_); return false; } \
    } while (0)

}  // namespace

// The host builder's tree, built on the current HIP device.  false: a HIP call failed (message
// in *err); the caller may fall back to build().
bool build_gpu(const float* verts, int n_tris, const BuildParams& bp, Tree& out, std::string* err) {
    if (n_tris <= 0) { build(verts, n_tris, bp, out); return true; }
    // Fault injection for the tests of the callers' fallback to the host builder: GPT_TEST_FAIL_GPU_BUILD=k makes the
    // k-th call of this function in the process fail the way an exhausted device would.
    if (const char* e = getenv("GPT_TEST_FAIL_GPU_BUILD")) {
        static std::atomic<int> calls{0};
        if (++calls == atoi(e)) {
            if (err) *err = "kd-tree build on the GPU failed (injected by GPT_TEST_FAIL_GPU_BUILD)";
            return false;
        }
    }
    DevPool pool;
    float* d_verts = pool.get<float>((size_t)n_tris * 9);
    unsigned long long* d_root_keys = pool.get<unsigned long long>(6);
    float* d_root_box = pool.get<float>(6);
    int* d_counters = pool.get<int>(4);        // max depth, leaves, empty leaves
    // capacities grow on demand, each array set on its own
    size_t cap_it = (size_t)n_tris + 16, cap_nx = (size_t)n_tris * 2 + 1024, cap_tm = cap_it;
    size_t ecap_en = 1 << 16, ecap_nx = 1 << 16, ecap_lv = 1 << 16;
    Items it = alloc_items(pool, cap_it), nx_it = alloc_items(pool, cap_nx);
    Entries en = alloc_entries(pool, ecap_en), nx_en = alloc_entries(pool, ecap_nx);
    Level lv = alloc_level(pool, ecap_lv);
    ItemTmp tm = alloc_tmp(pool, cap_tm);
    size_t node_cap = (size_t)n_tris * 2 + 4096, ref_cap = (size_t)n_tris * 3 + 4096;
    Node* d_nodes = pool.get<Node>(node_cap);
    int* d_refs = pool.get<int>(ref_cap);
    // scan scratch
    size_t scan_bytes = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, 1 << 30);
    void* d_scan = pool.get<char>(scan_bytes);
    if (!pool.ok) { if (err) *err = "kd build on the GPU: " + pool.err; return false; }

    GK_TRY(hipMemcpy(d_verts, verts, (size_t)n_tris * 9 * sizeof(float), hipMemcpyHostToDevice));
    {
        unsigned long long init[6] = {~0ull, ~0ull, ~0ull, 0ull, 0ull, 0ull};
        GK_TRY(hipMemcpy(d_root_keys, init, sizeof(init), hipMemcpyHostToDevice));
        GK_TRY(hipMemset(d_counters, 0, 4 * sizeof(int)));
    }
    hipLaunchKernelGGL(k_init_items, dim3(blocks((size_t)n_tris)), dim3(kB), 0, 0, d_verts, n_tris, it, d_root_keys);
    hipLaunchKernelGGL(k_resolve_root, dim3(1), dim3(64), 0, 0, d_root_keys, it, d_root_box);
    float root_box[6];
    GK_TRY(hipMemcpy(root_box, d_root_box, sizeof(root_box), hipMemcpyDeviceToHost));
    for (int k = 0; k < 6; k++) out.bounds[k] = root_box[k];
    float ext = 0;
    for (int k = 0; k < 3; k++) ext = std::max(ext, root_box[3 + k] - root_box[k]);
    const double clip_pad = 1e-5 * (double)ext;

    // queue entry 0 = the root
    {
        unsigned node0 = 0; int zero = 0, n = n_tris;
        GK_TRY(hipMemcpy(en.node, &node0, 4, hipMemcpyHostToDevice));
        GK_TRY(hipMemcpy(en.depth, &zero, 4, hipMemcpyHostToDevice));
        GK_TRY(hipMemcpy(en.small, &zero, 4, hipMemcpyHostToDevice));
        GK_TRY(hipMemcpy(en.box, root_box, sizeof(root_box), hipMemcpyHostToDevice));
        GK_TRY(hipMemcpy(en.begin, &zero, 4, hipMemcpyHostToDevice));
        GK_TRY(hipMemcpy(en.count, &n, 4, hipMemcpyHostToDevice));
    }
    size_t n_entries = 1, n_items = (size_t)n_tris;
    unsigned n_nodes = 1, n_refs = 0;

    auto scan = [&](const int* in, int* outp, size_t n) -> hipError_t {
        size_t bytes = scan_bytes;
        return hipcub::DeviceScan::ExclusiveSum(d_scan, bytes, in, outp, (int)n);
    };

    while (n_entries > 0) {
        hipLaunchKernelGGL(k_level_init, dim3(blocks(n_entries)), dim3(kB), 0, 0, (int)n_entries, lv);
        if (n_items > 0) {
            hipLaunchKernelGGL(k_item_entry, dim3(blocks(n_items)), dim3(kB), 0, 0, (int)n_items, (int)n_entries, en.begin, en.count,
                               tm.ent);
            hipLaunchKernelGGL(k_bounds, dim3(blocks(n_items)), dim3(kB), 0, 0, (int)n_items, it, tm.ent, en, lv);
        }
        hipLaunchKernelGGL(k_decide_large, dim3(blocks(n_entries)), dim3(kB), 0, 0, (int)n_entries, en, it, lv, bp);
        if (n_items > 0) {
            hipLaunchKernelGGL(k_classify, dim3(blocks(n_items)), dim3(kB), 0, 0, (int)n_items, it, tm, lv, d_verts, clip_pad);
            hipLaunchKernelGGL(k_set_last, dim3(1), dim3(1), 0, 0, tm.cL, (int)n_items, 0);
            hipLaunchKernelGGL(k_set_last, dim3(1), dim3(1), 0, 0, tm.cR, (int)n_items, 0);
            GK_TRY(scan(tm.cL, tm.sL, n_items + 1));
            GK_TRY(scan(tm.cR, tm.sR, n_items + 1));
        }
        hipLaunchKernelGGL(k_post_large, dim3(blocks(n_entries)), dim3(kB), 0, 0, (int)n_entries, en, lv, tm.sL, tm.sR);
        hipLaunchKernelGGL(k_sah, dim3((unsigned)((n_entries + kB / 64 - 1) / (kB / 64))), dim3(kB), 0, 0, (int)n_entries, en, it, tm,
                           lv, bp);
        if (n_items > 0) {
            GK_TRY(scan(tm.cL, tm.sL, n_items + 1));
            GK_TRY(scan(tm.cR, tm.sR, n_items + 1));
        }
        hipLaunchKernelGGL(k_emit_counts, dim3(blocks(n_entries + 1)), dim3(kB), 0, 0, (int)n_entries, en, lv, tm.sL, tm.sR);
        GK_TRY(scan(lv.c_nodes, lv.s_nodes, n_entries + 1));
        GK_TRY(scan(lv.c_refs, lv.s_refs, n_entries + 1));
        GK_TRY(scan(lv.c_items, lv.s_items, n_entries + 1));
        GK_TRY(scan(lv.c_split, lv.s_split, n_entries + 1));
        int totals[4];
        GK_TRY(hipMemcpy(&totals[0], lv.s_nodes + n_entries, 4, hipMemcpyDeviceToHost));
        GK_TRY(hipMemcpy(&totals[1], lv.s_refs + n_entries, 4, hipMemcpyDeviceToHost));
        GK_TRY(hipMemcpy(&totals[2], lv.s_items + n_entries, 4, hipMemcpyDeviceToHost));
        GK_TRY(hipMemcpy(&totals[3], lv.s_split + n_entries, 4, hipMemcpyDeviceToHost));
        const size_t next_entries = 2 * (size_t)totals[3], next_items = (size_t)totals[2];
        // room for what this level emits
        if ((size_t)n_nodes + (size_t)totals[0] > node_cap) {
            const size_t cap2 = std::max(node_cap * 2, (size_t)n_nodes + (size_t)totals[0]);
            Node* p = pool.get<Node>(cap2);
            if (!pool.ok) { if (err) *err = "kd build on the GPU: " + pool.err; return false; }
            GK_TRY(hipMemcpy(p, d_nodes, (size_t)n_nodes * sizeof(Node), hipMemcpyDeviceToDevice));
            pool.release(d_nodes);
            d_nodes = p;
            node_cap = cap2;
        }
        if ((size_t)n_refs + (size_t)totals[1] > ref_cap) {
            const size_t cap2 = std::max(ref_cap * 2, (size_t)n_refs + (size_t)totals[1]);
            int* p = pool.get<int>(cap2);
            if (!pool.ok) { if (err) *err = "kd build on the GPU: " + pool.err; return false; }
            GK_TRY(hipMemcpy(p, d_refs, (size_t)n_refs * sizeof(int), hipMemcpyDeviceToDevice));
            pool.release(d_refs);
            d_refs = p;
            ref_cap = cap2;
        }
        if (next_items > cap_nx) {
            free_items(pool, nx_it);
            cap_nx = std::max(cap_nx * 2, next_items);
            nx_it = alloc_items(pool, cap_nx);
        }
        if (next_entries > ecap_nx) {
            free_entries(pool, nx_en);
            ecap_nx = std::max(ecap_nx * 2, next_entries);
            nx_en = alloc_entries(pool, ecap_nx);
        }
        if (!pool.ok) { if (err) *err = "kd build on the GPU: " + pool.err; return false; }
        hipLaunchKernelGGL(k_commit, dim3(blocks(n_entries)), dim3(kB), 0, 0, (int)n_entries, en, lv, nx_en, d_nodes, n_nodes, n_refs,
                           &d_counters[0], &d_counters[1]);
        if (n_items > 0)
            hipLaunchKernelGGL(k_scatter, dim3(blocks(n_items)), dim3(kB), 0, 0, (int)n_items, it, tm, en, lv, nx_it, d_refs, n_refs);
        GK_TRY(hipGetLastError());
        GK_TRY(hipDeviceSynchronize());
        n_nodes += (unsigned)totals[0];
        n_refs += (unsigned)totals[1];
        // next level
        std::swap(it, nx_it);
        std::swap(cap_it, cap_nx);
        std::swap(en, nx_en);
        std::swap(ecap_en, ecap_nx);
        n_entries = next_entries;
        n_items = next_items;
        if (n_items > cap_tm) {
            free_tmp(pool, tm);
            cap_tm = std::max(cap_tm * 2, n_items);
            tm = alloc_tmp(pool, cap_tm);
        }
        if (n_entries > ecap_lv) {
            free_level(pool, lv);
            ecap_lv = std::max(ecap_lv * 2, n_entries);
            lv = alloc_level(pool, ecap_lv);
        }
        if (!pool.ok) { if (err) *err = "kd build on the GPU: " + pool.err; return false; }
    }

    out.nodes.resize(n_nodes);
    out.refs.resize(n_refs);
    GK_TRY(hipMemcpy(out.nodes.data(), d_nodes, (size_t)n_nodes * sizeof(Node), hipMemcpyDeviceToHost));
    if (n_refs) GK_TRY(hipMemcpy(out.refs.data(), d_refs, (size_t)n_refs * sizeof(int), hipMemcpyDeviceToHost));
    int counters[4];
    GK_TRY(hipMemcpy(counters, d_counters, sizeof(counters), hipMemcpyDeviceToHost));
    out.max_depth = counters[0];
    out.n_leaves = counters[1];
    out.n_empty_leaves = counters[2];
    return true;
}

}  // namespace gptkd
