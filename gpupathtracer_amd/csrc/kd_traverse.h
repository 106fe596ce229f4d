// kd_traverse.h -- closest-triangle query through the flattened kd-tree.
//
// The result is defined by the reference's brute-force loop
// (path_tracer_kernel.cu:536-557): among all triangles whose Moller-Trumbore
// test (:226-248) reports a hit with 0 < t < t_limit, the smallest t; equal t
// goes to the lowest triangle index (the loop's strict '<').  Traversal may
// only skip triangles that cannot win, so every interval decision below is
// taken against split planes thickened by a spatial slack `sigma` (and a small
// relative slack on t): a node is skipped only when the ray misses it by more
// than any rounding error of the plane distances or of the triangle test.
//
// Traversal order follows the classic kd scheme (design source:
// cukd/detail/kdtree_kernels.cu:50-170): descend towards the side of the
// origin, defer the far child with its [tmin,tmax], stop as soon as the best
// hit lies in front of every deferred interval.
#pragma once
#include "pt_device.h"

namespace pt {

struct KdNode { uint32_t w0, w1; };

// Triangle record, 40 bytes: a, edge1 = b - a, edge2 = c - a, index (int bits; the triangle's
// position in the caller's Poly array).
#define PT_TRI_REC_FLOATS 10
struct TriRec { float f[PT_TRI_REC_FLOATS]; };

// Leaves do not point at triangles through an index list: the records of a
// leaf's triangles are stored contiguously in leaf order (a triangle that several
// leaves reference is stored once per leaf), so a leaf visit reads
// count x 40 consecutive bytes.
struct KdView {
    const KdNode* nodes;
    const TriRec* leaf_tris;   // one record per (leaf, triangle) reference, leaf by leaf
    const int32_t* refs;       // non-null: leaves index `leaf_tris` (then one record per triangle) through this list
    float bmin[3], bmax[3];
    float sigma;       // spatial slack: 2e-5 * largest extent of the root box
};

#ifndef PT_KD_REL_SLACK
#define PT_KD_REL_SLACK 4e-6f
#endif
#define PT_KD_CHILD_MASK 0x0fffffffu   // child index field of an inner node's w0 (above the 2 axis bits)

// Sets bits 30 / 31 in the w0 of every inner node whose left / right child is an empty leaf
// (what the device copy of a tree gets; see KdCursor::node_step).  Node indices must fit 28 bits.
inline void kd_mark_empty_children(KdNode* nodes, size_t n) {
    for (size_t i = 0; i < n; i++) {
        if ((nodes[i].w0 & 3u) == 3u) continue;
        const uint32_t left = (nodes[i].w0 >> 2) & PT_KD_CHILD_MASK;
        const KdNode& l = nodes[left];
        const KdNode& r = nodes[left + 1];
        if ((l.w0 & 3u) == 3u && l.w1 == 0u) nodes[i].w0 |= 1u << 30;
        if ((r.w0 & 3u) == 3u && r.w1 == 0u) nodes[i].w0 |= 1u << 31;
    }
}

// Instrumented builds only: counts how many times a WAVE executes the body of the node loop
// (slot 0) and of the triangle loop (slot 1), to relate them to the lanes' own counts
// (SIMD utilisation of each loop).  The kernels define the macro; elsewhere it is empty.
#ifndef PT_COUNT_WAVE_ITERATION
#define PT_COUNT_WAVE_ITERATION(slot) ((void)0)
#endif

PT_HD float sel3(int axis, float x, float y, float z) { return axis == 0 ? x : (axis == 1 ? y : z); }

// A per-lane truth value of the traversal's decision logic.  On the GPU it is the wave's 64-bit
// lane mask of the condition, which lives in scalar registers: a float comparison writes such a
// mask directly, and every '&', '|', '~' between masks is then ONE scalar instruction for the
// whole wave, next to the vector pipeline that bounds the kernel (left as `bool`s, the compiler
// materialises them as 0 / 1 in vector registers and combines them there).  on() turns the mask
// back into the lane's own truth value for a select or a branch at no cost.  On the host it is
// a plain bool.
struct KdPred {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long m;
    static __device__ __forceinline__ KdPred of(bool b) { KdPred p; p.m = __builtin_amdgcn_ballot_w64(b); return p; }
    static __device__ __forceinline__ KdPred all() { KdPred p; p.m = ~0ull; return p; }
    __device__ __forceinline__ bool on() const { return __builtin_amdgcn_inverse_ballot_w64(m); }
    __device__ __forceinline__ KdPred operator~() const { KdPred p; p.m = ~m; return p; }
    __device__ __forceinline__ unsigned count() const { return (unsigned)__builtin_popcountll(m); }   // lanes set (of the active ones, for a mask made by of())
    __device__ __forceinline__ unsigned long long bits() const { return m; }
#else
    bool m;
    static PT_HD KdPred of(bool b) { KdPred p; p.m = b; return p; }
    static PT_HD KdPred all() { KdPred p; p.m = true; return p; }
    PT_HD bool on() const { return m; }
    PT_HD KdPred operator~() const { KdPred p; p.m = !m; return p; }
    PT_HD unsigned count() const { return m ? 1u : 0u; }
    PT_HD unsigned long long bits() const { return m ? ~0ull : 0ull; }
#endif
    static PT_HD KdPred none() { return ~all(); }
    PT_HD KdPred operator&(KdPred b) const { KdPred p; p.m = m & b.m; return p; }
    PT_HD KdPred operator|(KdPred b) const { KdPred p; p.m = m | b.m; return p; }
    PT_HD KdPred operator^(KdPred b) const { KdPred p; p.m = m ^ b.m; return p; }
};

PT_HD float bits_to_float(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; __builtin_memcpy(&f, &u, 4); return f;
#endif
}

// Reciprocal for the traversal's interval arithmetic.  Which nodes are visited only has to be
// conservative, not reproducible to the bit: the hardware reciprocal (1 ulp, one instruction
// instead of the ten of an IEEE division) changes a plane distance by ~1e-7 relative, far inside
// the 4e-6 relative slack every interval decision carries.  x = 0 gives +-inf on both sides.
PT_HD float kd_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}

PT_HD int float_to_int_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_int(f);
#else
    int i; __builtin_memcpy(&i, &f, 4); return i;
#endif
}

PT_HD int load_tri(const TriRec* tris, size_t slot, V3& a, V3& e1, V3& e2) {
    const float* f = tris[slot].f;
    a = v3(f[0], f[1], f[2]);
    e1 = v3(f[3], f[4], f[5]);
    e2 = v3(f[6], f[7], f[8]);
    return float_to_int_bits(f[9]);
}

// The part of the ray within [0, t_limit] that lies inside the root box padded by the slack; false:
// none (or a NaN component: such a ray hits nothing in the reference's loop -- every comparison of its
// triangle test fails -- and must not wander through the tree either).  `inv` receives 1 / d.
// The plane thickness a ray is traversed with: the tree's sigma (2e-5 of the scene's extent) plus 2e-6 of the ray origin's
// distance from the middle of the root box.  The triangle test forms `origin - vertex` first; far from the mesh that
// difference is rounded to half an ulp of the DISTANCE, which moves the test's idea of where the ray passes by up to
// ~5e-7 of it -- for an origin 10 000 scene extents away, more than sigma.  The traversal must visit every leaf the
// brute-force loop could report a hit in, rounding artefacts included, so its planes thicken with the distance.
#ifndef PT_KD_DISTANCE_SLACK
#define PT_KD_DISTANCE_SLACK 2e-6f
#endif
PT_HD float kd_ray_sigma(const KdView& kd, V3 o) {
    const float dx = fabsf(o.x - 0.5f * (kd.bmin[0] + kd.bmax[0])), dy = fabsf(o.y - 0.5f * (kd.bmin[1] + kd.bmax[1])),
                dz = fabsf(o.z - 0.5f * (kd.bmin[2] + kd.bmax[2]));
    return kd.sigma + PT_KD_DISTANCE_SLACK * fmaxf(dx, fmaxf(dy, dz));
}

PT_HD bool kd_root_interval(const KdView& kd, float sigma, V3 o, V3 d, float t_limit, V3& inv, float& tmin, float& tmax) {
    if (!(o.x == o.x && o.y == o.y && o.z == o.z && d.x == d.x && d.y == d.y && d.z == d.z)) return false;
    tmin = 0.0f;
    tmax = t_limit;
    inv = v3(kd_rcp(d.x), kd_rcp(d.y), kd_rcp(d.z));
    const float oc[3] = {o.x, o.y, o.z};
    const float dc[3] = {d.x, d.y, d.z};
    const float ic[3] = {inv.x, inv.y, inv.z};
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int a = 0; a < 3; a++) {
        float lo = kd.bmin[a] - sigma, hi = kd.bmax[a] + sigma;
        if (dc[a] != 0.0f) {
            float iv = ic[a];
            float ta = (lo - oc[a]) * iv, tb = (hi - oc[a]) * iv;
            float tn = fminf(ta, tb), tf = fmaxf(ta, tb);
            tn -= PT_KD_REL_SLACK * fabsf(tn);
            tf += PT_KD_REL_SLACK * fabsf(tf);
            tmin = fmaxf(tmin, tn);
            tmax = fminf(tmax, tf);
        } else if (!(oc[a] >= lo && oc[a] <= hi)) {
            return false;
        }
    }
    return tmin <= tmax;
}

// Stack concept: void reset(); void push(uint32_t node, float tmin, float tmax);
//                bool pop(uint32_t& node, float& tmin, float& tmax);
//
// KdCursor is a resumable traversal: begin() clips the ray against the root box,
// node_step() takes one inner node, next_interval() moves to the next deferred
// interval.  advance() strings them together for one lane on its own (walk to the
// next leaf, test its triangles, move on; true once nothing can beat the current
// best): the host harness loops over it.  The GPU kernel calls the pieces itself,
// because there the 64 lanes of a wave walk together and share the triangle tests.
template <class Stack, bool STATS>
struct KdCursor {
    V3 o, d, inv;
    float sigma;                 // this ray's plane thickness (kd_ray_sigma)
    float tmin, tmax, bestT;
    int bestTri;
    uint32_t node;
    Stack st;

    // false: the ray cannot hit any triangle (it misses the padded root box).
    PT_HD bool begin(const KdView& kd, V3 o_, V3 d_, float t_limit) {
        o = o_; d = d_;
        bestT = t_limit;
        bestTri = -1;
        sigma = kd_ray_sigma(kd, o);
        if (!kd_root_interval(kd, sigma, o, d, t_limit, inv, tmin, tmax)) return false;
        node = 0;
        st.reset();
        return true;
    }

    // Next deferred interval that can still hold a closer (or equal) hit; false: none left.
    PT_HD bool next_interval() {
        do {
            if (!st.pop(node, tmin, tmax)) return false;
        } while (bestT < tmin);
        tmax = fminf(tmax, bestT);
        return true;
    }

    // One inner node `w`, for the lanes in `act` (the others keep their state; what they compute
    // from a stale or dummy `w` is discarded): chooses the child to enter (into `node`), defers
    // the other one with its interval when the ray can reach it, and trims [tmin, tmax].
    // Bits 30 / 31 of an inner node's w0 say that its left / right child is an EMPTY leaf
    // (kd_mark_empty_children; the builder's own output has them clear, and 28 bits of child
    // index).  Such a child is never entered or deferred: no load, no stack traffic.  Returns
    // the lanes for which that leaves nothing to visit under this node (the caller moves on to
    // the next deferred interval, exactly as it would from an empty leaf).
    PT_HD KdPred node_step(const KdView& kd, KdNode w, KdPred act) {
        const int axis = (int)(w.w0 & 3u);
        const float split = bits_to_float(w.w1);
        const uint32_t left = (w.w0 >> 2) & PT_KD_CHILD_MASK, right = left + 1u;
        const KdPred emptyL = KdPred::of((w.w0 & 0x40000000u) != 0u), emptyR = KdPred::of((int32_t)w.w0 < 0);
        const float oa = sel3(axis, o.x, o.y, o.z);
        // the direction enters through its reciprocal only: +-inf stands for a ray parallel to the plane
        const float ia = sel3(axis, inv.x, inv.y, inv.z);
        // On the plane exactly, the side the ray leaves towards comes first (a parallel ray there
        // visits both children with the whole interval, in either order).
        const KdPred leftFirst = KdPred::of(oa < split) | (KdPred::of(oa == split) & KdPred::of(ia < 0.0f));
        const KdPred nearEmpty = (leftFirst & emptyL) | (~leftFirst & emptyR);
        const KdPred farEmpty = (leftFirst & emptyR) | (~leftFirst & emptyL);
        // Four outcomes:
        //   near only   plane beyond the interval, or behind the origin, or ray parallel
        //               to it and outside the thickened plane
        //   both, whole interval each   origin inside the thickened plane (the crossing may
        //               be real or lie behind the origin), or parallel inside it
        //   far only    plane crossed before the interval starts
        //   both, cut   near gets [tmin, min(tmax,hi)], far gets [max(tmin,lo), tmax]
        // (1 / d is infinite for d = 0; a finite reciprocal beyond 3e38 belongs to a ray that moves
        // less than 1e-29 along the axis over any interval -- parallel as well.  One comparison.)
        const KdPred parallel = KdPred::of(fabsf(ia) > 3.0e38f);
        const KdPred inSlab = KdPred::of(fabsf(oa - split) <= sigma);
        const float tp = (split - oa) * ia;
        const float s = sigma * fabsf(ia) + PT_KD_REL_SLACK * fabsf(tp);
        const float lo = tp - s, hi = tp + s;
        const KdPred crossing = ~parallel & ~(KdPred::of(lo > tmax) | KdPred::of(hi < 0.0f));   // the plane is met within the interval
        const KdPred behind = KdPred::of(lo < 0.0f);
        const KdPred before = KdPred::of(hi < tmin);
        const KdPred bothWhole = (parallel & inSlab) | (crossing & behind);
        const KdPred farOnly = crossing & ~behind & before;
        const KdPred bothCut = crossing & ~behind & ~before;
        const KdPred both = bothWhole | bothCut;
        // (tmin and tmax are never NaN; a NaN lo / hi compares false and leaves them as they are,
        // which is what fmaxf / fminf would do, at half the instructions)
        const float farTmin = (bothCut & KdPred::of(lo > tmin)).on() ? lo : tmin;
        if ((act & both & ~nearEmpty & ~farEmpty).on()) st.push(leftFirst.on() ? right : left, farTmin, tmax);
        const KdPred takeFar = farOnly | (both & nearEmpty);
        // the child entered is the right one when exactly one of "far child" and "left is near" holds
        const KdPred pickRight = (takeFar & leftFirst) | (~takeFar & ~leftFirst);
        const uint32_t child = pickRight.on() ? right : left;
        node = act.on() ? child : node;
        tmin = (act & both & nearEmpty).on() ? farTmin : tmin;
        tmax = (act & bothCut & ~nearEmpty & KdPred::of(hi < tmax)).on() ? hi : tmax;
        return act & ((takeFar & farEmpty) | (~takeFar & nearEmpty));
    }

    // One leaf visit.  Returns true when the traversal is complete.
    PT_HD bool advance(const KdView& kd, uint32_t& nodesVisited, uint32_t& trisTested) {
        KdNode w = kd.nodes[node];
        if (STATS) nodesVisited++;
        while ((w.w0 & 3u) != 3u) {
            if (STATS) PT_COUNT_WAVE_ITERATION(0);
            if (node_step(kd, w, KdPred::all()).on()) return !next_interval();
            w = kd.nodes[node];
            if (STATS) nodesVisited++;
        }
        // ---- leaf ----
        const uint32_t first = w.w0 >> 2, count = w.w1;
        for (uint32_t k = 0; k < count; k++) {
            if (STATS) PT_COUNT_WAVE_ITERATION(1);
            V3 a, e1, e2;
            const size_t rec = kd.refs ? (size_t)kd.refs[first + k] : (size_t)first + k;
            const int idx = load_tri(kd.leaf_tris, rec, a, e1, e2);
            float t;
            if (STATS) trisTested++;
            if (triangle_t(a, e1, e2, o, d, t)) {
                if (t > 0.0f && (t < bestT || (t == bestT && bestTri >= 0 && idx < bestTri))) {
                    bestT = t;
                    bestTri = idx;
                }
            }
        }
        return !next_interval();
    }
};

template <class Stack, bool STATS>
PT_HD void kd_closest_hit(const KdView& kd, V3 o, V3 d, float& bestT, int& bestTri, Stack& st,
                          uint32_t& nodesVisited, uint32_t& trisTested) {
    KdCursor<Stack, STATS> c;
    c.st = st;
    if (c.begin(kd, o, d, bestT)) {
        while (!c.advance(kd, nodesVisited, trisTested)) {}
        bestT = c.bestT;
        bestTri = c.bestTri;
    }
    st = c.st;
}

// The reference's loop itself (:536-557) over the same triangle records.
template <bool STATS>
PT_HD void brute_closest_hit(const TriRec* tris, int n, V3 o, V3 d, float& bestT, int& bestTri,
                             uint32_t& trisTested) {
    for (int i = 0; i < n; i++) {
        V3 a, e1, e2;
        load_tri(tris, (size_t)i, a, e1, e2);
        float t;
        if (triangle_t(a, e1, e2, o, d, t)) {
            if (t > 0 && t < bestT) { bestT = t; bestTri = i; }
        }
    }
    if (STATS) trisTested += (uint32_t)n;
}

}  // namespace pt
