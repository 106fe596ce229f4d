// gpt_device_structs.h -- kernel argument blocks shared by gpt_kernels.hip and
// gpt_api.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kd_traverse.h"

// Per-pass device counters: counts[d] = live paths entering depth d, followed by the work-queue
// heads of the kd intersect kernel: the queue of depth d is cut into GPT_WORK_PARTS contiguous
// parts, one per XCD, each with its own head word on its own 128-byte line
// (work[(d * GPT_WORK_PARTS + part) * GPT_WORK_STRIDE]).
#define GPT_MAX_DEPTH 64
#define GPT_MAX_DEPTH_SLOTS 72
#define GPT_WORK_PARTS 8
#define GPT_WORK_STRIDE 32
#define GPT_COUNTER_WORDS (GPT_MAX_DEPTH_SLOTS + GPT_MAX_DEPTH_SLOTS * GPT_WORK_PARTS * GPT_WORK_STRIDE)

// Cumulative 64-bit statistics.
#define GPT_STAT_LIVE0 0                      // [GPT_MAX_DEPTH] live paths per depth
#define GPT_STAT_KD_NODES (GPT_MAX_DEPTH + 0)
#define GPT_STAT_KD_TRIS (GPT_MAX_DEPTH + 1)
#define GPT_STAT_HITS (GPT_MAX_DEPTH + 2)     // closest-hit queries that found something
#define GPT_STAT_WAVE_ITERS_NODE (GPT_MAX_DEPTH + 3)   // wave executions of the node-loop body
#define GPT_STAT_WAVE_ITERS_TRI (GPT_MAX_DEPTH + 4)    // wave executions of the triangle-loop body
#define GPT_STAT_BUSY_LANE_ROUNDS (GPT_MAX_DEPTH + 5)  // walk iterations summed over the lanes holding a ray
#define GPT_STAT_WAVE_ROUNDS (GPT_MAX_DEPTH + 6)       // walk iterations of the waves
#define GPT_STAT_NODES_D0 (GPT_MAX_DEPTH + 8)          // [GPT_MAX_DEPTH] kd nodes visited per depth
#define GPT_STAT_TRIS_D0 (2 * GPT_MAX_DEPTH + 8)       // [GPT_MAX_DEPTH] triangles tested per depth
#define GPT_STAT_WORDS (3 * GPT_MAX_DEPTH + 8)

struct DevGeometry {
    const float4* spheres;     // (centre.xyz, radius * radius)
    const int* sphere_mat;     // material index per sphere
    int n_spheres;
    int n_tris;
    const float4* tri_nm;      // (Poly.n.xyz, material index as int bits) per triangle
    const float4* mats;        // 4 x float4 per material:
                               //   (diffuse.rgb, ior) (specular.rgb, transmits) (emitted.rgb, sigma_s) (sigma_a.rgb, 0)
    const pt::TriRec* tris;    // one record per triangle, in Poly order (brute-force loop)
    pt::KdView kd;             // nodes, leaf-ordered triangle records, root box, slack
    float ground_elevation;
    int ground_enabled;
};
