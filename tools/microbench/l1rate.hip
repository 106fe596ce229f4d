// l1rate.hip -- how fast a CU's address path (TA + L1) serves the traversal's load shapes when the
// data is cache resident: lane-requests per cycle per CU for 4/8/16-byte loads, random or
// lane-consecutive, and for a 48-byte record fetched as three 16-byte loads.
// Build: hipcc --offload-arch=gfx950 -O3 l1rate.hip -o l1rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(const float4* __restrict__ t, unsigned mask, int iters, float* out) {
    unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned x = tid * 2654435761u + 12345u;
    float acc = 0;
    for (int i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        unsigned r = (x >> 8);
        if (MODE == 0) {          // 4 B random
            acc += reinterpret_cast<const float*>(t)[r & (mask * 4 + 3)];
        } else if (MODE == 1) {   // 8 B random
            float2 v = reinterpret_cast<const float2*>(t)[r & (mask * 2 + 1)]; acc += v.x + v.y;
        } else if (MODE == 2) {   // 16 B random
            float4 v = t[r & mask]; acc += v.x + v.w;
        } else if (MODE == 3) {   // 48 B record, random per lane, three 16 B loads
            unsigned rec = (r & mask) / 3u * 3u;
            float4 a = t[rec], b = t[rec + 1], c = t[rec + 2]; acc += a.x + b.y + c.z;
        } else if (MODE == 4) {   // 48 B records, runs of 4 consecutive records per 4 lanes
            unsigned base = (__shfl(r, threadIdx.x & ~3u) & mask) / 3u * 3u;
            unsigned rec = base + 3u * (threadIdx.x & 3u);
            float4 a = t[rec], b = t[rec + 1], c = t[rec + 2]; acc += a.x + b.y + c.z;
        } else if (MODE == 5) {   // same runs of 4, part-major inside the run (A0..A3 B0..B3 C0..C3)
            unsigned base = (__shfl(r, threadIdx.x & ~3u) & mask) / 12u * 12u;
            unsigned k4 = threadIdx.x & 3u;
            float4 a = t[base + k4], b = t[base + 4 + k4], c = t[base + 8 + k4]; acc += a.x + b.y + c.z;
        } else if (MODE == 6) {   // 16 B, wave-consecutive (fully coalesced), random wave base
            unsigned base = __shfl(r, 0) & mask & ~63u;
            float4 v = t[base + (threadIdx.x & 63u)]; acc += v.x + v.w;
        } else if (MODE == 7) {   // 36 B record as 3 x 12 B loads, random per lane
            const float* f = reinterpret_cast<const float*>(t) + ((r & mask) / 3u * 3u) * 4u;
            float3 a = *reinterpret_cast<const float3*>(f), b = *reinterpret_cast<const float3*>(f + 3), c = *reinterpret_cast<const float3*>(f + 6);
            acc += a.x + b.y + c.z;
        }
    }
    out[tid] = acc;
}

template <int MODE>
void run(const char* name, const float4* d, unsigned mask, float* out, int loads_per_iter) {
    const int iters = 2000, grid = 256 * 8;
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, mask, 10, out);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, d, mask, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double lane_loads = (double)grid * 256 * iters * loads_per_iter;
    double cu_cycles = ms * 1e-3 * 2.4e9 * 256;
    printf("%-58s table %6u KB: %7.3f ms  %.3f lane-loads / cycle / CU  (%.2f wave-instr / 64 cycles)\n", name, (mask + 1) * 16 / 1024, ms,
           lane_loads / cu_cycles, lane_loads / cu_cycles);
}

int main() {
    for (unsigned kb : {16u, 1024u, 262144u}) {
        unsigned n = kb * 1024 / 16;           // float4 entries, power of two
        std::vector<float4> h(n + 64, make_float4(1, 2, 3, 4));
        float4* d; float* out;
        hipMalloc(&d, (n + 64) * sizeof(float4));
        hipMemcpy(d, h.data(), (n + 64) * sizeof(float4), hipMemcpyHostToDevice);
        hipMalloc(&out, 256 * 8 * 256 * 4);
        unsigned mask = n - 1;
        run<0>("4 B random", d, mask, out, 1);
        run<1>("8 B random", d, mask, out, 1);
        run<2>("16 B random", d, mask, out, 1);
        run<3>("48 B record = 3 x 16 B, random per lane", d, mask, out, 3);
        run<7>("36 B record = 3 x 12 B, random per lane", d, mask, out, 3);
        run<4>("48 B records, runs of 4 lanes, record-major", d, mask, out, 3);
        run<5>("48 B records, runs of 4 lanes, part-major", d, mask, out, 3);
        run<6>("16 B wave-consecutive", d, mask, out, 1);
        hipFree(d); hipFree(out);
    }
    return 0;
}
