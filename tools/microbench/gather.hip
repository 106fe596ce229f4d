// gather.hip -- ceiling for the traversal's access pattern on MI355X: every lane follows its
// own chain of dependent random 8-byte loads (pointer chasing through a random permutation),
// with a configurable share of idle lanes.  Reports lane-loads per second for several table
// sizes.  Build: hipcc --offload-arch=gfx950 -O3 gather.hip -o gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include <algorithm>

__global__ __launch_bounds__(256) void chase(const uint2* __restrict__ table, unsigned n, int steps, int active_lanes,
                                             unsigned* __restrict__ out) {
    unsigned tid = blockIdx.x * 256 + threadIdx.x;
    unsigned lane = threadIdx.x & 63;
    unsigned idx = (tid * 2654435761u) % n;
    unsigned acc = 0;
    if ((int)lane < active_lanes) {
        for (int s = 0; s < steps; s++) {
            uint2 v = table[idx];
            idx = v.x;
            acc += v.y;
        }
    }
    out[tid] = acc + idx;
}

int main(int argc, char** argv) {
    int steps = 256;
    std::vector<size_t> sizes = {4, 32, 128, 512, 2048};
    if (argc > 1) sizes = {(size_t)atoi(argv[1])};       // one table size (for counter passes)
    for (size_t mb : sizes) {
        unsigned n = (unsigned)(mb * 1024 * 1024 / 8);
        std::vector<uint2> h(n);
        std::vector<unsigned> perm(n);
        for (unsigned i = 0; i < n; i++) perm[i] = i;
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        for (unsigned i = 0; i < n; i++) { h[perm[i]].x = perm[(i + 1) % n]; h[perm[i]].y = i; }
        uint2* d; unsigned* out;
        hipMalloc(&d, (size_t)n * 8);
        hipMemcpy(d, h.data(), (size_t)n * 8, hipMemcpyHostToDevice);
        for (int blocks_per_cu : {6, 8}) {
            int grid = 256 * blocks_per_cu;
            hipMalloc(&out, (size_t)grid * 256 * 4);
            for (int active : {64, 16}) {
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, 0, d, n, 32, active, out);
                hipDeviceSynchronize();
                hipEventRecord(e0);
                hipLaunchKernelGGL(chase, dim3(grid), dim3(256), 0, 0, d, n, steps, active, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double loads = (double)grid * 4 * active * steps;
                printf("table %5zu MB  waves/CU %2d  active lanes %2d : %.3f ms  %.1f G lane-loads/s  (%.0f ns per dependent step)\n",
                       mb, blocks_per_cu * 4, active, ms, loads / ms / 1e6, ms * 1e6 / steps);
            }
            hipFree(out);
        }
        hipFree(d);
    }
    return 0;
}
