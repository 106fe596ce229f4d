// thrust_rng_dump.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (container only).
//
// The reference's random numbers come from Thrust (path_tracer_kernel.cu:139-142, 492-493:
// thrust::default_random_engine + thrust::uniform_real_distribution<float>(0,1)).  Thrust is not
// vendored by the reference; the image carries rocThrust's copy of the same classes under
// /opt/rocm/include/thrust.  This host program uses THOSE classes -- no restatement -- and writes,
// for every seed given on stdin (one unsigned decimal per line), the first N raw engine values and
// the first N uniforms of a freshly seeded engine, as little-endian binary:
//   nSeeds, N, then per seed: seed (uint32), raw[N] (uint32), uniform[N] (float32).
// (raw and uniform come from two engines seeded alike: a draw of the distribution advances it.)
//
//   thrust_rng_dump N out.bin < seeds.txt
#include <thrust/random.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
    if (argc != 3) { fprintf(stderr, "usage: thrust_rng_dump N out.bin < seeds\n"); return 2; }
    const int N = atoi(argv[1]);
    std::vector<uint32_t> seeds;
    unsigned long long s;
    while (scanf("%llu", &s) == 1) seeds.push_back((uint32_t)s);
    FILE* f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    const int32_t n = (int32_t)seeds.size(), nn = N;
    fwrite(&n, 4, 1, f);
    fwrite(&nn, 4, 1, f);
    for (uint32_t seed : seeds) {
        fwrite(&seed, 4, 1, f);
        thrust::default_random_engine raw(seed);
        for (int i = 0; i < N; i++) { uint32_t v = (uint32_t)raw(); fwrite(&v, 4, 1, f); }
        thrust::default_random_engine rng(seed);
        thrust::uniform_real_distribution<float> u01(0, 1);
        for (int i = 0; i < N; i++) { float v = (float)u01(rng); fwrite(&v, 4, 1, f); }
    }
    // the documented known answer, checked here against the real class as well
    thrust::minstd_rand kat;
    kat.discard(9999);
    const uint32_t v10000 = (uint32_t)kat();
    fwrite(&v10000, 4, 1, f);
    fclose(f);
    return v10000 == 399268537u ? 0 : 3;
}
