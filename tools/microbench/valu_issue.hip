// valu_issue.hip -- issue cost of wave64 vector instructions on one SIMD (gfx950): cycles per instruction for a stream of
// independent or dependent v_add_u32 / v_fma_f32 / v_cndmask_b32 / v_cmp_lt_f32, with W waves per SIMD competing.
// Also LDS instructions (ds_bpermute_b32, ds_read_b32, ds_read2st64_b32, ds_read_b128, ds_write_b32, ds_write2st64_b32), in
// groups of four with one s_waitcnt per group.
// And packed fp32 (v_pk_mul_f32 / v_pk_add_f32, also with a scalar-register pair and a broadcast low half as sources).
// Measured on MI355X at 8 waves per SIMD, cycles of a SIMD's time per wave64 instruction (2.4 GHz assumed): plain vector
// 2.4 - 2.7, compare + select 3.7, scalar 4.3 (the CU's scalar unit serves its four SIMDs), vector and scalar interleaved
// 2.5 (they overlap), packed fp32 5.0 (no gain over two plain instructions), ds_read_b32 10, ds_read2st64_b32 / ds_read_b128
// 18.5, ds_write_b32 18.5, ds_write2st64_b32 27.5, ds_bpermute_b32 26.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip ; run: ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int iters, float seed) {
    __shared__ float lds[2048];
    lds[threadIdx.x] = seed; lds[threadIdx.x + 64] = seed; lds[threadIdx.x + 512] = seed; lds[threadIdx.x * 4 + 1024] = seed;
    __syncthreads();
    const unsigned ad4 = (unsigned)(size_t)lds + threadIdx.x * 4, ad16 = (unsigned)(size_t)lds + 4096 + threadIdx.x * 16;
    typedef float vf4 __attribute__((ext_vector_type(4)));
    typedef float vf2 __attribute__((ext_vector_type(2)));
    vf4 q0, q1, q2, q3;
    vf2 p0 = {seed, seed}, p1 = {seed, 1.0f}, p2 = {1.0f, seed}, p3 = {seed, seed};
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float f0 = seed + threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP / 4; r++) {
            if (KIND == 0) {          // independent integer adds
                asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %1, %1, 1\n v_add_u32 %2, %2, 1\n v_add_u32 %3, %3, 1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
            } else if (KIND == 1) {   // dependent integer adds
                asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(a0));
            } else if (KIND == 2) {   // independent fp32 fma
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
            } else if (KIND == 3) {   // dependent fp32 multiplies
                asm volatile("v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0\n v_mul_f32 %0, %0, %0" : "+v"(f0));
            } else if (KIND == 4) {   // compares into a scalar mask + selects
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %0\n v_cndmask_b32 %3, %3, %2, vcc" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : : "vcc");
            } else if (KIND == 5) {   // scalar ALU
                asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_and_b64 s[22:23], s[22:23], s[24:25]\n s_or_b64 s[26:27], s[26:27], s[24:25]" : : : "s20", "s21", "s22", "s23", "s26", "s27", "scc");
            } else if (KIND == 7) {   // arbitrary lane permute through the LDS crossbar
                asm volatile("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(ad4));
            } else if (KIND == 8) {
                asm volatile("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n s_waitcnt lgkmcnt(0)" : "=v"(f0), "=v"(f1), "=v"(f2), "=v"(f3) : "v"(ad4));
            } else if (KIND == 9) {
                asm volatile("ds_read2st64_b32 %0, %4 offset1:2\n ds_read2st64_b32 %1, %4 offset1:2\n ds_read2st64_b32 %2, %4 offset1:2\n ds_read2st64_b32 %3, %4 offset1:2\n s_waitcnt lgkmcnt(0)" : "=v"(p0), "=v"(p1), "=v"(p2), "=v"(p3) : "v"(ad4));
            } else if (KIND == 10) {
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4\n ds_read_b128 %2, %4\n ds_read_b128 %3, %4\n s_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3) : "v"(ad16));
            } else if (KIND == 11) {
                asm volatile("ds_write_b32 %4, %0\n ds_write_b32 %4, %1 offset:256\n ds_write_b32 %4, %2 offset:512\n ds_write_b32 %4, %3 offset:768\n s_waitcnt lgkmcnt(0)" : : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(ad4));
            } else if (KIND == 12) {
                asm volatile("ds_write2st64_b32 %2, %0, %1 offset1:2\n ds_write2st64_b32 %2, %1, %0 offset1:2\n ds_write2st64_b32 %2, %0, %1 offset1:2\n ds_write2st64_b32 %2, %1, %0 offset1:2\n s_waitcnt lgkmcnt(0)" : : "v"(f0), "v"(f1), "v"(ad4));
            } else if (KIND == 13) {  // packed fp32: two multiplications / additions per lane and instruction
                asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            } else if (KIND == 14) {  // packed fp32 with a scalar-register pair and a broadcast low half as sources
                asm volatile("v_pk_mul_f32 %0, %0, s[20:21]\n v_pk_add_f32 %1, %1, %2 op_sel_hi:[1,0]\n v_pk_mul_f32 %2, %2, s[20:21]\n v_pk_add_f32 %3, %3, %0 op_sel_hi:[1,0]" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            } else {                  // vector and scalar interleaved
                asm volatile("v_add_u32 %0, %0, 1\n s_add_u32 s20, s20, 1\n v_add_u32 %1, %1, 1\n s_and_b64 s[22:23], s[22:23], s[24:25]" : "+v"(a0), "+v"(a1) : : "s20", "s22", "s23", "scc");
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (KIND == 9 || KIND == 13 || KIND == 14) { f0 = p0.x + p1.y; f1 = p2.x + p3.y; }
    if (KIND == 10) { f0 = q0.x + q1.y; f1 = q2.z + q3.w; }
    if (a0 + a1 + a2 + a3 == 12345u && f0 + f1 + f2 + f3 == 1.5f) out[0] = 0;
}

template <int KIND>
void run(const char* name, unsigned long long* d, int waves_per_simd) {
    // one CU's worth: 4 SIMDs x waves_per_simd single-wave workgroups would need placement control; instead fill the
    // whole GPU evenly: 256 CUs x 4 SIMDs x waves_per_simd workgroups of one wave each
    const int blocks = 256 * 4 * waves_per_simd, iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    const double n_inst = (double)iters * REP;
    // s_memtime / readcyclecounter ticks at a fixed 100 MHz on this part; use wall time with the instruction count instead
    printf("%-34s waves/SIMD %d: %.3f ms -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz), counter ticks/inst %.3f\n", name, waves_per_simd,
           ms, ms * 1e6 / (n_inst * waves_per_simd), ms * 1e6 / (n_inst * waves_per_simd) * 2.4, avg / n_inst);
}

int main() {
    unsigned long long* d;
    hipMalloc(&d, 256 * 4 * 16 * 8);
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_add_u32 independent", d, w);
        run<1>("v_add_u32 dependent", d, w);
        run<2>("v_fma_f32 independent", d, w);
        run<3>("v_mul_f32 dependent", d, w);
        run<4>("v_cmp + v_cndmask", d, w);
        run<5>("s_add / s_and / s_or", d, w);
        run<6>("v_add_u32 + scalar interleaved", d, w);
        run<7>("ds_bpermute_b32", d, w);
        run<8>("ds_read_b32", d, w);
        run<9>("ds_read2st64_b32", d, w);
        run<10>("ds_read_b128", d, w);
        run<11>("ds_write_b32", d, w);
        run<12>("ds_write2st64_b32", d, w);
        run<13>("v_pk_mul_f32 / v_pk_add_f32", d, w);
        run<14>("v_pk with sgpr pair / op_sel_hi", d, w);
    }
    return 0;
}
