// helpers_dump.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (container only).
//
// Driver around the REFERENCE's own text of the per-path helper functions:
// GPUPathTracer/GPUPathTracer/path_tracer_kernel.cu lines 46-57 (its #defines), 75-84 (randhash)
// and 194-475 (positionAlongRay, findGroundPlaneIntersection, FindTriangleIntersect,
// findSphereIntersection, randomCosineWeightedDirectionInHemisphere, randomDirectionInSphere,
// computeReflectionDirection, computeTransmissionDirection, computeFresnel,
// computeBackgroundColor, computeTransmission).  The Makefile cuts those line ranges out of the
// reference checkout into a temporary directory at build time (REF_LINES_DIR, deleted after the
// compile; never committed) and compiles this file with plain g++ against
//   * the genuine CUDA toolkit headers that ship inside the image's Triton wheel
//     (cuda_runtime.h, vector_types.h, ...: __host__ / __device__ are attributes g++ ignores),
//   * the reference's own cutil_math.h and struct headers (-I into the checkout).
// No header of this repository's making stands in for anything, and no min / max / abs overload
// is added.  Consequence: findSphereIntersection is compiled (it is part of the line range) but
// NOT dumped -- under a host compiler its min(t1, t2) / max(t1, t2) bind to cutil_math.h's host
// int min(int, int) (cutil_math.h:48-56) and truncate, which is not what the device build does.
// The two __global__ kernels (:125-191, :477-725) stay outside: they need thrust/random.h, and
// rocThrust only builds under HIP, whose float3 clashes with the CUDA headers'.
//
//   helpers_dump <function> <n> <in.bin> <out.bin>
// in.bin holds n records of 32-bit words (float or uint32, as the function takes them), out.bin
// receives n records of outputs; record layouts are in the table in main().
#include "cuda_runtime.h"
#include "device_launch_parameters.h"
#include "driver_functions.h"

#include "cutil_math.h"
#include <stdio.h>
#include <stdlib.h>
#include <cmath>
#include <cassert>
#include <ctime>
#include <iostream>
#include <iomanip>
#include <limits>

#include "basic_math.h"
#include "image.h"
#include "camera.h"
#include "sphere.h"
#include "poly.h"
#include "ray.h"
#include "fresnel.h"
#include "medium.h"
#include "absorption_and_scattering_properties.h"

#include "ref_lines_46_57.inc"
#include "ref_lines_75_84.inc"
#include "ref_lines_194_475.inc"

#include <cstring>
#include <vector>

static float3 f3at(const float* p) { return make_float3(p[0], p[1], p[2]); }
static void put3(float* p, float3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

struct Fn { const char* name; int n_in, n_out; void (*run)(const float* in, float* out); };

static void run_randhash(const float* in, float* out) {
    unsigned int a; memcpy(&a, in, 4);
    unsigned int r = randhash(a);
    memcpy(out, &r, 4);
}
static void run_position(const float* in, float* out) {
    Ray r; r.origin = f3at(in); r.direction = f3at(in + 3);
    put3(out, positionAlongRay(r, in[6]));
}
static void run_ground(const float* in, float* out) {
    Ray r; r.origin = f3at(in + 1); r.direction = f3at(in + 4);
    float3 n = make_float3(0, 0, 0);
    out[0] = findGroundPlaneIntersection(in[0], r, n);
    put3(out + 1, n);
}
static void run_triangle(const float* in, float* out) {
    // outputs keep their initial bits where the function returns before writing them (det == 0)
    float t, u, v;
    const unsigned int unset = 0x7fc0dead;
    memcpy(&t, &unset, 4); memcpy(&u, &unset, 4); memcpy(&v, &unset, 4);
    bool hit = FindTriangleIntersect(f3at(in), f3at(in + 3), f3at(in + 6), f3at(in + 9), f3at(in + 12), t, u, v);
    unsigned int h = hit ? 1u : 0u;
    memcpy(out, &h, 4);
    out[1] = t; out[2] = u; out[3] = v;
}
static void run_cosine(const float* in, float* out) {
    put3(out, randomCosineWeightedDirectionInHemisphere(f3at(in), in[3], in[4]));
}
static void run_sphere_dir(const float* in, float* out) { put3(out, randomDirectionInSphere(in[0], in[1])); }
static void run_reflect(const float* in, float* out) { put3(out, computeReflectionDirection(f3at(in), f3at(in + 3))); }
static void run_transmit(const float* in, float* out) {
    put3(out, computeTransmissionDirection(f3at(in), f3at(in + 3), in[6], in[7]));
}
// computeFresnel on a transmission direction given by the caller
static void run_fresnel(const float* in, float* out) {
    Fresnel f = computeFresnel(f3at(in), f3at(in + 3), in[6], in[7], f3at(in + 8), f3at(in + 11));
    out[0] = f.reflectionCoefficient; out[1] = f.transmissionCoefficient;
}
// the chain traceRayKernel runs at a surface (:652-665): reflection and transmission directions
// from (normal, incident, n1, n2), then computeFresnel on them
static void run_surface(const float* in, float* out) {
    float3 n = f3at(in), i = f3at(in + 3);
    float3 R = computeReflectionDirection(n, i);
    float3 T = computeTransmissionDirection(n, i, in[6], in[7]);
    Fresnel f = computeFresnel(n, i, in[6], in[7], R, T);
    put3(out, R); put3(out + 3, T);
    out[6] = f.reflectionCoefficient; out[7] = f.transmissionCoefficient;
}
static void run_background(const float* in, float* out) { put3(out, computeBackgroundColor(f3at(in))); }
static void run_transmission(const float* in, float* out) { put3(out, computeTransmission(f3at(in), in[3])); }

int main(int argc, char** argv) {
    static const Fn table[] = {
        {"randhash", 1, 1, run_randhash},            // a -> hash
        {"position", 7, 3, run_position},            // origin, direction, t -> point
        {"ground", 7, 4, run_ground},                // elevation, origin, direction -> t, normal
        {"triangle", 15, 4, run_triangle},           // a, b, c, origin, direction -> hit, t, bary1, bary2
        {"cosine", 5, 3, run_cosine},                // normal, xi1, xi2 -> direction
        {"sphere_dir", 2, 3, run_sphere_dir},        // xi1, xi2 -> direction
        {"reflect", 6, 3, run_reflect},              // normal, incident -> direction
        {"transmit", 8, 3, run_transmit},            // normal, incident, n1, n2 -> direction (zero on TIR)
        {"fresnel", 14, 2, run_fresnel},             // normal, incident, n1, n2, R, T -> reflection, transmission coefficient
        {"surface", 8, 8, run_surface},              // normal, incident, n1, n2 -> R, T, reflection, transmission coefficient
        {"background", 3, 3, run_background},        // direction -> colour
        {"transmission", 4, 3, run_transmission},    // absorption coefficient, distance -> colour
    };
    if (argc != 5) { fprintf(stderr, "usage: helpers_dump function n in.bin out.bin\n"); return 2; }
    const Fn* fn = nullptr;
    for (const Fn& f : table) if (!strcmp(f.name, argv[1])) fn = &f;
    if (!fn) { fprintf(stderr, "unknown function %s\n", argv[1]); return 2; }
    const long n = atol(argv[2]);
    std::vector<float> in((size_t)n * fn->n_in), out((size_t)n * fn->n_out);
    FILE* f = fopen(argv[3], "rb");
    if (!f || fread(in.data(), 4, in.size(), f) != in.size()) { fprintf(stderr, "short read %s\n", argv[3]); return 1; }
    fclose(f);
    for (long k = 0; k < n; k++) fn->run(in.data() + (size_t)k * fn->n_in, out.data() + (size_t)k * fn->n_out);
    f = fopen(argv[4], "wb");
    if (!f || fwrite(out.data(), 4, out.size(), f) != out.size()) { fprintf(stderr, "short write %s\n", argv[4]); return 1; }
    fclose(f);
    return 0;
}
