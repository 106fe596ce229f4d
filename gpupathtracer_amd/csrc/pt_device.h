// pt_device.h -- per-path arithmetic of the path tracer, shared by every HIP
// kernel.  Plain inline functions over scalars and a 3-float vector; nothing
// here touches memory layouts or the HIP runtime, so tests/host_harness can
// also compile it for the host to check the logic on a CPU.
//
// Numeric contract (what makes results comparable with a host build of the
// reference, path_tracer_kernel.cu):
//   * every expression is evaluated in the type a host C++ compiler gives it:
//     float, except where the reference mixes in a double literal or calls
//     pow(float,int) -- those few sub-expressions are double here too;
//   * no fused multiply-add (the translation units are built with
//     -ffp-contract=off), IEEE divide and sqrt;
//   * sin/cos/log/exp are taken in double and rounded once to float, which
//     agrees with a correctly rounded float libm call except when the exact
//     value lies within ~2^-29 ulp of a rounding boundary.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ __forceinline__
#else
#define PT_HD inline
#endif

namespace pt {

struct V3 { float x, y, z; };

PT_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
PT_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
PT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
// cutil_math.h:1123-1126, 1295-1298, 1166-1168, 1184-1188 (+ host rsqrtf :58-61)
PT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD V3 cross(V3 a, V3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
PT_HD float length(V3 v) { return sqrtf(dot(v, v)); }
PT_HD V3 normalize(V3 v) { float invLen = 1.0f / sqrtf(dot(v, v)); return v * invLen; }

// ---- transcendental policy ---------------------------------------------------
// sin / cos of the angles the path tracer forms, TWO_PI * xi in [0, 2 pi]: the algorithm of the
// host C library's sinf / cosf (glibc 2.28 and later: reduction by multiples of pi/2 and two short
// polynomials, all in double, one rounding to float at the end), restated operation for operation.
// Compared exhaustively with sinf / cosf for every float in [0, 6.2831855] (1 086 918 620 values):
// identical bits, with and without fused multiply-add in the polynomials.  So the device takes
// exactly the values a host build of the reference gets from its libm -- and at a third of the
// instructions of a full double-precision sincos.  Host builds send any other argument to double
// sincos rounded once; device code has no such path (every caller passes TWO_PI * xi with xi in
// [0, 1], and a second, never-taken copy of sincos would only cost registers and code size).
PT_HD void sincos_f(float a, float& s, float& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    {
#else
    if (a >= 0.0f && a <= 6.2831855f) {
#endif
        const double hpi_inv = 0x1.45F306DC9C883p+23, hpi = 0x1.921FB54442D18p0;
        const double C0 = 0x1p0, C1 = -0x1.ffffffd0c621cp-2, C2 = 0x1.55553e1068f19p-5, C3 = -0x1.6c087e89a359dp-10,
                     C4 = 0x1.99343027bf8c3p-16;
        const double S1 = -0x1.555545995a603p-3, S2 = 0x1.1107605230bc4p-7, S3 = -0x1.994eb3774cf24p-13;
        double x = (double)a;
        int n = 0;
        if (!(a < 0.75f)) {                      // below 0.75 (the library's top-12-bit test against pi/4): no reduction
            const double r = x * hpi_inv;
            n = ((int32_t)r + 0x800000) >> 24;   // nearest multiple of pi/2
            x = x - (double)n * hpi;
        }
        const double x2 = x * x;
        const double xs = ((n + 1) & 2) ? -x : x;           // sign of the sine in quadrants 0..3: + - - +
        // sine polynomial of the reduced argument
        const double x3 = xs * x2;
        const double s1 = S2 + x2 * S3;
        const double x7 = x3 * x2;
        const double sp0 = xs + x3 * S1;
        const float SP = (float)(sp0 + x7 * s1);
        // cosine polynomial; in quadrants 2 and 3 the library negates every coefficient, which negates the result exactly
        const double x4 = x2 * x2;
        const double c2 = C3 + x2 * C4;
        const double c1 = C0 + x2 * C1;
        const double x6 = x4 * x2;
        const double cp0 = c1 + x4 * C2;
        float CP = (float)(cp0 + x6 * c2);
        if (n & 2) CP = -CP;
        if (n & 1) { s = CP; c = SP; } else { s = SP; c = CP; }
        if (a < 0x1p-12f) { s = a; c = 1.0f; }
        return;
    }
#if !defined(__HIP_DEVICE_COMPILE__)
    double sd, cd;
    ::sincos((double)a, &sd, &cd);
    s = (float)sd; c = (float)cd;
#endif
}
// log and pow((float)E, y) (path_tracer_kernel.cu:566, 471-473) likewise follow the host C library's logf and
// powf (glibc 2.28 and later, the algorithms of Arm's optimized routines: a 16-entry table of (1/c, log c) and
// a cubic in double for the logarithm; for the power, log2 of the base -- a constant here, the base being
// (float)E -- times y, then 2^x from a 32-entry table of 2^(i/32) and a cubic).  Compared exhaustively with the
// library: logf for every positive normal float (2 130 706 432 values), powf((float)E, y) for every non-NaN
// float y (4 278 190 082 values) -- identical bits, with and without fused multiply-add.
PT_HD uint32_t f32_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; __builtin_memcpy(&u, &f, 4); return u;
#endif
}
PT_HD float f32_from_bits(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; __builtin_memcpy(&f, &u, 4); return f;
#endif
}
PT_HD uint64_t f64_bits(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u; __builtin_memcpy(&u, &d, 8); return u;
#endif
}
PT_HD double f64_from_bits(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d; __builtin_memcpy(&d, &u, 8); return d;
#endif
}

PT_HD float log_f(float x) {
    // (1/c, log c) for the 16 sub-intervals of [0x1.66p-1, 0x1.66p0), glibc's logf table
    const double T[16][2] = {
        {0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2},
        {0x1.49539f0f010b0p+0, -0x1.01eae7f513a67p-2}, {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3},
        {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8ea0p+0, -0x1.1aa2bc79c8100p-3},
        {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4},
        {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5}, {0x1p+0, 0x0p+0},
        {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5},  {0x1.ca4b31f026aa0p-1, 0x1.c5e53aa362eb4p-4},
        {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3},  {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d224770p-3},
        {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},  {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = f32_bits(x);
    if (ix == 0x3f800000u) return 0.0f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        // zero, subnormal, negative, infinity or NaN (the renderer's arguments, k * 2^-31, reach the zero case only)
        if (ix * 2u == 0u) return -__builtin_inff();
        if (ix == 0x7f800000u) return x;
        if ((ix & 0x80000000u) || ix * 2u >= 0xff000000u) return __builtin_nanf("");
        ix = f32_bits(x * 0x1p23f);
        ix -= 23u << 23;
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) % 16u);
    const int k = (int32_t)tmp >> 23;
    const double z = (double)f32_from_bits(ix - (tmp & 0xff800000u));
    const double r = z * T[i][0] - 1;
    const double y0 = T[i][1] + (double)k * Ln2;
    const double r2 = r * r;
    double y = A1 * r + A2;
    y = A0 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}
// pow((float)E, y) of path_tracer_kernel.cu:471-473: E narrowed to float, so the base is 2.7182817459106445.
PT_HD float pow_e_f(float y) {
    // 2^(i/32) as double bits, minus i << 47 (so that adding k << 47 for k = 32 q + i yields 2^(k/32))
    const uint64_t T[32] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
    const double C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
    const double SHIFT = 0x1.8p+52 / 32;
    // what the library's log2 step returns for the base (float)E: the one double for which every float y agrees
    const double LOG2_BASE = 0x1.715475968e9b3p+0;
    if (y != y) return y + y;
    const double xd = (double)y * LOG2_BASE;
    if (((f64_bits(xd) >> 47) & 0xffffu) >= (f64_bits(126.0) >> 47)) {
        if (xd > 0x1.fffffffd1d571p+6) return __builtin_inff();      // overflow
        if (xd <= -150.0) return 0.0f;                               // underflow
    }
    double kd = xd + SHIFT;
    const uint64_t ki = f64_bits(kd);
    kd -= SHIFT;
    const double r = xd - kd;
    const double s = f64_from_bits(T[ki % 32u] + (ki << 47));
    const double z = C0 * r + C1;
    const double r2 = r * r;
    double p = C2 * r + 1;
    p = z * r2 + p;
    return (float)(p * s);
}

// ---- constants (path_tracer_kernel.cu:54-57, medium.h:17) ------------------
#define PT_TWO_PI   6.2831853071795864769252867665590057683943
#define PT_SQRT13   0.5773502691896257645091487805019574556476
#define PT_AIR_IOR  1.000293f

// ---- randhash (:75-84) and thrust::default_random_engine ------------------
PT_HD uint32_t randhash(uint32_t a) {
    a = (a + 0x7ed55d16u) + (a << 12);
    a = (a ^ 0xc761c23cu) ^ (a >> 19);
    a = (a + 0x165667b1u) + (a << 5);
    a = (a + 0xd3a2646cu) ^ (a << 9);
    a = (a + 0xfd7046c5u) + (a << 3);
    a = (a ^ 0xb55a4f09u) ^ (a >> 16);
    return a;
}

// minstd_rand = linear_congruential_engine<uint32, 48271, 0, 2^31-1>.
struct Rng {
    uint32_t x;
    // seed(s): x = s mod m, and 0 is replaced by 1.  s < 2^32 < 2m + 2.
    PT_HD void seed(uint32_t s) {
        const uint32_t m = 2147483647u;
        if (s >= m) s -= m;
        if (s >= m) s -= m;
        x = (s == 0u) ? 1u : s;
    }
    // x <- 48271 x mod (2^31-1); 2^31 == 1 (mod m) folds the 47-bit product.
    PT_HD uint32_t next() {
        const uint32_t m = 2147483647u;
        uint64_t p = (uint64_t)x * 48271u;
        uint32_t r = (uint32_t)(p & m) + (uint32_t)(p >> 31);
        if (r >= m) r -= m;
        x = r;
        return r;
    }
    // uniform_real_distribution<float>(0,1): float(x - min) / (1.0f + float(max - min))
    // with min = 1, max = m - 1; the denominator is exactly 2^31.
    PT_HD float uniform() { return (float)(next() - 1u) * 4.656612873077392578125e-10f; }
};

// ---- camera (raycastFromCameraKernel, :125-191) ----------------------------
// Everything that depends on the camera alone is computed once on the host by
// camera_precompute() with the same float/double operations the reference
// performs per thread (:145-154).
struct CamPre {
    V3 position, hAxis, vAxis, middle, horizontal, vertical;
    float resx_m1, resy_m1;   // resolution.x - 1, resolution.y - 1  (:156-157)
    float stride;             // resolution.y (:135-136), or resolution.x with the width fix
    float focal, aperture;
    int   lens;               // apertureRadius > 0.00001 (:167)
    int   width, height;
};

PT_HD CamPre camera_precompute(V3 position, V3 view, V3 up, float resx, float resy,
                               float fovx, float fovy, float aperture, float focal,
                               int row_stride_is_width) {
    const double PI = 3.1415926535897932384626422832795028841971;
    CamPre c;
    view = normalize(view);
    c.hAxis = normalize(cross(view, up));
    c.vAxis = normalize(cross(c.hAxis, view));
    c.position = position;
    c.middle = position + view;
    c.horizontal = c.hAxis * (float)::tan(fovx * 0.5 * (PI / 180));
    c.vertical = c.vAxis * (float)::tan(-fovy * 0.5 * (PI / 180));
    c.resx_m1 = resx - 1;
    c.resy_m1 = resy - 1;
    c.stride = row_stride_is_width ? resx : resy;
    c.focal = focal;
    c.aperture = aperture;
    c.lens = (aperture > 0.00001) ? 1 : 0;
    c.width = (int)resx;
    c.height = (int)resy;
    return c;
}

PT_HD void camera_ray(const CamPre& c, int pixelIndex, uint32_t seed, V3& origin, V3& direction) {
    int y = (int)(pixelIndex / c.stride);
    int x = (int)(pixelIndex - (y * c.stride));

    Rng rng;
    rng.seed(randhash(seed) * randhash((uint32_t)pixelIndex) * randhash(seed));
    float jitterX = rng.uniform() - 0.5f;     // double subtract narrowed == float subtract
    float jitterY = rng.uniform() - 0.5f;

    float sx = (jitterX + x) / c.resx_m1;
    float sy = (jitterY + y) / c.resy_m1;

    V3 onePlane = c.middle + (((2 * sx) - 1) * c.horizontal) + (((2 * sy) - 1) * c.vertical);
    V3 onImagePlane = c.position + ((onePlane - c.position) * c.focal);

    V3 aperturePoint;
    if (c.lens) {
        float random1 = rng.uniform();
        float random2 = rng.uniform();
        float angle = (float)(PT_TWO_PI * (double)random1);
        float distance = c.aperture * sqrtf(random2);
        float s, co;
        sincos_f(angle, s, co);
        float apertureX = co * distance;
        float apertureY = s * distance;
        aperturePoint = c.position + (apertureX * c.hAxis) + (apertureY * c.vAxis);
    } else {
        aperturePoint = c.position;
    }
    direction = normalize(onImagePlane - aperturePoint);
    origin = aperturePoint;
}

// ---- intersections ---------------------------------------------------------
// findGroundPlaneIntersection (:200-219); the normal is (0,1,0) on both sides.
PT_HD float ground_t(float elevation, V3 o, V3 d) {
    if (d.y != 0) return (elevation - o.y) / d.y;
    return -1;
}
// findSphereIntersection (:250-281) without the normal.
PT_HD float sphere_t(V3 center, float radius, V3 o, V3 d) {
    V3 v = o - center;
    float vDotDirection = dot(v, d);
    float radicand = vDotDirection * vDotDirection - (dot(v, v) - radius * radius);
    if (radicand < 0) return -1;
    float squareRoot = sqrtf(radicand);
    float firstTerm = -vDotDirection;
    float t1 = firstTerm + squareRoot;
    float t2 = firstTerm - squareRoot;
    if (t1 < 0 && t2 < 0) return -1;
    if (t1 > 0 && t2 > 0) return (t1 < t2) ? t1 : t2;
    return (t1 > t2) ? t1 : t2;
}
// The caller's loop around sphere_t (:521-533): true, and bestT lowered, when this sphere's t
// satisfies 0 < t < bestT.  Same arithmetic as sphere_t; everything after the radicand test sits
// in one branch, which a wave skips altogether when none of its rays' lines meets the sphere.
// `radius2` is radius * radius, formed once per sphere instead of once per test (one float
// multiplication either way: the same bits).
PT_HD bool sphere_closer(V3 center, float radius2, V3 o, V3 d, float& bestT) {
    V3 v = o - center;
    float vDotDirection = dot(v, d);
    float radicand = vDotDirection * vDotDirection - (dot(v, v) - radius2);
    bool closer = false;
    if (radicand >= 0) {            // (a NaN radicand ends without a hit in sphere_t as well)
        float squareRoot = sqrtf(radicand);
        float firstTerm = -vDotDirection;
        float t1 = firstTerm + squareRoot;
        float t2 = firstTerm - squareRoot;
        float t;
        if (t1 < 0 && t2 < 0) t = -1;
        else if (t1 > 0 && t2 > 0) t = (t1 < t2) ? t1 : t2;
        else t = (t1 > t2) ? t1 : t2;
        if (t > 0 && t < bestT) { bestT = t; closer = true; }
    }
    return closer;
}
PT_HD V3 sphere_normal(V3 center, V3 o, V3 d, float t) {
    V3 p = o + t * d;
    return normalize(p - center);
}
// FindTriangleIntersect (:226-248) with edge1 = b - a and edge2 = c - a formed
// beforehand (one float subtraction each, so the same bits wherever it happens).
PT_HD bool triangle_t(V3 a, V3 edge1, V3 edge2, V3 o, V3 d, float& t) {
    V3 pvec = cross(d, edge2);
    float det = dot(edge1, pvec);
    if (det == 0.f) return false;
    float inv_det = 1.0f / det;
    V3 tvec = o - a;
    float bary1 = dot(tvec, pvec) * inv_det;
    V3 qvec = cross(tvec, edge1);
    float bary2 = dot(d, qvec) * inv_det;
    t = dot(edge2, qvec) * inv_det;
    return (bary1 >= 0.0f && bary2 >= 0.0f && (bary1 + bary2) <= 1.0f);
}

// Closest-hit encoding shared by the intersect and shade kernels.
#define PT_PRIM_MISS    (-1)
#define PT_PRIM_GROUND  (-2)
#define PT_PRIM_SPHERE  0x40000000   // | sphere index; triangles are plain indices
#define PT_T_START      123456789.0f // bestT initial value (:495)

// ---- materials --------------------------------------------------------------
struct Mat {
    V3 diffuse, emitted, specular;
    float ior;            // medium.refractiveIndex
    V3 sigma_a;           // medium absorption coefficient
    float sigma_s;        // medium reduced scattering coefficient
    int transmits;        // hasTransmission
};
// The in-kernel ground material (:611-619): defaults + diffuse (0.455,0.43,0.39).
PT_HD Mat ground_material() {
    Mat m;
    m.diffuse = v3(0.455f, 0.43f, 0.39f);
    m.emitted = v3(0, 0, 0);
    m.specular = v3(0, 0, 0);
    m.ior = PT_AIR_IOR;
    m.sigma_a = v3(0, 0, 0);
    m.sigma_s = 0;
    m.transmits = 0;
    return m;
}

struct ShadeParams {
    float  ray_bias;
    double min_ray_weight;
};

// ---- sampling / BSDF helpers -------------------------------------------------
// randomCosineWeightedDirectionInHemisphere (:352-374)
PT_HD V3 cosine_weighted_direction(V3 normal, float xi1, float xi2) {
    float up = sqrtf(xi1);
    float over = sqrtf(1 - up * up);
    float around = (float)((double)xi2 * PT_TWO_PI);
    V3 directionNotNormal;
    if ((double)fabsf(normal.x) < PT_SQRT13) directionNotNormal = v3(1, 0, 0);
    else if ((double)fabsf(normal.y) < PT_SQRT13) directionNotNormal = v3(0, 1, 0);
    else directionNotNormal = v3(0, 0, 1);
    V3 perpendicular1 = normalize(cross(normal, directionNotNormal));
    V3 perpendicular2 = cross(normal, perpendicular1);
    float s, c;
    sincos_f(around, s, c);
    return (up * normal) + (c * over * perpendicular1) + (s * over * perpendicular2);
}
// randomDirectionInSphere (:376-385)
PT_HD V3 direction_in_sphere(float xi1, float xi2) {
    float up = xi1 * 2 - 1;
    float over = sqrtf(1 - up * up);
    float around = (float)((double)xi2 * PT_TWO_PI);
    float s, c;
    sincos_f(around, s, c);
    return v3(up, c * over, s * over);
}
// computeReflectionDirection (:387-390): 2.0 * dot is a double product narrowed to float, i.e. an exact doubling.
PT_HD V3 reflection_direction(V3 normal, V3 incident) { return (2.0f * dot(normal, incident)) * normal - incident; }
// computeTransmissionDirection (:392-411); the radicand is formed in double.
PT_HD V3 transmission_direction(V3 normal, V3 incident, float iorIncident, float iorTransmitted) {
    float cosTheta1 = dot(normal, incident);
    float n1_n2 = iorIncident / iorTransmitted;
    double r = (double)n1_n2, c1 = (double)cosTheta1;
    float radicand = (float)(1 - (r * r) * (1 - (c1 * c1)));
    if (radicand < 0) return v3(0, 0, 0);
    float cosTheta2 = sqrtf(radicand);
    V3 minusIncident = -1.0f * incident;
    if (cosTheta1 > 0) return n1_n2 * minusIncident + (n1_n2 * cosTheta1 - cosTheta2) * normal;
    return n1_n2 * minusIncident + (n1_n2 * cosTheta1 + cosTheta2) * normal;
}
// computeFresnel (:413-456), reflection coefficient
PT_HD float fresnel_reflection(V3 normal, V3 incident, float n1, float n2, V3 transmissionDirection) {
    if ((double)length(transmissionDirection) <= 0.12345 || dot(normal, transmissionDirection) > 0)
        return 1;
    float ci = dot(normal, incident);
    float ct = dot(-1.0f * normal, transmissionDirection);
    float qs = (n1 * ci - n2 * ct) / (n1 * ci + n2 * ct);
    float qp = (n1 * ct - n2 * ci) / (n1 * ct + n2 * ci);
    float rs = qs * qs;            // pow(q,2) in double then narrowed == one float multiply
    float rp = qp * qp;
    return (rs + rp) * 0.5f;       // (rs+rp)/2.0 narrowed == exact halving
}
// computeBackgroundColor (:458-466)
PT_HD V3 background_color(V3 direction) {
    float invLen = 1.0f / sqrtf(1.5f);   // normalize((-0.5, 0.5, -1.0))
    V3 sun = v3(-0.5f * invLen, 0.5f * invLen, -1.0f * invLen);
    float position = (dot(direction, sun) + 1) / 2;
    V3 firstColor = v3(0.15f, 0.3f, 0.5f);
    V3 secondColor = v3(1.0f, 1.0f, 1.0f);
    return ((1 - position) * firstColor + position * secondColor) * 1.0f;
}
// computeTransmission (:468-475)
PT_HD V3 transmission_color(V3 absorption, float distance) {
    return v3(pow_e_f(-1 * absorption.x * distance),
              pow_e_f(-1 * absorption.y * distance),
              pow_e_f(-1 * absorption.z * distance));
}

// ---- one bounce after the closest hit is known (traceRayKernel, :560-716) --
// In/out: ray (o,d), throughput `thr` (notAbsorbedColors), `medium` = index of
// the material whose medium the path is inside, or -1 for air.
// `cur_sigma_a/_s` are that medium's coefficients (zeros for air).
// `hit` says whether anything was hit; `n`, `m`, `hit_mat_index` describe it.
// `add` receives what the reference adds to accumulatedColors this bounce
// (returned through has_add so exact zeros are not even added).
// `event` receives the branch the bounce left through (PT_EV_*, | PT_EV_KILLED when the weight
// test ended the path): what parity checks compare before they compare radiance.
// Returns true while the path stays alive.
#define PT_EV_SCATTER  1   // scattered inside the medium (:567-590)
#define PT_EV_REFLECT  2   // specular reflection (:665-676)
#define PT_EV_TRANSMIT 3   // refraction into / out of the material (:678-683)
#define PT_EV_DIFFUSE  4   // diffuse bounce (:685-703)
#define PT_EV_MISS     5   // nothing hit: sky (:709-716)
#define PT_EV_KILLED   8   // length(throughput) <= MIN_RAY_WEIGHT (:583, :705)
PT_HD bool shade_bounce(V3& o, V3& d, V3& thr, int& medium,
                        V3 cur_sigma_a, float cur_sigma_s,
                        bool hit, float bestT, V3 n, const Mat& m, int hit_mat_index,
                        int pixelIndex, int rayDepth, uint32_t seedOrPass,
                        const ShadeParams& P, V3& add, bool& has_add, int& event) {
    has_add = false;
    Rng rng;
    rng.seed(randhash(seedOrPass) * randhash((uint32_t)pixelIndex) * randhash((uint32_t)rayDepth));

    // absorption and scattering (:561-597)
    if (cur_sigma_s > 0 || (double)dot(cur_sigma_a, cur_sigma_a) > 0.00001) {
        float xi = rng.uniform();
        float scatteringDistance = -log_f(xi) / cur_sigma_s;
        if (scatteringDistance < bestT) {
            V3 no = o + scatteringDistance * d;
            float xi1 = rng.uniform();
            float xi2 = rng.uniform();
            d = direction_in_sphere(xi1, xi2);
            o = no;
            thr = thr * transmission_color(cur_sigma_a, scatteringDistance);
            const bool aliveS = !((double)length(thr) <= P.min_ray_weight);
            event = aliveS ? PT_EV_SCATTER : (PT_EV_SCATTER | PT_EV_KILLED);
            return aliveS;
        }
        thr = thr * transmission_color(cur_sigma_a, bestT);
    }

    if (!hit) {
        add = thr * background_color(d);       // (:712)
        has_add = true;
        event = PT_EV_MISS;
        return false;
    }

    V3 incident = -d;
    float iorIncident = PT_AIR_IOR;            // incident medium starts as air (:634-636)
    float iorTransmitted = m.ior;
    int mediumTransmitted = hit_mat_index;     // absorption/scattering of the far side
    bool backFace = (dot(n, incident) < 0);
    if (backFace) {
        n = n * -1.0f;
        float tmp = iorIncident; iorIncident = iorTransmitted; iorTransmitted = tmp;
        mediumTransmitted = -1;                // far side is air
    }

    V3 reflectionDirection = reflection_direction(n, incident);
    V3 transmissionDirection = transmission_direction(n, incident, iorIncident, iorTransmitted);
    V3 point = o + bestT * d;
    V3 biasVector = P.ray_bias * n;

    bool doSpecular = (m.ior > 1.0f);
    float roulette = rng.uniform();
    bool reflectFromSurface = doSpecular &&
        roulette < fresnel_reflection(n, incident, iorIncident, iorTransmitted, transmissionDirection);

    if (reflectFromSurface) {
        event = PT_EV_REFLECT;
        thr = thr * m.specular;
        o = point + biasVector;
        d = reflectionDirection;
    } else if (m.transmits) {
        event = PT_EV_TRANSMIT;
        medium = mediumTransmitted;
        o = point - biasVector;
        d = transmissionDirection;
    } else {
        event = PT_EV_DIFFUSE;
        // accumulated += notAbsorbed * emitted (:690).  A zero emitted colour with a
        // finite throughput adds exact zeros, which is skipped.
        bool zeroEmission = (m.emitted.x == 0.0f && m.emitted.y == 0.0f && m.emitted.z == 0.0f);
        bool finiteThr = (fabsf(thr.x) <= 3.402823466e38f) && (fabsf(thr.y) <= 3.402823466e38f) &&
                         (fabsf(thr.z) <= 3.402823466e38f);
        if (!(zeroEmission && finiteThr)) { add = thr * m.emitted; has_add = true; }
        thr = thr * m.diffuse;
        float r1 = rng.uniform();
        float r2 = rng.uniform();
        o = point + biasVector;
        d = cosine_weighted_direction(n, r1, r2);
    }
    const bool alive = !((double)length(thr) <= P.min_ray_weight);
    if (!alive) event |= PT_EV_KILLED;
    return alive;
}

}  // namespace pt
