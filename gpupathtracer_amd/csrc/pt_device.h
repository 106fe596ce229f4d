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
PT_HD float log_f(float x) { return (float)::log((double)x); }
// pow((float)E, y) of path_tracer_kernel.cu:471-473: E narrowed to float, so the
// base is 2.7182817459106445 and its natural log 0.99999996963214.
PT_HD float pow_e_f(float y) { return (float)::exp((double)y * 0.99999996963214); }

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

    V3 reflectionDirection = (2.0f * dot(n, incident)) * n - incident;   // (:387-390)
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
