// pt_helper_eval.h -- the per-path helper functions of pt_device.h on caller-given inputs, one record per call
// (parity checks only: the host harness and the library's gpt_debug_helpers kernel both run THIS code, so the
// host build and the GPU are compared on the same expressions the render kernels inline).
//
// Record layouts (32-bit words in / out) follow oracle/ref_build/helpers_dump.cpp, the program that runs the
// reference's own text (path_tracer_kernel.cu:75-84, 194-475) on the same inputs.  Output words the library's
// functions do not compute -- the ground plane's constant normal, the barycentrics of a triangle test, the
// transmission coefficient 1 - R -- are left as PT_HELPER_UNSET.
#pragma once
#include "pt_device.h"

namespace pt {

enum HelperFn {
    HELPER_RANDHASH = 0, HELPER_POSITION, HELPER_GROUND, HELPER_TRIANGLE, HELPER_COSINE, HELPER_SPHERE_DIR,
    HELPER_REFLECT, HELPER_TRANSMIT, HELPER_FRESNEL, HELPER_SURFACE, HELPER_BACKGROUND, HELPER_TRANSMISSION,
    HELPER_COUNT
};
#define PT_HELPER_UNSET 0xffffffffu

PT_HD int helper_words_in(int fn) {
    const int w[HELPER_COUNT] = {1, 7, 7, 15, 5, 2, 6, 8, 14, 8, 3, 4};
    return fn >= 0 && fn < HELPER_COUNT ? w[fn] : 0;
}
PT_HD int helper_words_out(int fn) {
    const int w[HELPER_COUNT] = {1, 3, 4, 4, 3, 3, 3, 3, 2, 8, 3, 3};
    return fn >= 0 && fn < HELPER_COUNT ? w[fn] : 0;
}

PT_HD V3 helper_v3(const uint32_t* w) { return v3(f32_from_bits(w[0]), f32_from_bits(w[1]), f32_from_bits(w[2])); }
PT_HD void helper_put3(uint32_t* w, V3 v) { w[0] = f32_bits(v.x); w[1] = f32_bits(v.y); w[2] = f32_bits(v.z); }

PT_HD void helper_eval(int fn, const uint32_t* in, uint32_t* out) {
    switch (fn) {
    case HELPER_RANDHASH: out[0] = randhash(in[0]); break;
    case HELPER_POSITION:      // positionAlongRay (:194-197), as shade_bounce forms the hit point
        helper_put3(out, helper_v3(in) + f32_from_bits(in[6]) * helper_v3(in + 3));
        break;
    case HELPER_GROUND:
        out[0] = f32_bits(ground_t(f32_from_bits(in[0]), helper_v3(in + 1), helper_v3(in + 4)));
        out[1] = out[2] = out[3] = PT_HELPER_UNSET;
        break;
    case HELPER_TRIANGLE: {    // edges formed as the scene upload forms them (b - a, c - a)
        const V3 a = helper_v3(in), b = helper_v3(in + 3), c = helper_v3(in + 6);
        float t = f32_from_bits(0x7fc0deadu);
        const bool hit = triangle_t(a, b - a, c - a, helper_v3(in + 9), helper_v3(in + 12), t);
        out[0] = hit ? 1u : 0u;
        out[1] = f32_bits(t);
        out[2] = out[3] = PT_HELPER_UNSET;
        break;
    }
    case HELPER_COSINE:
        helper_put3(out, cosine_weighted_direction(helper_v3(in), f32_from_bits(in[3]), f32_from_bits(in[4])));
        break;
    case HELPER_SPHERE_DIR: helper_put3(out, direction_in_sphere(f32_from_bits(in[0]), f32_from_bits(in[1]))); break;
    case HELPER_REFLECT: helper_put3(out, reflection_direction(helper_v3(in), helper_v3(in + 3))); break;
    case HELPER_TRANSMIT:
        helper_put3(out, transmission_direction(helper_v3(in), helper_v3(in + 3), f32_from_bits(in[6]), f32_from_bits(in[7])));
        break;
    case HELPER_FRESNEL:
        out[0] = f32_bits(fresnel_reflection(helper_v3(in), helper_v3(in + 3), f32_from_bits(in[6]), f32_from_bits(in[7]),
                                             helper_v3(in + 11)));
        out[1] = PT_HELPER_UNSET;
        break;
    case HELPER_SURFACE: {     // the chain of shade_bounce at a surface (:652-665)
        const V3 n = helper_v3(in), i = helper_v3(in + 3);
        const float n1 = f32_from_bits(in[6]), n2 = f32_from_bits(in[7]);
        const V3 R = reflection_direction(n, i);
        const V3 T = transmission_direction(n, i, n1, n2);
        helper_put3(out, R);
        helper_put3(out + 3, T);
        out[6] = f32_bits(fresnel_reflection(n, i, n1, n2, T));
        out[7] = PT_HELPER_UNSET;
        break;
    }
    case HELPER_BACKGROUND: helper_put3(out, background_color(helper_v3(in))); break;
    case HELPER_TRANSMISSION: helper_put3(out, transmission_color(helper_v3(in), f32_from_bits(in[3]))); break;
    default: break;
    }
}

}  // namespace pt
