// harness.cpp -- TEST INFRASTRUCTURE.  Compiles the library's per-path
// arithmetic (pt_device.h), kd traversal (kd_traverse.h) and host kd builder
// for the HOST so that their logic can be checked against the oracle on a CPU,
// without a GPU.  Nothing here is shipped or reachable from the product path;
// the product's kernels call the same inline functions on the device.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../gpupathtracer_amd/csrc/kd_traverse.h"
#include "../../gpupathtracer_amd/csrc/kdtree_build.h"
#include "../../gpupathtracer_amd/csrc/pt_device.h"
#include "../../gpupathtracer_amd/csrc/pt_helper_eval.h"
#include "../../include/gpt.h"

using namespace pt;

struct HostStack {
    uint32_t n[128]; float a[128], b[128]; int sp; int max_sp;
    void reset() { sp = 0; }
    void push(uint32_t node, float tmin, float tmax) { n[sp] = node; a[sp] = tmin; b[sp] = tmax; sp++; if (sp > max_sp) max_sp = sp; }
    bool pop(uint32_t& node, float& tmin, float& tmax) { if (sp == 0) return false; sp--; node = n[sp]; tmin = a[sp]; tmax = b[sp]; return true; }
};

struct HTree {
    gptkd::Tree tree;
    std::vector<gptkd::Node> dev_nodes;   // cache-line clustered order, as uploaded to the GPU
    std::vector<TriRec> tris;             // per triangle
    std::vector<TriRec> leaf_tris;        // per leaf reference
    KdView view;
};

static void make_recs(const float* verts, int n, std::vector<TriRec>& out) {
    out.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        const float* v = verts + 9 * (size_t)i;
        float* f = out[(size_t)i].f;
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2];
        f[3] = v[3] - v[0]; f[4] = v[4] - v[1]; f[5] = v[5] - v[2];
        f[6] = v[6] - v[0]; f[7] = v[7] - v[1]; f[8] = v[8] - v[2];
        memcpy(&f[9], &i, 4);
    }
}

extern "C" {

void* hh_tree_build(const float* verts, int n, int small_threshold, float traversal_cost) {
    HTree* h = new HTree();
    gptkd::BuildParams bp; bp.small_threshold = small_threshold; bp.traversal_cost = traversal_cost;
    gptkd::build(verts, n, bp, h->tree);
    make_recs(verts, n, h->tris);
    gptkd::treelet_layout(h->tree, h->dev_nodes);
    std::vector<int32_t> dev_refs;
    gptkd::leaf_order_refs(h->dev_nodes, h->tree.refs, dev_refs);
    h->leaf_tris.resize(dev_refs.size());
    for (size_t r = 0; r < dev_refs.size(); r++) h->leaf_tris[r] = h->tris[(size_t)dev_refs[r]];
    h->view.nodes = (const KdNode*)h->dev_nodes.data();
    h->view.leaf_tris = h->leaf_tris.data();
    h->view.refs = nullptr;
    float ext = 0;
    for (int k = 0; k < 3; k++) {
        h->view.bmin[k] = h->tree.bounds[k]; h->view.bmax[k] = h->tree.bounds[3 + k];
        float e = h->tree.bounds[3 + k] - h->tree.bounds[k]; if (e > ext) ext = e;
    }
    h->view.sigma = 2e-5f * ext;
    return h;
}
void hh_tree_free(void* p) { delete (HTree*)p; }
void hh_tree_info(void* p, int* n_nodes, int* n_refs, int* max_depth, int* n_leaves, int* n_empty) {
    HTree* h = (HTree*)p;
    *n_nodes = (int)h->tree.nodes.size(); *n_refs = (int)h->tree.refs.size();
    *max_depth = h->tree.max_depth; *n_leaves = h->tree.n_leaves; *n_empty = h->tree.n_empty_leaves;
}
void hh_tree_copy(void* p, uint32_t* nodes, int32_t* refs, float* bounds) {
    HTree* h = (HTree*)p;
    memcpy(nodes, h->tree.nodes.data(), h->tree.nodes.size() * 8);
    memcpy(refs, h->tree.refs.data(), h->tree.refs.size() * 4);
    memcpy(bounds, h->tree.bounds, 24);
}

// Closest triangle for n rays (6 floats each) through the tree, or brute force.
void hh_closest(void* p, const float* rays, int n, float t_init, int use_kd,
                int32_t* tri, float* t, uint64_t* stats /* nodes, tris, max stack */) {
    HTree* h = (HTree*)p;
    uint64_t sn = 0, stt = 0; int smax = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+:sn, stt) reduction(max:smax)
    for (int i = 0; i < n; i++) {
        V3 o = v3(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]);
        V3 d = v3(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
        float bt = t_init; int bi = -1; uint32_t cn = 0, ct = 0;
        if (use_kd) {
            HostStack st; st.max_sp = 0;
            kd_closest_hit<HostStack, true>(h->view, o, d, bt, bi, st, cn, ct);
            if (st.max_sp > smax) smax = st.max_sp;
        } else {
            brute_closest_hit<true>(h->tris.data(), (int)h->tris.size(), o, d, bt, bi, ct);
        }
        tri[i] = bi; t[i] = bt; sn += cn; stt += ct;
    }
    if (stats) { stats[0] = sn; stats[1] = stt; stats[2] = (uint64_t)smax; }
}

// One pass through the library's arithmetic on the host, pixel by pixel:
// camera_ray -> [closest hit -> shade_bounce] x depth.  Mirrors the kernels'
// data flow (materials deduplicated by the caller is not needed here: the
// material of the hit primitive is converted on the fly).
static Mat to_mat(const Material& m) {
    Mat r;
    r.diffuse = v3(m.diffuseColor.x, m.diffuseColor.y, m.diffuseColor.z);
    r.emitted = v3(m.emittedColor.x, m.emittedColor.y, m.emittedColor.z);
    r.specular = v3(m.specularColor.x, m.specularColor.y, m.specularColor.z);
    r.ior = m.medium.refractiveIndex;
    const AbsorptionAndScatteringProperties& as = m.medium.absorptionAndScatteringProperties;
    r.sigma_a = v3(as.absorptionCoefficient.x, as.absorptionCoefficient.y, as.absorptionCoefficient.z);
    r.sigma_s = as.reducedScatteringCoefficient;
    r.transmits = m.hasTransmission ? 1 : 0;
    return r;
}

static uint64_t g_hist[64]; static uint64_t g_maxnodes; static float g_worst[8];
void hh_hist(uint64_t* out, uint64_t* maxnodes, float* worst) { memcpy(out, g_hist, sizeof(g_hist)); *maxnodes = g_maxnodes; memcpy(worst, g_worst, 32); memset(g_hist,0,sizeof(g_hist)); g_maxnodes=0; }
int64_t hh_render_pass(void* tree_or_null, int numPolys, const Poly* polys, int numSpheres, const Sphere* spheres,
                       const int32_t* pixel_ids, int n, int counter, const Camera* cam,
                       const gpt_render_params* rp, float* out_rgb, int32_t* first_prim, float* first_t) {
    HTree* h = (HTree*)tree_or_null;
    CamPre cp = camera_precompute(v3(cam->position.x, cam->position.y, cam->position.z),
                                  v3(cam->view.x, cam->view.y, cam->view.z),
                                  v3(cam->up.x, cam->up.y, cam->up.z),
                                  cam->resolution.x, cam->resolution.y, cam->fov.x, cam->fov.y,
                                  cam->apertureRadius, cam->focalDistance, rp->row_stride_is_width);
    ShadeParams sp; sp.ray_bias = rp->ray_bias; sp.min_ray_weight = rp->min_ray_weight;
    std::vector<TriRec> recs;
    if (!h) {
        std::vector<float> verts((size_t)numPolys * 9);
        for (int i = 0; i < numPolys; i++) {
            memcpy(&verts[9 * (size_t)i], &polys[i].p0, 36);
        }
        make_recs(verts.data(), numPolys, recs);
    }
    int64_t total = 0;
    uint32_t seed = (uint32_t)counter;
#pragma omp parallel for schedule(dynamic, 64) reduction(+:total)
    for (int k = 0; k < n; k++) {
        int pix = pixel_ids[k];
        V3 o, d, thr = v3(1, 1, 1), acc = v3(0, 0, 0);
        int medium = -1;   // here: index into polys (>=0), spheres (0x40000000|i), -1 air
        camera_ray(cp, pix, seed, o, d);
        for (int depth = 0; depth < rp->max_depth; depth++) {
            total++;
            float bestT = PT_T_START; int prim = PT_PRIM_MISS;
            if (rp->ground_enabled) {
                float t = ground_t(rp->ground_elevation, o, d);
                if (t > 0) { bestT = t; prim = PT_PRIM_GROUND; }
            }
            for (int i = 0; i < numSpheres; i++) {
                float t = sphere_t(v3(spheres[i].position.x, spheres[i].position.y, spheres[i].position.z),
                                   spheres[i].radius, o, d);
                if (t > 0 && t < bestT) { bestT = t; prim = PT_PRIM_SPHERE | i; }
            }
            {
                int bi = -1; uint32_t c0 = 0, c1 = 0; float bt = bestT;
                if (h) { HostStack st; st.max_sp = 0; kd_closest_hit<HostStack, true>(h->view, o, d, bt, bi, st, c0, c1);
                    int b = 0; while ((1u << b) <= c0 && b < 63) b++;
                    #pragma omp critical
                    { g_hist[b]++; if (c0 > g_maxnodes) { g_maxnodes = c0; g_worst[0]=o.x; g_worst[1]=o.y; g_worst[2]=o.z; g_worst[3]=d.x; g_worst[4]=d.y; g_worst[5]=d.z; g_worst[6]=(float)depth; g_worst[7]=(float)c1; } } }
                else brute_closest_hit<false>(recs.data(), numPolys, o, d, bt, bi, c1);
                if (bi >= 0) { bestT = bt; prim = bi; }
            }
            if (depth == 0 && first_prim) { first_prim[k] = prim; first_t[k] = bestT; }
            V3 nrm = v3(0, 1, 0);
            Mat m = ground_material();
            int matIndex = -1;
            if (prim >= 0 && (prim & PT_PRIM_SPHERE)) {
                int si = prim & ~PT_PRIM_SPHERE;
                nrm = sphere_normal(v3(spheres[si].position.x, spheres[si].position.y, spheres[si].position.z), o, d, bestT);
                m = to_mat(spheres[si].material); matIndex = prim;
            } else if (prim >= 0) {
                nrm = v3(polys[prim].n.x, polys[prim].n.y, polys[prim].n.z);
                m = to_mat(polys[prim].material); matIndex = prim;
            }
            V3 sa = v3(0, 0, 0); float ss = 0;
            if (medium >= 0) {
                const Material& mm = (medium & PT_PRIM_SPHERE) ? spheres[medium & ~PT_PRIM_SPHERE].material : polys[medium].material;
                Mat q = to_mat(mm); sa = q.sigma_a; ss = q.sigma_s;
            }
            V3 add; bool has_add;
            int event = 0;
            bool alive = shade_bounce(o, d, thr, medium, sa, ss, prim != PT_PRIM_MISS, bestT, nrm, m, matIndex,
                                      pix, depth, seed, sp, add, has_add, event);
            if (has_add) acc = acc + add;
            if (!alive) break;
        }
        out_rgb[3 * k] = acc.x; out_rgb[3 * k + 1] = acc.y; out_rgb[3 * k + 2] = acc.z;
    }
    return total;
}

// pt::sincos_f against the C library's sinf / cosf for the floats with bit patterns first, first + step, ... <= last:
// returns the number of arguments for which either result differs in any bit.
long long hh_sincos_mismatches(uint32_t first, uint32_t last, uint32_t step) {
    long long bad = 0;
    const long long n = ((long long)last - (long long)first) / (long long)step + 1;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long k = 0; k < n; k++) {
        const uint32_t u = first + (uint32_t)(k * step);
        float a;
        memcpy(&a, &u, 4);
        float s, c;
        pt::sincos_f(a, s, c);
        volatile float av = a;
        const float gs = sinf(av), gc = cosf(av);
        if (memcmp(&s, &gs, 4) != 0 || memcmp(&c, &gc, 4) != 0) bad++;
    }
    return bad;
}

// pt::log_f against logf and pt::pow_e_f against powf((float)E, y) for the floats with bit patterns first,
// first + step, ... <= last (NaN results compare equal to each other).
long long hh_log_mismatches(uint32_t first, uint32_t last, uint32_t step) {
    long long bad = 0;
    const long long n = ((long long)last - (long long)first) / (long long)step + 1;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long k = 0; k < n; k++) {
        const uint32_t u = first + (uint32_t)(k * step);
        float a;
        memcpy(&a, &u, 4);
        volatile float av = a;
        const float m = pt::log_f(a), g = logf(av);
        if (memcmp(&m, &g, 4) != 0 && !(m != m && g != g)) bad++;
    }
    return bad;
}
long long hh_pow_e_mismatches(uint32_t first, uint32_t last, uint32_t step) {
    long long bad = 0;
    const long long n = ((long long)last - (long long)first) / (long long)step + 1;
    volatile float base = (float)2.7182818284590452353602874713526624977572;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long k = 0; k < n; k++) {
        const uint32_t u = first + (uint32_t)(k * step);
        float a;
        memcpy(&a, &u, 4);
        volatile float av = a;
        const float m = pt::pow_e_f(a), g = powf(base, av);
        if (memcmp(&m, &g, 4) != 0 && !(m != m && g != g)) bad++;
    }
    return bad;
}

// pt::Rng (the generator the kernels use), host build: n_draws uniforms per seed
void hh_rng_uniforms(const uint32_t* seeds, int n_seeds, int n_draws, float* out) {
    for (int s = 0; s < n_seeds; s++) {
        pt::Rng rng;
        rng.seed(seeds[s]);
        for (int k = 0; k < n_draws; k++) out[(size_t)s * n_draws + k] = rng.uniform();
    }
}

// The per-path helper functions on caller-given records (layouts: pt_helper_eval.h); -1: unknown function.
int hh_helper_eval(int fn, int n, const uint32_t* in, uint32_t* out) {
    const int wi = pt::helper_words_in(fn), wo = pt::helper_words_out(fn);
    if (wi == 0) return -1;
    for (int k = 0; k < n; k++) pt::helper_eval(fn, in + (size_t)k * wi, out + (size_t)k * wo);
    return 0;
}

void hh_camera_rays(const int32_t* pixel_ids, int n, int counter, const Camera* cam, int row_stride_is_width, float* out) {
    CamPre cp = camera_precompute(v3(cam->position.x, cam->position.y, cam->position.z),
                                  v3(cam->view.x, cam->view.y, cam->view.z),
                                  v3(cam->up.x, cam->up.y, cam->up.z),
                                  cam->resolution.x, cam->resolution.y, cam->fov.x, cam->fov.y,
                                  cam->apertureRadius, cam->focalDistance, row_stride_is_width);
    for (int k = 0; k < n; k++) {
        V3 o, d;
        camera_ray(cp, pixel_ids[k], (uint32_t)counter, o, d);
        out[6 * k] = o.x; out[6 * k + 1] = o.y; out[6 * k + 2] = o.z;
        out[6 * k + 3] = d.x; out[6 * k + 4] = d.y; out[6 * k + 5] = d.z;
    }
}

}  // extern "C"
