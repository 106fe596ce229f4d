/*
 * kdtree_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Independent plain-C restatement of the kd-tree construction rules that the
 * library's host builder implements (design source: the reference's vendored
 * cukd library, which its renderer never calls and which cannot run here:
 * cukd/detail/node_chunk_array_kernels.cu:133-199,201-291,
 * cukd/detail/kdtree_kernels.cu:196-281, cukd/detail/small_node_array_kernels.cu:23-91,
 * cukd/kdtree.cu:247-258).  cukd's own output is racy on ties, so the layout is
 * defined by the deterministic rules below, and the library's builder must
 * reproduce this file's output bit for bit ("parity unpinned" by the reference,
 * which holds no kd-tree fixture).
 *
 * Rules
 *  - every triangle carries a box (initially its vertex bounds); the root region
 *    is the union of all boxes;
 *  - a node with more than `small_threshold` triangles at depth < 48 is a LARGE
 *    node: T = union of its triangles' boxes; for axis x,y,z in turn, if the gap
 *    between the region and T above T exceeds 12 % of the region's extent, an
 *    inner node at T.max[axis] with an empty right leaf is emitted, then likewise
 *    below T (empty left leaf at T.min[axis]); then T is split at the middle of
 *    its longest axis (x if dx>dy and dx>dz; else y if dy>dz; else z), pos =
 *    0.5f*extent + min.  A triangle goes left if min<pos or max<=pos, right if
 *    max>pos or min>=pos; one that goes both ways gets, per side, the bounds of
 *    the triangle clipped (Sutherland-Hodgman, double precision, region padded
 *    by 1e-5 x the root's largest extent, bounds rounded outward to float)
 *    intersected with its previous box and the child region, and is dropped
 *    from a side where nothing is left.  If both sides keep every triangle the
 *    node is handed to the SAH stage instead (at the FRONT of the work queue);
 *  - SAH stage: on entry the region is tightened to the triangles' boxes.  For
 *    axis x,y,z, side min,max, triangle j in order: candidate plane = that box
 *    face if strictly inside the region; nl = #(min<pos), nr = #(max>=pos), both
 *    non-zero; cost = (nl*area(left)+nr*area(right))/area(region) + Ct in float.
 *    The first candidate with the lowest cost wins; a leaf is made when there is
 *    none, when cost >= n, when the region has no area or at depth 56.  Children
 *    keep their boxes; left gets min<pos, right gets max>=pos;
 *  - nodes are numbered in processing order of a FIFO work queue; the two
 *    children of an inner node are allocated together when it is split; leaves
 *    append their triangle indices to `refs` in the order held.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { int tri; float lo[3], hi[3]; } kitem;
typedef struct { float lo[3], hi[3]; } kbox;

typedef struct kwork {
    uint32_t node;
    kitem* items;
    int n;
    kbox region;
    int depth;
    int sah;
    struct kwork* next;
} kwork;

typedef struct {
    uint32_t* nodes;   /* 2 words per node */
    size_t n_nodes, cap_nodes;
    int32_t* refs;
    size_t n_refs, cap_refs;
    int max_depth;
    kwork *head, *tail;
} kbuild;

static uint32_t float_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

static uint32_t new_pair(kbuild* b) {
    if (b->n_nodes + 2 > b->cap_nodes) {
        b->cap_nodes = b->cap_nodes * 2 + 64;
        b->nodes = (uint32_t*)realloc(b->nodes, b->cap_nodes * 8);
    }
    uint32_t left = (uint32_t)b->n_nodes;
    memset(b->nodes + 2 * b->n_nodes, 0, 16);
    b->n_nodes += 2;
    return left;
}
static uint32_t set_inner(kbuild* b, uint32_t node, int axis, float pos) {
    uint32_t left = new_pair(b);
    b->nodes[2 * node] = (left << 2) | (uint32_t)axis;
    b->nodes[2 * node + 1] = float_bits(pos);
    return left;
}
static void set_leaf(kbuild* b, uint32_t node, const kitem* items, int n, int depth) {
    if (b->n_refs + (size_t)n > b->cap_refs) {
        b->cap_refs = (b->n_refs + (size_t)n) * 2 + 64;
        b->refs = (int32_t*)realloc(b->refs, b->cap_refs * 4);
    }
    b->nodes[2 * node] = ((uint32_t)b->n_refs << 2) | 3u;
    b->nodes[2 * node + 1] = (uint32_t)n;
    for (int i = 0; i < n; i++) b->refs[b->n_refs++] = items[i].tri;
    if (depth > b->max_depth) b->max_depth = depth;
}
static void enqueue(kbuild* b, kwork* w, int front) {
    w->next = NULL;
    if (front) { w->next = b->head; b->head = w; if (!b->tail) b->tail = w; }
    else if (b->tail) { b->tail->next = w; b->tail = w; }
    else { b->head = b->tail = w; }
}
static kwork* make_work(uint32_t node, kitem* items, int n, const kbox* region, int depth, int sah) {
    kwork* w = (kwork*)malloc(sizeof(kwork));
    w->node = node; w->items = items; w->n = n; w->region = *region; w->depth = depth; w->sah = sah; w->next = NULL;
    return w;
}

static float surface(const float lo[3], const float hi[3]) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.f * (dx * dy + dx * dz + dy * dz);
}

/* bounds of triangle `v` clipped to `r` padded by `pad`; 0 if nothing is left */
static int clip_bounds(const float* v, const kbox* r, double pad, float lo[3], float hi[3]) {
    double p[16][3], q[16][3];
    int n = 3;
    for (int i = 0; i < 3; i++) for (int k = 0; k < 3; k++) p[i][k] = v[3 * i + k];
    for (int ax = 0; ax < 3 && n > 0; ax++) {
        for (int side = 0; side < 2 && n > 0; side++) {
            double pos = side == 0 ? (double)r->lo[ax] - pad : (double)r->hi[ax] + pad;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const double* a = p[i];
                const double* c = p[(i + 1) % n];
                int ain = side == 0 ? (a[ax] >= pos) : (a[ax] <= pos);
                int cin = side == 0 ? (c[ax] >= pos) : (c[ax] <= pos);
                if (ain) { q[m][0] = a[0]; q[m][1] = a[1]; q[m][2] = a[2]; m++; }
                if (ain != cin) {
                    double t = (pos - a[ax]) / (c[ax] - a[ax]);
                    for (int k = 0; k < 3; k++) q[m][k] = a[k] + (c[k] - a[k]) * t;
                    q[m][ax] = pos;
                    m++;
                }
            }
            n = m;
            memcpy(p, q, sizeof(double) * 3 * (size_t)n);
        }
    }
    if (n == 0) return 0;
    for (int k = 0; k < 3; k++) {
        double mn = p[0][k], mx = p[0][k];
        for (int i = 1; i < n; i++) { if (p[i][k] < mn) mn = p[i][k]; if (p[i][k] > mx) mx = p[i][k]; }
        float fl = (float)mn, fh = (float)mx;
        if ((double)fl > mn) fl = nextafterf(fl, -INFINITY);
        if ((double)fh < mx) fh = nextafterf(fh, INFINITY);
        lo[k] = fl; hi[k] = fh;
    }
    return 1;
}

static float fmax2(float a, float b) { return a > b ? a : b; }
static float fmin2(float a, float b) { return a < b ? a : b; }

/* Returns 0 on success.  Caller frees *out_nodes / *out_refs with orc_kd_free. */
int orc_kd_build(const float* verts, int n_tris, int small_threshold, float ct,
                 uint32_t** out_nodes, int* out_n_nodes, int32_t** out_refs, int* out_n_refs,
                 float bounds[6], int* out_max_depth) {
    kbuild b;
    memset(&b, 0, sizeof(b));
    kbox root;
    for (int k = 0; k < 3; k++) { root.lo[k] = INFINITY; root.hi[k] = -INFINITY; }
    kitem* items = (kitem*)malloc(sizeof(kitem) * (size_t)(n_tris > 0 ? n_tris : 1));
    for (int i = 0; i < n_tris; i++) {
        const float* v = verts + 9 * (size_t)i;
        items[i].tri = i;
        for (int k = 0; k < 3; k++) {
            items[i].lo[k] = fmin2(v[k], fmin2(v[3 + k], v[6 + k]));
            items[i].hi[k] = fmax2(v[k], fmax2(v[3 + k], v[6 + k]));
            root.lo[k] = fmin2(root.lo[k], items[i].lo[k]);
            root.hi[k] = fmax2(root.hi[k], items[i].hi[k]);
        }
    }
    if (n_tris == 0) for (int k = 0; k < 3; k++) { root.lo[k] = 0; root.hi[k] = 0; }
    float ext = 0;
    for (int k = 0; k < 3; k++) { bounds[k] = root.lo[k]; bounds[3 + k] = root.hi[k]; ext = fmax2(ext, root.hi[k] - root.lo[k]); }
    const double pad = 1e-5 * (double)ext;

    b.cap_nodes = 64; b.nodes = (uint32_t*)calloc(b.cap_nodes, 8); b.n_nodes = 1;
    enqueue(&b, make_work(0, items, n_tris, &root, 0, 0), 0);

    while (b.head) {
        kwork* w = b.head;
        b.head = w->next;
        if (!b.head) b.tail = NULL;
        const int n = w->n;
        if (n == 0) { set_leaf(&b, w->node, w->items, 0, w->depth); free(w->items); free(w); continue; }

        if (!w->sah && n > small_threshold && w->depth < 48) {
            kbox t;
            for (int k = 0; k < 3; k++) { t.lo[k] = INFINITY; t.hi[k] = -INFINITY; }
            for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) {
                t.lo[k] = fmin2(t.lo[k], w->items[i].lo[k]);
                t.hi[k] = fmax2(t.hi[k], w->items[i].hi[k]);
            }
            uint32_t node = w->node;
            int depth = w->depth;
            for (int a = 0; a < 3; a++) {
                float thr = 0.12f * (w->region.hi[a] - w->region.lo[a]);
                if ((w->region.hi[a] - t.hi[a]) - thr > 0) {
                    uint32_t l = set_inner(&b, node, a, t.hi[a]);
                    set_leaf(&b, l + 1, NULL, 0, depth + 1);
                    node = l; depth++;
                }
                if ((t.lo[a] - w->region.lo[a]) - thr > 0) {
                    uint32_t l = set_inner(&b, node, a, t.lo[a]);
                    set_leaf(&b, l, NULL, 0, depth + 1);
                    node = l + 1; depth++;
                }
            }
            float dx = t.hi[0] - t.lo[0], dy = t.hi[1] - t.lo[1], dz = t.hi[2] - t.lo[2];
            int axis; float pos;
            if (dx > dy) { if (dx > dz) { axis = 0; pos = 0.5f * dx + t.lo[0]; } else { axis = 2; pos = 0.5f * dz + t.lo[2]; } }
            else { if (dy > dz) { axis = 1; pos = 0.5f * dy + t.lo[1]; } else { axis = 2; pos = 0.5f * dz + t.lo[2]; } }
            kbox rl = t, rr = t;
            rl.hi[axis] = pos; rr.lo[axis] = pos;
            kitem* li = (kitem*)malloc(sizeof(kitem) * (size_t)n);
            kitem* ri = (kitem*)malloc(sizeof(kitem) * (size_t)n);
            int nl = 0, nr = 0;
            for (int i = 0; i < n; i++) {
                const kitem* it = &w->items[i];
                int goL = (it->lo[axis] < pos) || (it->hi[axis] <= pos);
                int goR = (it->hi[axis] > pos) || (it->lo[axis] >= pos);
                if (goL && goR) {
                    for (int side = 0; side < 2; side++) {
                        const kbox* cr = side == 0 ? &rl : &rr;
                        float clo[3], chi[3];
                        if (!clip_bounds(verts + 9 * (size_t)it->tri, cr, pad, clo, chi)) continue;
                        kitem c = *it;
                        int gone = 0;
                        for (int k = 0; k < 3; k++) {
                            c.lo[k] = fmax2(it->lo[k], fmax2(clo[k], cr->lo[k]));
                            c.hi[k] = fmin2(it->hi[k], fmin2(chi[k], cr->hi[k]));
                            if (c.lo[k] > c.hi[k]) gone = 1;
                        }
                        if (gone) continue;
                        if (side == 0) li[nl++] = c; else ri[nr++] = c;
                    }
                } else if (goL) li[nl++] = *it;
                else ri[nr++] = *it;
            }
            if (nl == n && nr == n) {
                free(li); free(ri);
                w->node = node; w->depth = depth; w->region = t; w->sah = 1;
                enqueue(&b, w, 1);
                continue;
            }
            uint32_t l = set_inner(&b, node, axis, pos);
            enqueue(&b, make_work(l, li, nl, &rl, depth + 1, 0), 0);
            enqueue(&b, make_work(l + 1, ri, nr, &rr, depth + 1, 0), 0);
            free(w->items); free(w);
            continue;
        }

        /* SAH stage */
        kbox r = w->region;
        if (!w->sah) {
            for (int k = 0; k < 3; k++) {
                float lo = INFINITY, hi = -INFINITY;
                for (int i = 0; i < n; i++) { lo = fmin2(lo, w->items[i].lo[k]); hi = fmax2(hi, w->items[i].hi[k]); }
                r.lo[k] = fmax2(r.lo[k], lo);
                r.hi[k] = fmin2(r.hi[k], hi);
            }
        }
        float area = surface(r.lo, r.hi);
        int best_axis = -1; float best_pos = 0, best_cost = INFINITY;
        if (area > 0 && w->depth < 56) {
            float inv = 1.f / area;
            for (int a = 0; a < 3; a++) for (int side = 0; side < 2; side++) for (int j = 0; j < n; j++) {
                float pos = side == 0 ? w->items[j].lo[a] : w->items[j].hi[a];
                if (!(pos > r.lo[a] && pos < r.hi[a])) continue;
                int cl = 0, cr = 0;
                for (int i = 0; i < n; i++) { cl += w->items[i].lo[a] < pos; cr += w->items[i].hi[a] >= pos; }
                if (cl == 0 || cr == 0) continue;
                float lhi[3] = {r.hi[0], r.hi[1], r.hi[2]}, rlo[3] = {r.lo[0], r.lo[1], r.lo[2]};
                lhi[a] = pos; rlo[a] = pos;
                float cost = inv * ((float)cl * surface(r.lo, lhi) + (float)cr * surface(rlo, r.hi)) + ct;
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_pos = pos; }
            }
        }
        if (best_axis < 0 || best_cost >= (float)n) {
            set_leaf(&b, w->node, w->items, n, w->depth);
            free(w->items); free(w);
            continue;
        }
        kbox rl = r, rr = r;
        rl.hi[best_axis] = best_pos; rr.lo[best_axis] = best_pos;
        kitem* li = (kitem*)malloc(sizeof(kitem) * (size_t)n);
        kitem* ri = (kitem*)malloc(sizeof(kitem) * (size_t)n);
        int nl = 0, nr = 0;
        for (int i = 0; i < n; i++) {
            if (w->items[i].lo[best_axis] < best_pos) li[nl++] = w->items[i];
            if (w->items[i].hi[best_axis] >= best_pos) ri[nr++] = w->items[i];
        }
        uint32_t l = set_inner(&b, w->node, best_axis, best_pos);
        enqueue(&b, make_work(l, li, nl, &rl, w->depth + 1, 1), 0);
        enqueue(&b, make_work(l + 1, ri, nr, &rr, w->depth + 1, 1), 0);
        free(w->items); free(w);
    }
    *out_nodes = b.nodes; *out_n_nodes = (int)b.n_nodes;
    *out_refs = b.refs ? b.refs : (int32_t*)malloc(4); *out_n_refs = (int)b.n_refs;
    *out_max_depth = b.max_depth;
    return 0;
}

void orc_kd_free(void* p) { free(p); }
