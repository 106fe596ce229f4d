// kdtree_build.cpp -- host kd-tree builder (plain C++, no HIP).
//
// Design source for the split rules: the vendored cukd library of the
// reference (never called by its renderer):
//   large nodes  cukd/detail/node_chunk_array_kernels.cu:133-199 (left/right
//                tags, empty-space cut at 12 % of the parent extent),
//                cukd/detail/kdtree_kernels.cu:196-281 (cut order x-max, x-min,
//                y-max, y-min, z-max, z-min; spatial median of the tight box on
//                its longest axis), :201-291 of node_chunk_array_kernels.cu
//                (straddlers get clipped boxes);
//   small nodes  cukd/detail/small_node_array_kernels.cu:23-91 (SAH over the
//                triangle-box faces, cost = (nL*AL + nR*AR)/A + Ct, leaf when
//                the best cost >= the triangle count, cukd/kdtree.cu:247-258).
// Unlike cukd (a GPU builder with racy tie-breaks) this builder is fully
// deterministic -- its worker threads only take decisions that do not depend on
// each other, numbering happens in queue order -- and ties go to the first
// candidate in the order
// x-min faces, x-max faces, y-min, y-max, z-min, z-max, each in triangle order.
//
// Output layout (see gpt.h): nodes numbered breadth-first, the two children of
// an inner node adjacent (left, left+1); leaves reference a contiguous run of
// `refs`.
#include "kdtree_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <atomic>
#include <deque>
#include <cstdlib>
#include <thread>
#include <vector>

namespace gptkd {

namespace {

struct Item {
    int tri;
    float mn[3], mx[3];
};

struct Box {
    float mn[3], mx[3];
};

struct Work {
    uint32_t node;          // index of the node to fill in
    std::vector<Item> items;
    Box box;                // region handed down by the parent
    int depth;
    bool small;             // already in the SAH stage
};

const int kMaxDepth = 56;

inline uint32_t fbits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

inline float box_area(const float mn[3], const float mx[3]) {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return 2.f * (dx * dy + dx * dz + dy * dz);
}

// Sutherland-Hodgman clip of a triangle against a box padded by `pad`, in
// double; returns false when nothing is left.  The resulting bounds are
// rounded OUTWARD to float.
bool clipped_bounds(const float* v9, const Box& b, double pad, float out_mn[3], float out_mx[3]) {
    double poly[16][3], tmp[16][3];
    int n = 3;
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) poly[i][k] = v9[3 * i + k];
    for (int axis = 0; axis < 3 && n > 0; axis++) {
        for (int side = 0; side < 2 && n > 0; side++) {
            double pos = side == 0 ? (double)b.mn[axis] - pad : (double)b.mx[axis] + pad;
            int m = 0;
            for (int i = 0; i < n; i++) {
                const double* a = poly[i];
                const double* c = poly[(i + 1) % n];
                bool ain = side == 0 ? (a[axis] >= pos) : (a[axis] <= pos);
                bool cin = side == 0 ? (c[axis] >= pos) : (c[axis] <= pos);
                if (ain) { tmp[m][0] = a[0]; tmp[m][1] = a[1]; tmp[m][2] = a[2]; m++; }
                if (ain != cin) {
                    double t = (pos - a[axis]) / (c[axis] - a[axis]);
                    for (int k = 0; k < 3; k++) tmp[m][k] = a[k] + (c[k] - a[k]) * t;
                    tmp[m][axis] = pos;
                    m++;
                }
            }
            n = m;
            for (int i = 0; i < n; i++)
                for (int k = 0; k < 3; k++) poly[i][k] = tmp[i][k];
        }
    }
    if (n == 0) return false;
    for (int k = 0; k < 3; k++) {
        double lo = poly[0][k], hi = poly[0][k];
        for (int i = 1; i < n; i++) { lo = std::min(lo, poly[i][k]); hi = std::max(hi, poly[i][k]); }
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
        if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
        out_mn[k] = flo;
        out_mx[k] = fhi;
    }
    return true;
}

}  // namespace

namespace {

// What processing one queue entry decides, recorded so that it can be applied later: the
// decisions of all entries of a tree level are independent of each other (they are taken by
// worker threads), while node numbers and leaf reference runs are handed out afterwards, one
// entry after the other in queue order -- the order a single-threaded breadth-first build uses.
struct Cut { int axis; float pos; bool empty_right; };
struct Decision {
    int n_cuts = 0;
    Cut cuts[6];
    bool split = false;     // false: the (remaining) node becomes a leaf holding `w.items`
    int axis = 0;
    float pos = 0;
    Work L, R;              // children of a split (items, box, stage); node / depth set on commit
};

// Runs fn(part, begin, end) over `parts` contiguous slices of [0, n), one thread each.
template <class F>
void parallel_slices(size_t n, int parts, F fn) {
    std::vector<std::thread> pool;
    for (int p = 1; p < parts; p++) pool.emplace_back([=]() { fn(p, n * (size_t)p / (size_t)parts, n * (size_t)(p + 1) / (size_t)parts); });
    fn(0, (size_t)0, n / (size_t)parts);
    for (std::thread& t : pool) t.join();
}

// `inner_threads` > 1: the entry is one of the few, huge ones at the top of the tree; its
// per-triangle loops are cut into slices whose results are joined in slice order, so the outcome
// does not depend on the thread count.
void decide(const float* verts, const BuildParams& bp, double clip_pad, Work& w, Decision& d, int inner_threads) {
    const int n = (int)w.items.size();
    if (n == 0) return;
    int depth = w.depth;
    if (!w.small && n > bp.small_threshold && w.depth < kMaxDepth - 8) {
        // ---------------- large-node stage ----------------
        const int slices = (inner_threads > 1 && n >= 100000) ? inner_threads : 1;
        Box tb;
        for (int k = 0; k < 3; k++) { tb.mn[k] = INFINITY; tb.mx[k] = -INFINITY; }
        {
            std::vector<Box> part((size_t)slices, tb);
            parallel_slices((size_t)n, slices, [&](int p, size_t b, size_t e) {
                Box pb = part[(size_t)p];
                for (size_t i = b; i < e; i++)
                    for (int k = 0; k < 3; k++) {
                        pb.mn[k] = std::min(pb.mn[k], w.items[i].mn[k]);
                        pb.mx[k] = std::max(pb.mx[k], w.items[i].mx[k]);
                    }
                part[(size_t)p] = pb;
            });
            for (const Box& pb : part)          // min / max: exact, any order
                for (int k = 0; k < 3; k++) {
                    tb.mn[k] = std::min(tb.mn[k], pb.mn[k]);
                    tb.mx[k] = std::max(tb.mx[k], pb.mx[k]);
                }
        }
        // empty-space cuts, thresholds from the incoming box
        for (int a = 0; a < 3; a++) {
            float thr = bp.empty_ratio * (w.box.mx[a] - w.box.mn[a]);
            if ((w.box.mx[a] - tb.mx[a]) - thr > 0) { d.cuts[d.n_cuts++] = {a, tb.mx[a], true}; depth++; }   // right child empty
            if ((tb.mn[a] - w.box.mn[a]) - thr > 0) { d.cuts[d.n_cuts++] = {a, tb.mn[a], false}; depth++; }  // left child empty
        }
        // spatial median of the tight box on its longest axis
        float dx = tb.mx[0] - tb.mn[0], dy = tb.mx[1] - tb.mn[1], dz = tb.mx[2] - tb.mn[2];
        int axis;
        float pos;
        if (dx > dy) {
            if (dx > dz) { axis = 0; pos = 0.5f * dx + tb.mn[0]; }
            else { axis = 2; pos = 0.5f * dz + tb.mn[2]; }
        } else {
            if (dy > dz) { axis = 1; pos = 0.5f * dy + tb.mn[1]; }
            else { axis = 2; pos = 0.5f * dz + tb.mn[2]; }
        }
        Work& L = d.L;
        Work& R = d.R;
        L.box = tb; L.box.mx[axis] = pos;
        R.box = tb; R.box.mn[axis] = pos;
        {
            std::vector<std::vector<Item>> partL((size_t)slices), partR((size_t)slices);
            parallel_slices((size_t)n, slices, [&](int p, size_t b, size_t e) {
                std::vector<Item>& Lp = partL[(size_t)p];
                std::vector<Item>& Rp = partR[(size_t)p];
                for (size_t i = b; i < e; i++) {
                    const Item& it = w.items[i];
                    bool toL = (it.mn[axis] < pos) || (it.mx[axis] <= pos);
                    bool toR = (it.mx[axis] > pos) || (it.mn[axis] >= pos);
                    if (toL && toR) {
                        // straddler: clipped bounds in each child, never larger than before
                        const float* v = verts + 9 * (size_t)it.tri;
                        for (int side = 0; side < 2; side++) {
                            const Box& cb = side == 0 ? L.box : R.box;
                            Item c = it;
                            float cmn[3], cmx[3];
                            if (!clipped_bounds(v, cb, clip_pad, cmn, cmx)) continue;
                            bool empty = false;
                            for (int k = 0; k < 3; k++) {
                                c.mn[k] = std::max(it.mn[k], std::max(cmn[k], cb.mn[k]));
                                c.mx[k] = std::min(it.mx[k], std::min(cmx[k], cb.mx[k]));
                                if (c.mn[k] > c.mx[k]) empty = true;
                            }
                            if (empty) continue;
                            (side == 0 ? Lp : Rp).push_back(c);
                        }
                    } else if (toL) {
                        Lp.push_back(it);
                    } else {
                        Rp.push_back(it);
                    }
                }
            });
            if (slices == 1) {
                L.items.swap(partL[0]);
                R.items.swap(partR[0]);
            } else {
                for (int p = 0; p < slices; p++) {      // slice order = triangle order
                    L.items.insert(L.items.end(), partL[(size_t)p].begin(), partL[(size_t)p].end());
                    R.items.insert(R.items.end(), partR[(size_t)p].begin(), partR[(size_t)p].end());
                }
            }
        }
        if (!((int)L.items.size() == n && (int)R.items.size() == n)) {
            d.split = true;
            d.axis = axis;
            d.pos = pos;
            L.small = R.small = false;
            return;
        }
        // the median separates nothing: the node (after its cuts) goes through the SAH stage
        L.items.clear();
        R.items.clear();
        w.box = tb;
        w.small = true;
    }

    // ---------------- SAH stage ----------------
    Box nb = w.box;
    if (!w.small) {
        // entering the stage: tighten the region to the triangles' boxes
        for (int k = 0; k < 3; k++) {
            float lo = INFINITY, hi = -INFINITY;
            for (const Item& it : w.items) { lo = std::min(lo, it.mn[k]); hi = std::max(hi, it.mx[k]); }
            nb.mn[k] = std::max(nb.mn[k], lo);
            nb.mx[k] = std::min(nb.mx[k], hi);
        }
    }
    float area = box_area(nb.mn, nb.mx);
    int best_axis = -1;
    float best_pos = 0, best_cost = INFINITY;
    if (area > 0 && depth < kMaxDepth) {
        float inv_area = 1.f / area;
        for (int a = 0; a < 3; a++) {
            for (int side = 0; side < 2; side++) {
                for (int j = 0; j < n; j++) {
                    float pos = side == 0 ? w.items[(size_t)j].mn[a] : w.items[(size_t)j].mx[a];
                    if (!(pos > nb.mn[a] && pos < nb.mx[a])) continue;
                    int nl = 0, nr = 0;
                    for (const Item& it : w.items) {
                        nl += (it.mn[a] < pos) ? 1 : 0;
                        nr += (it.mx[a] >= pos) ? 1 : 0;
                    }
                    if (nl == 0 || nr == 0) continue;
                    float lmx[3] = {nb.mx[0], nb.mx[1], nb.mx[2]};
                    float rmn[3] = {nb.mn[0], nb.mn[1], nb.mn[2]};
                    lmx[a] = pos;
                    rmn[a] = pos;
                    float al = box_area(nb.mn, lmx);
                    float ar = box_area(rmn, nb.mx);
                    float cost = inv_area * ((float)nl * al + (float)nr * ar) + bp.traversal_cost;
                    if (cost < best_cost) { best_cost = cost; best_axis = a; best_pos = pos; }
                }
            }
        }
    }
    if (best_axis < 0 || best_cost >= (float)n) return;     // leaf
    d.split = true;
    d.axis = best_axis;
    d.pos = best_pos;
    d.L.box = nb; d.L.box.mx[best_axis] = best_pos;
    d.R.box = nb; d.R.box.mn[best_axis] = best_pos;
    for (const Item& it : w.items) {
        if (it.mn[best_axis] < best_pos) d.L.items.push_back(it);
        if (it.mx[best_axis] >= best_pos) d.R.items.push_back(it);
    }
    d.L.small = d.R.small = true;
}

int build_threads() {
    int t = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("GPT_BUILD_THREADS")) t = atoi(e);
    return std::max(1, std::min(t, 16));
}

}  // namespace

void build(const float* verts, int n_tris, const BuildParams& bp, Tree& out) {
    out.nodes.clear();
    out.refs.clear();
    out.max_depth = 0;
    out.n_leaves = 0;
    out.n_empty_leaves = 0;

    Work root;
    root.node = 0;
    root.depth = 0;
    root.small = false;
    root.items.resize((size_t)n_tris);
    for (int k = 0; k < 3; k++) { root.box.mn[k] = INFINITY; root.box.mx[k] = -INFINITY; }
    for (int i = 0; i < n_tris; i++) {
        Item& it = root.items[(size_t)i];
        it.tri = i;
        const float* v = verts + 9 * (size_t)i;
        for (int k = 0; k < 3; k++) {
            it.mn[k] = std::min(v[k], std::min(v[3 + k], v[6 + k]));
            it.mx[k] = std::max(v[k], std::max(v[3 + k], v[6 + k]));
            root.box.mn[k] = std::min(root.box.mn[k], it.mn[k]);
            root.box.mx[k] = std::max(root.box.mx[k], it.mx[k]);
        }
    }
    if (n_tris == 0)
        for (int k = 0; k < 3; k++) { root.box.mn[k] = 0; root.box.mx[k] = 0; }
    for (int k = 0; k < 3; k++) { out.bounds[k] = root.box.mn[k]; out.bounds[3 + k] = root.box.mx[k]; }

    float ext = 0;
    for (int k = 0; k < 3; k++) ext = std::max(ext, root.box.mx[k] - root.box.mn[k]);
    const double clip_pad = 1e-5 * (double)ext;

    out.nodes.push_back({0u, 0u});

    auto make_leaf = [&](uint32_t node, const std::vector<Item>& items, int depth) {
        out.nodes[node].w0 = ((uint32_t)out.refs.size() << 2) | 3u;
        out.nodes[node].w1 = (uint32_t)items.size();
        for (const Item& it : items) out.refs.push_back(it.tri);
        out.max_depth = std::max(out.max_depth, depth);
        out.n_leaves++;
        if (items.empty()) out.n_empty_leaves++;
    };
    auto make_inner = [&](uint32_t node, int axis, float pos) -> uint32_t {
        uint32_t left = (uint32_t)out.nodes.size();
        out.nodes.push_back({0u, 0u});
        out.nodes.push_back({0u, 0u});
        out.nodes[node].w0 = (left << 2) | (uint32_t)axis;
        out.nodes[node].w1 = fbits(pos);
        return left;
    };

    // Breadth-first, one tree level at a time: `level` holds the queue entries of the level in
    // queue order.
    const int n_threads = build_threads();
    const std::vector<Item> no_items;
    std::vector<Work> level, next;
    level.push_back(std::move(root));
    while (!level.empty()) {
        std::vector<Decision> dec(level.size());
        // ---- decide (parallel) ----
        size_t total_items = 0;
        for (const Work& w : level) total_items += w.items.size();
        const int nt = total_items < 20000 ? 1 : (int)std::min<size_t>((size_t)n_threads, level.size());
        if (nt < n_threads / 2 || nt <= 1) {
            // few entries (the top of the tree, or a tiny tree): one after the other, threads inside
            const int inner = total_items < 20000 ? 1 : n_threads;
            for (size_t i = 0; i < level.size(); i++) decide(verts, bp, clip_pad, level[i], dec[i], inner);
        } else {
            std::atomic<size_t> next_entry(0);
            auto worker = [&]() {
                for (;;) {
                    const size_t i = next_entry.fetch_add(1);
                    if (i >= level.size()) break;
                    decide(verts, bp, clip_pad, level[i], dec[i], 1);
                }
            };
            std::vector<std::thread> pool;
            for (int t = 1; t < nt; t++) pool.emplace_back(worker);
            worker();
            for (std::thread& t : pool) t.join();
        }
        // ---- commit (queue order) ----
        next.clear();
        for (size_t i = 0; i < level.size(); i++) {
            Work& w = level[i];
            Decision& d = dec[i];
            uint32_t node = w.node;
            int depth = w.depth;
            for (int c = 0; c < d.n_cuts; c++) {
                const uint32_t left = make_inner(node, d.cuts[c].axis, d.cuts[c].pos);
                if (d.cuts[c].empty_right) { make_leaf(left + 1, no_items, depth + 1); node = left; }
                else { make_leaf(left, no_items, depth + 1); node = left + 1; }
                depth++;
            }
            if (!d.split) { make_leaf(node, w.items, depth); continue; }
            const uint32_t left = make_inner(node, d.axis, d.pos);
            d.L.node = left; d.R.node = left + 1;
            d.L.depth = d.R.depth = depth + 1;
            next.push_back(std::move(d.L));
            next.push_back(std::move(d.R));
        }
        level.swap(next);
    }
}

double expected_cost(const Tree& t, double node_cost, double tri_cost) {
    struct Entry { uint32_t node; double lo[3], hi[3]; };
    auto area = [](const double lo[3], const double hi[3]) {
        const double dx = std::max(hi[0] - lo[0], 0.0), dy = std::max(hi[1] - lo[1], 0.0), dz = std::max(hi[2] - lo[2], 0.0);
        return 2.0 * (dx * dy + dy * dz + dx * dz);
    };
    if (t.nodes.empty()) return 0.0;
    Entry root;
    root.node = 0;
    for (int k = 0; k < 3; k++) { root.lo[k] = t.bounds[k]; root.hi[k] = t.bounds[3 + k]; }
    const double a0 = area(root.lo, root.hi);
    if (!(a0 > 0.0)) return 0.0;
    double e_nodes = 0.0, e_tris = 0.0;
    std::vector<Entry> stack;
    stack.push_back(root);
    while (!stack.empty()) {
        const Entry e = stack.back();
        stack.pop_back();
        const Node& n = t.nodes[e.node];
        const double p = area(e.lo, e.hi) / a0;
        e_nodes += p;
        const uint32_t axis = n.w0 & 3u;
        if (axis == 3u) { e_tris += p * (double)n.w1; continue; }
        float split;
        std::memcpy(&split, &n.w1, 4);
        Entry l = e, r = e;
        l.node = n.w0 >> 2; l.hi[axis] = split;
        r.node = (n.w0 >> 2) + 1u; r.lo[axis] = split;
        stack.push_back(l);
        stack.push_back(r);
    }
    return node_cost * e_nodes + tri_cost * e_tris;
}

void treelet_layout(const Tree& in, std::vector<Node>& out) {
    const std::vector<Node>& src = in.nodes;
    out.clear();
    const uint32_t kPairsPerLine = 8;
    const Node pad = {3u, 0u};                       // an empty leaf
    // A "pair" is identified by the old index of its left node; the root is a pair of
    // its own whose right slot is padding (old index UINT32_MAX marks "no node").
    struct Pair { uint32_t left_old; bool is_root; };
    std::deque<Pair> roots;                          // treelet roots still to be placed
    roots.push_back({0u, true});
    std::vector<uint32_t> new_index(src.size(), 0u); // old node index -> new node index
    std::vector<std::pair<uint32_t, uint32_t>> placed;   // (old, new) in placement order
    while (!roots.empty()) {
        // Pieces are laid back to back (no padding): a piece of up to 8 pairs spans at most
        // two lines.  Only the first line is aligned (the root pair starts it).
        std::deque<Pair> frontier;
        frontier.push_back(roots.back());              // depth-first over pieces: a subtree's pieces
        roots.pop_back();                              // end up next to each other in memory
        uint32_t used = 0;
        while (!frontier.empty() && used < kPairsPerLine) {
            Pair p = frontier.front();
            frontier.pop_front();
            const uint32_t at = (uint32_t)out.size();
            const uint32_t olds[2] = {p.left_old, p.is_root ? UINT32_MAX : p.left_old + 1u};
            for (int k = 0; k < 2; k++) {
                if (olds[k] == UINT32_MAX) { out.push_back(pad); continue; }
                new_index[olds[k]] = at + (uint32_t)k;
                placed.push_back({olds[k], at + (uint32_t)k});
                out.push_back(src[olds[k]]);
                if ((src[olds[k]].w0 & 3u) != 3u) frontier.push_back({src[olds[k]].w0 >> 2, false});
            }
            used++;
        }
        for (auto it = frontier.rbegin(); it != frontier.rend(); ++it) roots.push_back(*it);   // new pieces
    }
    // rewrite child links
    for (const auto& pn : placed) {
        Node& n = out[pn.second];
        if ((n.w0 & 3u) != 3u) {
            const uint32_t old_left = n.w0 >> 2;
            n.w0 = (new_index[old_left] << 2) | (n.w0 & 3u);
        }
    }
}

// Leaf-ordered copy of the triangle references: leaves visited left to right (depth first),
// so leaves that are neighbours in space are neighbours in memory.  Rewrites the leaves'
// first-reference fields of `nodes` (any node order) to point into `out_refs`.
void leaf_order_refs(std::vector<Node>& nodes, const std::vector<int32_t>& refs, std::vector<int32_t>& out_refs) {
    out_refs.clear();
    out_refs.reserve(refs.size());
    std::vector<uint32_t> stack;
    stack.push_back(0u);
    while (!stack.empty()) {
        const uint32_t n = stack.back();
        stack.pop_back();
        Node& w = nodes[n];
        if ((w.w0 & 3u) == 3u) {
            const uint32_t first = w.w0 >> 2, count = w.w1;
            const uint32_t at = (uint32_t)out_refs.size();
            for (uint32_t k = 0; k < count; k++) out_refs.push_back(refs[first + k]);
            w.w0 = (at << 2) | 3u;
        } else {
            const uint32_t left = w.w0 >> 2;
            stack.push_back(left + 1u);
            stack.push_back(left);
        }
    }
}

}  // namespace gptkd
