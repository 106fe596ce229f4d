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
// Unlike cukd (a GPU builder with racy tie-breaks) this builder is sequential
// and fully deterministic; ties go to the first candidate in the order
// x-min faces, x-max faces, y-min, y-max, z-min, z-max, each in triangle order.
//
// Output layout (see gpt.h): nodes numbered breadth-first, the two children of
// an inner node adjacent (left, left+1); leaves reference a contiguous run of
// `refs`.
#include "kdtree_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <deque>

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
const float kEmptyRatio = 0.12f;

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
    std::deque<Work> queue;
    queue.push_back(std::move(root));

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

    while (!queue.empty()) {
        Work w = std::move(queue.front());
        queue.pop_front();
        const int n = (int)w.items.size();

        if (n == 0) { make_leaf(w.node, w.items, w.depth); continue; }

        if (!w.small && n > bp.small_threshold && w.depth < kMaxDepth - 8) {
            // ---------------- large-node stage ----------------
            Box tb;
            for (int k = 0; k < 3; k++) { tb.mn[k] = INFINITY; tb.mx[k] = -INFINITY; }
            for (const Item& it : w.items)
                for (int k = 0; k < 3; k++) {
                    tb.mn[k] = std::min(tb.mn[k], it.mn[k]);
                    tb.mx[k] = std::max(tb.mx[k], it.mx[k]);
                }
            // empty-space cuts, thresholds from the incoming box
            uint32_t node = w.node;
            int depth = w.depth;
            for (int a = 0; a < 3; a++) {
                float thr = kEmptyRatio * (w.box.mx[a] - w.box.mn[a]);
                if ((w.box.mx[a] - tb.mx[a]) - thr > 0) {          // cut above: right child empty
                    uint32_t left = make_inner(node, a, tb.mx[a]);
                    make_leaf(left + 1, std::vector<Item>(), depth + 1);
                    node = left;
                    depth++;
                }
                if ((tb.mn[a] - w.box.mn[a]) - thr > 0) {          // cut below: left child empty
                    uint32_t left = make_inner(node, a, tb.mn[a]);
                    make_leaf(left, std::vector<Item>(), depth + 1);
                    node = left + 1;
                    depth++;
                }
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
            Work L, R;
            L.box = tb; L.box.mx[axis] = pos;
            R.box = tb; R.box.mn[axis] = pos;
            for (const Item& it : w.items) {
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
                        (side == 0 ? L : R).items.push_back(c);
                    }
                } else if (toL) {
                    L.items.push_back(it);
                } else {
                    R.items.push_back(it);
                }
            }
            if ((int)L.items.size() == n && (int)R.items.size() == n) {
                // the median separates nothing: hand the node to the SAH stage
                w.node = node;
                w.depth = depth;
                w.box = tb;
                w.small = true;
                queue.push_front(std::move(w));
                continue;
            }
            uint32_t left = make_inner(node, axis, pos);
            L.node = left; R.node = left + 1;
            L.depth = R.depth = depth + 1;
            L.small = R.small = false;
            queue.push_back(std::move(L));
            queue.push_back(std::move(R));
            continue;
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
        if (area > 0 && w.depth < kMaxDepth) {
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
        if (best_axis < 0 || best_cost >= (float)n) {
            make_leaf(w.node, w.items, w.depth);
            continue;
        }
        Work L, R;
        L.box = nb; L.box.mx[best_axis] = best_pos;
        R.box = nb; R.box.mn[best_axis] = best_pos;
        for (const Item& it : w.items) {
            if (it.mn[best_axis] < best_pos) L.items.push_back(it);
            if (it.mx[best_axis] >= best_pos) R.items.push_back(it);
        }
        uint32_t left = make_inner(w.node, best_axis, best_pos);
        L.node = left; R.node = left + 1;
        L.depth = R.depth = w.depth + 1;
        L.small = R.small = true;
        queue.push_back(std::move(L));
        queue.push_back(std::move(R));
    }
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
