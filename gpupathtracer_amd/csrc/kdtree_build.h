// kdtree_build.h -- host kd-tree builder interface (plain C++).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace gptkd {

struct Node { uint32_t w0, w1; };

struct BuildParams {
    // <= this many triangles: SAH stage (cukd/kdtree.cu:76 uses 64).  The pair (24, 3) measured
    // fastest with the wave-cooperative traversal kernel on MI355X: a tree level costs a dependent
    // memory access, a leaf triangle is tested 64 at a time -- so slightly larger leaves under
    // median splits that reach further down.
    int   small_threshold = 24;
    float empty_ratio = 0.12f;    // large nodes: an empty margin wider than this share of the node extent is cut off
    float traversal_cost = 3.0f;  // constant term of the SAH cost (cukd uses 6, small_node_array_kernels.cu:78-79)
};

struct Tree {
    std::vector<Node> nodes;      // inner: w0 = left<<2 | axis, w1 = split bits; leaf: w0 = first_ref<<2 | 3, w1 = count
    std::vector<int32_t> refs;    // triangle indices, leaf by leaf
    float bounds[6];              // root box: min xyz, max xyz
    int max_depth;
    int n_leaves;
    int n_empty_leaves;
};

// verts: n_tris x 9 floats (p0, p1, p2).
void build(const float* verts, int n_tris, const BuildParams& bp, Tree& out);

// Expected traversal work of a tree for a random line through its root box, by the surface-area rule: every node
// counts with the chance that the line meets its region (region area / root area); returns
// node_cost * E[nodes visited] + tri_cost * E[triangles tested].
double expected_cost(const Tree& t, double node_cost, double tri_cost);

// The same tree -- word for word -- built on the current HIP device (kdtree_build_gpu.hip): the level-by-level
// decisions run as kernels, numbering by prefix sums in queue order.  false: a HIP call failed (*err says which).
bool build_gpu(const float* verts, int n_tris, const BuildParams& bp, Tree& out, std::string* err);

// Device storage order of the nodes: the same tree with its nodes renumbered so that
// every 128-byte line (16 nodes = 8 sibling pairs) holds a connected piece of the tree
// (a pair, its child pairs, their child pairs ... breadth-first until the line is
// full); pieces are stored back to back, so each spans at most two lines.  A root-to-leaf
// walk then touches a new line every ~3 levels instead of every level.  Sibling pairs
// stay adjacent and start at even indices; the root stays node 0.  The triangle
// references (refs) are untouched.
void treelet_layout(const Tree& in, std::vector<Node>& out_nodes);

// Device storage order of the triangle references: depth-first (left to right) over the
// leaves; `nodes` (in any order) get their leaves' first-reference fields rewritten.
void leaf_order_refs(std::vector<Node>& nodes, const std::vector<int32_t>& refs, std::vector<int32_t>& out_refs);

}  // namespace gptkd
