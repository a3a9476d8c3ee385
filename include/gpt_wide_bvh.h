/*
 * gpt_wide_bvh.h - the 4-wide BVH of GPT_TRAVERSAL_WIDE4 ("stage B"): its structure, how it is
 * derived from the reference's binary tree, and the order in which a ray walks it.  Shared by the host side of libgpt and
 * by the CPU oracle, like gpt_traversal.h: this file is the SPECIFICATION both follow; the walk itself is written twice
 * (oracle/pt_oracle.c: one ray at a time; csrc/pt_kernel.hip: one lane per ray, all four boxes of a node tested by that lane,
 * the ray's stack in an LDS slice of its lane).
 *
 * Why: the reference's tree (binned SAH, <= 4 primitives per leaf, src/bvh.cpp:38-173) is walked one 40-byte node per
 * dependent memory access, left child first (src/pathtracer.cu:214-255).  On scenes that do not fit LDS that chain of
 * dependent fetches is the bound.  A wide node holds the boxes of four (grand)children of the SAME tree in
 * one 128-byte record: one fetch tests four boxes, the tree is half as deep, and the children are entered nearest first.
 * The reference has no such structure (src/sbvh.h is an empty class); the results contract stays its arithmetic:
 * every box is a box of the reference's tree (bit for bit), the box test is bbox.h:77-96 and the triangle test
 * mesh.h:45-67, unchanged.  What changes is the ORDER of the tests, hence
 *   - which boxes a shrinking tmax culls, and
 *   - which of two hits at EXACTLY the same distance wins: here the larger primitive index (in BVH order), whatever the
 *     order of the tests - the reference keeps whichever it tested last (mesh.h:63 accepts tt == tmax).
 * CPU oracle and GPU kernel agree bit for bit in this mode; agreement with the reference order is statistical (tests:
 * relative RMS <= 1e-4; measured: identical films).
 *
 * Structure.  Wide node w = the (up to) four subtrees obtained from a binary inner node by repeatedly replacing, in place,
 * the inner candidate of largest surface area by its two children (left to right order is kept: children stay in the
 * reference's preorder order).  A child is
 *   count < 0   another wide node, ref = its index
 *   count > 0   a leaf: primitives ref .. ref + count - 1 (BVH order), count <= GPT_WIDE_LEAF_MAX
 *   count = 0   empty (never hit)
 * A reference leaf with more than GPT_WIDE_LEAF_MAX primitives (the reference makes ONE leaf of any set whose box is thinner
 * than 1e-4, however large: bvh.cpp:43) becomes a small subtree of index ranges; their boxes are the exact min / max of
 * the triangles' vertices, i.e. sub-boxes of the reference's leaf box.
 *
 * Walk of one ray (closest hit; any-hit rays stop at the first accepted triangle):
 *   stack <- {}, current <- wide node 0
 *   wide node:  test the four boxes against the ray's CURRENT interval; push the hit children so that they pop in order
 *               of increasing entry distance tn (gpt_wide_key: near-ties in slot order); pop
 *   leaf:       test its triangles in index order, each against the ray's CURRENT interval; an accepted distance that is
 *               not NaN and nearer than the interval's end becomes the interval's end; then pop
 *   an accepted triangle replaces the best hit when it is nearer, or exactly as near with a larger primitive index.
 */
#ifndef GPT_WIDE_BVH_H
#define GPT_WIDE_BVH_H

#include "gpt_types.h"

#define GPT_TRAVERSAL_WIDE4 2
#define GPT_WIDE_LEAF_MAX 16
#define GPT_WIDE_STACK_MAX 256    /* deepest stack the walk may need: 3 * depth + 1 must not exceed it (wide depth <= 85) */

typedef struct {
    float bmin[3], bmax[3];
    int32_t ref;
    int32_t count;
} gpt_wide_child;                 /* 32 bytes */
typedef struct {
    gpt_wide_child c[4];
} gpt_wide_node;                  /* 128 bytes */

/* stack / "current" entries: a wide node (its index) or a leaf range */
#define GPT_WIDE_NONE 0xffffffffu
static inline uint32_t gpt_wide_leaf_entry(int32_t first, int32_t count) { return 0x80000000u | ((uint32_t)(count - 1) << 27) | (uint32_t)first; }
static inline int gpt_wide_entry_is_leaf(uint32_t e) { return (e >> 31) != 0; }
static inline int32_t gpt_wide_entry_first(uint32_t e) { return (int32_t)(e & 0x07ffffffu); }
static inline int32_t gpt_wide_entry_count(uint32_t e) { return (int32_t)((e >> 27) & 15u) + 1; }

typedef struct {
    const gpt_bvh_node *nodes;
    const gpt_primitive *prims;
    gpt_wide_node *out;
    int32_t n_out, cap;
    int32_t max_depth;
    int overflow;
} gpt_wide_builder;

static inline float gpt_wide_area(const gpt_bvh_node *n)
{
    const float dx = n->fmax.x - n->fmin.x, dy = n->fmax.y - n->fmin.y, dz = n->fmax.z - n->fmin.z;
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

static inline void gpt_wide_set_box(gpt_wide_child *c, const float lo[3], const float hi[3])
{
    for (int a = 0; a < 3; ++a) { c->bmin[a] = lo[a]; c->bmax[a] = hi[a]; }
}

/* child `c` <- primitives first .. first + count - 1; ranges longer than GPT_WIDE_LEAF_MAX become a subtree */
static inline void gpt_wide_emit_range(gpt_wide_builder *b, gpt_wide_child *c, int32_t first, int32_t count, int depth)
{
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
    for (int32_t i = first; i < first + count; ++i) {
        const gpt_triangle *t = &b->prims[i].triangle;
        const gpt_float3 *v[3] = {&t->v1.v, &t->v2.v, &t->v3.v};
        for (int k = 0; k < 3; ++k) {
            const float p[3] = {v[k]->x, v[k]->y, v[k]->z};
            for (int a = 0; a < 3; ++a) {
                if (p[a] < lo[a]) lo[a] = p[a];
                if (p[a] > hi[a]) hi[a] = p[a];
            }
        }
    }
    gpt_wide_set_box(c, lo, hi);
    if (count <= GPT_WIDE_LEAF_MAX) {
        c->ref = first;
        c->count = count;
        return;
    }
    if (b->n_out >= b->cap) { b->overflow = 1; c->ref = 0; c->count = 0; return; }
    const int32_t w = b->n_out++;
    if (depth + 1 > b->max_depth) b->max_depth = depth + 1;
    c->ref = w;
    c->count = -1;
    gpt_wide_node node;
    __builtin_memset(&node, 0, sizeof(node));
    int32_t at = first;
    for (int k = 0; k < 4; ++k) {
        const int32_t n = (count - (at - first) + (3 - k)) / (4 - k);      /* four nearly equal parts */
        if (n > 0) gpt_wide_emit_range(b, &node.c[k], at, n, depth + 1);
        at += n;
    }
    b->out[w] = node;
}

/* wide node for the binary inner node `i`; returns its index */
static inline int32_t gpt_wide_build_node(gpt_wide_builder *b, int32_t i, int depth)
{
    if (b->n_out >= b->cap) { b->overflow = 1; return 0; }
    const int32_t w = b->n_out++;
    if (depth > b->max_depth) b->max_depth = depth;
    int32_t cand[4];
    int n = 0;
    if (b->nodes[i].is_leaf) {
        cand[n++] = i;                                   /* a tree that is one leaf */
    } else {
        cand[n++] = i + 1;
        cand[n++] = b->nodes[i].second_child_offset;
    }
    while (n < 4) {
        int pick = -1;
        float best = -1.f;
        for (int k = 0; k < n; ++k) {
            if (b->nodes[cand[k]].is_leaf) continue;
            const float a = gpt_wide_area(&b->nodes[cand[k]]);
            if (pick < 0 || a > best) { pick = k; best = a; }     /* ties, NaN: the earlier candidate */
        }
        if (pick < 0) break;
        const int32_t c = cand[pick];
        for (int k = n; k > pick + 1; --k) cand[k] = cand[k - 1];
        cand[pick] = c + 1;
        cand[pick + 1] = b->nodes[c].second_child_offset;
        ++n;
    }
    gpt_wide_node node;
    __builtin_memset(&node, 0, sizeof(node));
    for (int k = 0; k < n; ++k) {
        const gpt_bvh_node *src = &b->nodes[cand[k]];
        gpt_wide_child *c = &node.c[k];
        const float lo[3] = {src->fmin.x, src->fmin.y, src->fmin.z}, hi[3] = {src->fmax.x, src->fmax.y, src->fmax.z};
        if (!src->is_leaf) {
            gpt_wide_set_box(c, lo, hi);
            c->ref = gpt_wide_build_node(b, cand[k], depth + 1);
            c->count = -1;
        } else {
            const int32_t count = (src->start >= 0 && src->end >= src->start) ? src->end - src->start + 1 : 0;
            if (count > GPT_WIDE_LEAF_MAX) {
                gpt_wide_emit_range(b, c, src->start, count, depth);
            } else {
                gpt_wide_set_box(c, lo, hi);                /* the reference's own leaf box */
                c->ref = count > 0 ? src->start : 0;
                c->count = count;
            }
        }
    }
    b->out[w] = node;
    return w;
}

/* capacity that always suffices for gpt_wide_build */
static inline int32_t gpt_wide_capacity(int32_t n_nodes, int32_t n_prims) { return n_nodes + n_prims / 4 + 8; }

/* Builds the wide tree of a reference BVH (nodes in the reference's preorder, prims in BVH order).  Returns the number of
 * wide nodes (0 for an empty tree, -1 if `cap` was too small); *depth_out = number of wide levels. */
static inline int32_t gpt_wide_build(const gpt_bvh_node *nodes, int32_t n_nodes, const gpt_primitive *prims, gpt_wide_node *out,
                                     int32_t cap, int32_t *depth_out)
{
    gpt_wide_builder b;
    b.nodes = nodes; b.prims = prims; b.out = out; b.n_out = 0; b.cap = cap; b.max_depth = 0; b.overflow = 0;
    if (depth_out) *depth_out = 0;
    if (n_nodes <= 0) return 0;
    gpt_wide_build_node(&b, 0, 1);
    if (b.overflow) return -1;
    if (depth_out) *depth_out = b.max_depth;
    return b.n_out;
}

/* The order key of hit child `slot` (0..3) with entry distance tn: children pop in order of increasing key.  The key is the
 * bit pattern of max(tn, +0) - every distance that is not positive (the ray starts inside the box; -0; NaN) counts as +0, and
 * positive floats sort like their bit patterns - with its two lowest bits replaced by the slot: keys of one node are all
 * different, and distances that agree up to their two last mantissa bits pop in slot order.  (On the GPU: one v_max_f32
 * against +0, which returns +0 for a NaN and for -0, and one v_and_or_b32.) */
static inline uint32_t gpt_wide_key(float tn, int slot)
{
    uint32_t u = 0;
    if (tn > 0.0f) __builtin_memcpy(&u, &tn, 4);
    return (u & ~3u) | (uint32_t)slot;
}

#endif /* GPT_WIDE_BVH_H */
