/*
 * gpt_wide8_bvh.h - the compressed 8-wide BVH of GPT_TRAVERSAL_WIDE8: its structure, how it is derived from the reference's
 * binary tree, and the walk of one ray.  Shared by the host side of libgpt and by the CPU oracle, like gpt_wide_bvh.h: this file
 * is the SPECIFICATION both follow, the walk is written twice (oracle/pt_oracle.c: one ray at a time; csrc/pt_kernel.hip: one lane
 * per ray).
 *
 * Why: on scenes in global memory the drain is bound by the NUMBER of trips a ray needs and by the round trip
 * of each (its fetches, then its arithmetic), not by lanes per trip nor by instruction count - round 4 measured that.  The 4-wide
 * walk needs 33.7 node trips and ~14 leaf trips per sample on the config-5 stand-in (two triangles per leaf trip).  This structure
 * cuts both: a node holds EIGHT children in 80 bytes (five 16-byte fetches instead of seven for four children), and a leaf trip
 * tests a WHOLE leaf (<= 4 triangles of 36 bytes).  It follows Ylitie, Karras and Laine, "Efficient Incoherent Ray Traversal on
 * GPUs Through Compressed Wide BVHs" (HPG 2017) in its three ideas - child boxes quantised to 8 bits against the node's own box,
 * children visited in an order fixed by the ray's octant instead of sorted by distance, and a traversal stack of child GROUPS (one
 * entry per node: its inner children still to visit) instead of single children - re-laid-out for a wave64 machine where one lane
 * owns one ray and a trip serves node lanes and leaf lanes with one set of fetches.  The reference has no such structure
 * (src/sbvh.h is an empty class).
 *
 * What stays the reference's: the tree the nodes are collapsed from (src/bvh.cpp:38-173), and the triangle test
 * (src/mesh.h:45-67: same operations, same order, same records v1 / e1 / e2).  What changes:
 *   - the boxes: a child's box is the reference's box ENLARGED to the node's 8-bit grid (never shrunk: checked exactly in double
 *     precision by the builder), and the slab test is t = fma(q, step / d, (p - o) / d) per plane.  A larger box can only ADD node
 *     visits and triangle tests; every triangle a reference-order ray accepts lies in boxes this walk enters (up to the rounding of
 *     the slab test itself, which the reference's own test has too: bbox.h:77-96).
 *   - the ORDER of the tests, as in GPT_TRAVERSAL_WIDE4: hence which boxes a shrinking interval culls, and which of two hits at
 *     EXACTLY the same distance wins - here the larger index in this tree's triangle order.
 * CPU oracle and GPU kernel agree bit for bit in this mode; agreement with the reference order is north_star's 1e-4 relative RMS
 * (measured: see DESIGN.md).
 *
 * Structure.  Node w holds up to eight children in SLOTS 0..7: the (up to) eight subtrees obtained from a binary inner node by
 * repeatedly replacing the inner candidate of largest surface area by its two children (gpt_wide_bvh.h's rule), as long as the leaf
 * children hold at most GPT_WIDE8_NODE_TRIS triangles together.  A child is
 *   inner   bit s of imask set; the inner children of a node are CONSECUTIVE records from child_base on, in slot order
 *   leaf    <= GPT_WIDE8_LEAF_MAX triangles; triangle k of slot s has bit 4 s + k of `valid` set.  The triangles of a node's leaf
 *           children are CONSECUTIVE in this tree's own triangle order (tri_order[]: position -> primitive index in BVH order),
 *           slot by slot from tri_base on
 *   empty   neither
 * A reference leaf with more than GPT_WIDE8_LEAF_MAX primitives (bvh.cpp:43 makes ONE leaf of any set whose box is thinner than
 * 1e-4) is treated as a binary subtree of index ranges halved until they fit; their boxes are the exact min / max of the vertices.
 * Slots: the children are placed so that, for a ray whose direction has sign bits oct = (dx < 0) | (dy < 0) << 1 | (dz < 0) << 2,
 * ascending (slot ^ oct) is roughly front to back: slot bit a = 1 means "towards +a from the node's centre" (greedy assignment of
 * the largest centroid projections first, gpt_wide8_assign_slots).
 *
 * Quantisation.  p = the lo corner of the union of the children's boxes; per axis a biased exponent e (step = 2^(e-127), the
 * smallest that covers the extent with 255 steps); child planes qlo = floor, qhi = ceil of (plane - p) / step, then corrected
 * until p + qlo step <= lo and p + qhi step >= hi hold EXACTLY (double arithmetic: both sides are exact there).  Empty slots get
 * qlo = 255, qhi = 0 on every axis.
 *
 * Walk of one ray (closest hit; any-hit rays stop at the first accepted triangle):
 *   group <- {node 0 as the only child of a virtual parent}, stack <- {}
 *   loop: if the triangle mask is not empty: take its lowest slot's leaf: test its triangles in order, each against the ray's
 *           CURRENT interval (an accepted distance that is not NaN and nearer than the interval's end becomes the end); clear
 *         else: if the group is empty pop one from the stack (none: done); remove the group's child of lowest (slot ^ oct);
 *           test that node's eight boxes against the CURRENT interval (gpt_wide8_slab); the hit inner children form the new
 *           group (the old one is pushed first if it is not empty), the hit leaf children's triangles the triangle mask
 *   an accepted triangle replaces the best hit when it is nearer, or exactly as near with a larger index in tri_order.
 */
#ifndef GPT_WIDE8_BVH_H
#define GPT_WIDE8_BVH_H

#include <math.h>
#include <string.h>
#include "gpt_types.h"

#define GPT_TRAVERSAL_WIDE8 3
#define GPT_WIDE8_LEAF_MAX 4
#define GPT_WIDE8_NODE_TRIS 32     /* triangles of all leaf children of one node: the `valid` mask has 4 bits per slot */
#define GPT_WIDE8_STACK_MAX 64     /* one group per level: deeper trees have no 8-wide form */

typedef struct {
    float p[3];                    /* origin of the node's grid */
    uint8_t e[3];                  /* per axis: step = 2^(e - 127), i.e. the float with bit pattern e << 23 */
    uint8_t imask;                 /* bit s: slot s holds an inner node */
    uint32_t child_base;           /* index of the first inner child (consecutive, in slot order) */
    uint32_t tri_base;             /* position in tri_order of the first triangle of this node's leaf children */
    uint32_t valid;                /* bit 4 s + k: leaf slot s has a k-th triangle */
    uint32_t pad;
    uint8_t qlo[3][8], qhi[3][8];  /* [axis][slot] */
} gpt_wide8_node;                  /* 80 bytes = five 16-byte fetches */

/* the slab test of one child.  s[a] = step[a] * inv_dir[a], b[a] = (p[a] - o[a]) * inv_dir[a] (per node), near / far plane by the
 * SIGN BIT of inv_dir[a].  fmaxf / fminf drop a NaN operand (v_max_f32 / v_min_f32): an axis the ray is parallel to constrains
 * nothing.  Hit unless the entry is PROVABLY behind the exit (a comparison with a NaN counts as a hit). */
static inline float gpt_w8_max(float a, float b) { return (a != a) ? b : (b != b) ? a : (a > b ? a : b); }
static inline float gpt_w8_min(float a, float b) { return (a != a) ? b : (b != b) ? a : (a < b ? a : b); }
static inline int gpt_wide8_slab(const gpt_wide8_node *n, int slot, const float s[3], const float b[3], const int neg[3], float tmax)
{
    float tn[3], tf[3];
    for (int a = 0; a < 3; ++a) {
        const float lo = fmaf((float)n->qlo[a][slot], s[a], b[a]), hi = fmaf((float)n->qhi[a][slot], s[a], b[a]);
        tn[a] = neg[a] ? hi : lo;
        tf[a] = neg[a] ? lo : hi;
    }
    const float t0 = gpt_w8_max(gpt_w8_max(gpt_w8_max(tn[0], tn[1]), tn[2]), 0.0f);
    const float t1 = gpt_w8_min(gpt_w8_min(gpt_w8_min(tf[0], tf[1]), tf[2]), tmax);
    return !(t0 > t1);
}

/* ---- builder ------------------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t node;                  /* binary node index, or -1: a range of an oversized reference leaf */
    int32_t first, count;          /* range (node < 0) */
    float lo[3], hi[3];
} gpt_w8_cand;

typedef struct {
    const gpt_bvh_node *nodes;
    const gpt_primitive *prims;
    gpt_wide8_node *out;
    int32_t *tri_order;
    int32_t n_out, cap, n_tris;
    int32_t max_depth;
    int overflow;
} gpt_w8_builder;

static inline void gpt_w8_range_box(const gpt_primitive *prims, int32_t first, int32_t count, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; ++a) { lo[a] = __builtin_inff(); hi[a] = -__builtin_inff(); }
    for (int32_t i = first; i < first + count; ++i) {
        const gpt_triangle *t = &prims[i].triangle;
        const gpt_float3 *v[3] = {&t->v1.v, &t->v2.v, &t->v3.v};
        for (int k = 0; k < 3; ++k) {
            const float q[3] = {v[k]->x, v[k]->y, v[k]->z};
            for (int a = 0; a < 3; ++a) {
                if (q[a] < lo[a]) lo[a] = q[a];
                if (q[a] > hi[a]) hi[a] = q[a];
            }
        }
    }
}

static inline gpt_w8_cand gpt_w8_cand_node(const gpt_w8_builder *b, int32_t i)
{
    gpt_w8_cand c;
    const gpt_bvh_node *n = &b->nodes[i];
    c.node = i; c.first = 0; c.count = 0;
    c.lo[0] = n->fmin.x; c.lo[1] = n->fmin.y; c.lo[2] = n->fmin.z;
    c.hi[0] = n->fmax.x; c.hi[1] = n->fmax.y; c.hi[2] = n->fmax.z;
    if (n->is_leaf) {
        c.first = n->start;
        c.count = (n->start >= 0 && n->end >= n->start) ? n->end - n->start + 1 : 0;
        if (c.count > GPT_WIDE8_LEAF_MAX) c.node = -1;          /* an oversized leaf: from here on a range (the reference's box stays) */
    }
    return c;
}
static inline gpt_w8_cand gpt_w8_cand_range(const gpt_w8_builder *b, int32_t first, int32_t count)
{
    gpt_w8_cand c;
    c.node = -1; c.first = first; c.count = count;
    gpt_w8_range_box(b->prims, first, count, c.lo, c.hi);
    return c;
}
/* a candidate that can still be opened: a binary inner node, or a range of more than GPT_WIDE8_LEAF_MAX primitives */
static inline int gpt_w8_is_inner(const gpt_w8_builder *b, const gpt_w8_cand *c)
{
    return c->node >= 0 ? !b->nodes[c->node].is_leaf : c->count > GPT_WIDE8_LEAF_MAX;
}
static inline int32_t gpt_w8_leaf_count(const gpt_w8_builder *b, const gpt_w8_cand *c)
{
    return gpt_w8_is_inner(b, c) ? 0 : c->count;
}
static inline void gpt_w8_open(const gpt_w8_builder *b, const gpt_w8_cand *c, gpt_w8_cand *l, gpt_w8_cand *r)
{
    if (c->node >= 0) {
        *l = gpt_w8_cand_node(b, c->node + 1);
        *r = gpt_w8_cand_node(b, b->nodes[c->node].second_child_offset);
    } else {
        const int32_t h = (c->count + 1) / 2;
        *l = gpt_w8_cand_range(b, c->first, h);
        *r = gpt_w8_cand_range(b, c->first + h, c->count - h);
    }
}
static inline float gpt_w8_area(const gpt_w8_cand *c)
{
    const float dx = c->hi[0] - c->lo[0], dy = c->hi[1] - c->lo[1], dz = c->hi[2] - c->lo[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

/* slots: greedy - of all (child, free slot) pairs the one with the largest projection of the child's centroid (relative to the
 * centre of the children's union) on the slot's diagonal (+-1, +-1, +-1) is fixed first; ties: the lower child, then the lower slot */
static inline void gpt_wide8_assign_slots(const gpt_w8_cand *c, int n, int slot_of[8])
{
    double centre[3] = {0, 0, 0}, lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = __builtin_inf(); hi[a] = -__builtin_inf(); }
    for (int k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) {
            if (c[k].lo[a] < lo[a]) lo[a] = c[k].lo[a];
            if (c[k].hi[a] > hi[a]) hi[a] = c[k].hi[a];
        }
    for (int a = 0; a < 3; ++a) centre[a] = 0.5 * (lo[a] + hi[a]);
    double cost[8][8];
    for (int k = 0; k < n; ++k)
        for (int s = 0; s < 8; ++s) {
            double v = 0;
            for (int a = 0; a < 3; ++a) {
                const double x = 0.5 * ((double)c[k].lo[a] + (double)c[k].hi[a]) - centre[a];
                v += ((s >> a) & 1) ? x : -x;
            }
            cost[k][s] = (v == v) ? v : 0.0;
        }
    int child_done[8] = {0}, slot_used[8] = {0};
    for (int round = 0; round < n; ++round) {
        int bk = -1, bs = -1;
        for (int k = 0; k < n; ++k) {
            if (child_done[k]) continue;
            for (int s = 0; s < 8; ++s) {
                if (slot_used[s]) continue;
                if (bk < 0 || cost[k][s] > cost[bk][bs]) { bk = k; bs = s; }
            }
        }
        child_done[bk] = 1; slot_used[bs] = 1; slot_of[bk] = bs;
    }
}

static inline void gpt_w8_fill(gpt_w8_builder *b, const gpt_w8_cand *self, int32_t w, int depth);

/* node record w <- the wide node of candidate `self` (an inner candidate) */
static inline void gpt_w8_fill(gpt_w8_builder *b, const gpt_w8_cand *self, int32_t w, int depth)
{
    if (depth > b->max_depth) b->max_depth = depth;
    gpt_w8_cand cand[8];
    int n = 0;
    if (gpt_w8_is_inner(b, self)) {
        gpt_w8_open(b, self, &cand[0], &cand[1]);
        n = 2;
    } else {
        cand[n++] = *self;                               /* a tree that is one leaf */
    }
    while (n < 8) {
        int32_t tris = 0;
        for (int k = 0; k < n; ++k) tris += gpt_w8_leaf_count(b, &cand[k]);
        int pick = -1;
        float best = -1.f;
        gpt_w8_cand l, r;
        for (int k = 0; k < n; ++k) {
            if (!gpt_w8_is_inner(b, &cand[k])) continue;
            gpt_w8_open(b, &cand[k], &l, &r);
            if (tris + gpt_w8_leaf_count(b, &l) + gpt_w8_leaf_count(b, &r) > GPT_WIDE8_NODE_TRIS) continue;
            const float a = gpt_w8_area(&cand[k]);
            if (pick < 0 || a > best) { pick = k; best = a; }     /* ties, NaN: the earlier candidate */
        }
        if (pick < 0) break;
        gpt_w8_open(b, &cand[pick], &l, &r);
        for (int k = n; k > pick + 1; --k) cand[k] = cand[k - 1];
        cand[pick] = l;
        cand[pick + 1] = r;
        ++n;
    }
    int slot_of[8];
    gpt_wide8_assign_slots(cand, n, slot_of);
    int child_in_slot[8];
    for (int s = 0; s < 8; ++s) child_in_slot[s] = -1;
    for (int k = 0; k < n; ++k) child_in_slot[slot_of[k]] = k;

    gpt_wide8_node node;
    memset(&node, 0, sizeof(node));
    /* the grid: origin = lo corner of the union, per axis the smallest step that covers the extent with 255 steps */
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) { lo[a] = __builtin_inf(); hi[a] = -__builtin_inf(); }
    for (int k = 0; k < n; ++k)
        for (int a = 0; a < 3; ++a) {
            if (cand[k].lo[a] < lo[a]) lo[a] = cand[k].lo[a];
            if (cand[k].hi[a] > hi[a]) hi[a] = cand[k].hi[a];
        }
    double step[3];
    for (int a = 0; a < 3; ++a) {
        node.p[a] = (float)lo[a];
        int e = 1;
        const double ext = hi[a] - lo[a];
        if (ext > 0 && ext < __builtin_inf()) {
            int x;
            (void)frexp(ext / 255.0, &x);                  /* ext / 255 = m 2^x, 0.5 <= m < 1: 2^x covers it */
            e = x + 127;
            if (e < 1) e = 1;
            if (e > 254) e = 254;
            while (e < 254 && lo[a] + 255.0 * ldexp(1.0, e - 127) < hi[a]) ++e;
        }
        node.e[a] = (uint8_t)e;
        step[a] = ldexp(1.0, e - 127);
    }
    for (int s = 0; s < 8; ++s) {
        const int k = child_in_slot[s];
        for (int a = 0; a < 3; ++a) {
            if (k < 0) { node.qlo[a][s] = 255; node.qhi[a][s] = 0; continue; }
            double ql = floor(((double)cand[k].lo[a] - lo[a]) / step[a]), qh = ceil(((double)cand[k].hi[a] - lo[a]) / step[a]);
            if (!(ql >= 0)) ql = 0;
            if (!(qh >= 0)) qh = 0;
            if (ql > 255) ql = 255;
            if (qh > 255) qh = 255;
            while (ql > 0 && lo[a] + ql * step[a] > (double)cand[k].lo[a]) ql -= 1;       /* exact: never inside the child's box */
            while (qh < 255 && lo[a] + qh * step[a] < (double)cand[k].hi[a]) qh += 1;
            node.qlo[a][s] = (uint8_t)ql;
            node.qhi[a][s] = (uint8_t)qh;
        }
    }
    /* inner children: consecutive records; leaf children: consecutive triangles, slot by slot */
    int n_inner = 0;
    for (int s = 0; s < 8; ++s) {
        const int k = child_in_slot[s];
        if (k >= 0 && gpt_w8_is_inner(b, &cand[k])) { node.imask |= (uint8_t)(1u << s); ++n_inner; }
    }
    node.child_base = (uint32_t)b->n_out;
    if (b->n_out + n_inner > b->cap) { b->overflow = 1; n_inner = 0; node.imask = 0; }
    b->n_out += n_inner;
    node.tri_base = (uint32_t)b->n_tris;
    for (int s = 0; s < 8; ++s) {
        const int k = child_in_slot[s];
        if (k < 0 || ((node.imask >> s) & 1)) continue;
        if (gpt_w8_is_inner(b, &cand[k])) continue;            /* (dropped by an overflow) */
        for (int32_t i = 0; i < cand[k].count; ++i) {
            node.valid |= 1u << (4 * s + i);
            b->tri_order[b->n_tris++] = cand[k].first + i;
        }
    }
    b->out[w] = node;
    int at = 0;
    for (int s = 0; s < 8; ++s)
        if ((node.imask >> s) & 1) {
            gpt_w8_fill(b, &cand[child_in_slot[s]], (int32_t)node.child_base + at, depth + 1);
            ++at;
        }
}

/* capacity that always suffices */
static inline int32_t gpt_wide8_capacity(int32_t n_nodes, int32_t n_prims) { return n_nodes + n_prims / 2 + 8; }

/* Builds the 8-wide tree of a reference BVH.  tri_order holds n_prims entries (every primitive of a leaf exactly once).  Returns
 * the number of nodes (0 for an empty tree, -1 if `cap` was too small); *depth_out = number of levels (= deepest group stack + 1). */
static inline int32_t gpt_wide8_build(const gpt_bvh_node *nodes, int32_t n_nodes, const gpt_primitive *prims, gpt_wide8_node *out,
                                      int32_t cap, int32_t *tri_order, int32_t *n_tris_out, int32_t *depth_out)
{
    gpt_w8_builder b;
    b.nodes = nodes; b.prims = prims; b.out = out; b.tri_order = tri_order; b.n_out = 0; b.cap = cap; b.n_tris = 0; b.max_depth = 0; b.overflow = 0;
    if (depth_out) *depth_out = 0;
    if (n_tris_out) *n_tris_out = 0;
    if (n_nodes <= 0 || cap < 1) return n_nodes <= 0 ? 0 : -1;
    const gpt_w8_cand root = gpt_w8_cand_node(&b, 0);
    b.n_out = 1;
    gpt_w8_fill(&b, &root, 0, 1);
    if (b.overflow) return -1;
    if (depth_out) *depth_out = b.max_depth;
    if (n_tris_out) *n_tris_out = b.n_tris;
    return b.n_out;
}

#endif /* GPT_WIDE8_BVH_H */
