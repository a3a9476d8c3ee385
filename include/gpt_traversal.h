/*
 * gpt_traversal.h - traversal orders of the reference's BVH (shared by the host side of libgpt and the oracle).
 *
 * GPT_TRAVERSAL_REFERENCE   the reference's order: left child first, always (src/pathtracer.cu:221-252 pushes the
 *                           right child, then the left one).  Results are the reference's bit for bit.  Default.
 * GPT_TRAVERSAL_NEAR_FIRST  SURVEY.md 8(f) rank 3 "stage B": the same tree, but at every inner node the child that
 *                           lies first along the ray is visited first, so the closest hit is found earlier and more
 *                           of the tree is culled.  The tree, the box test and the triangle test are unchanged; what
 *                           can change is which of two EXACTLY equal hits wins and which boxes a shrinking tmax
 *                           culls.  CPU oracle and GPU kernel implement the same order and stay bit-identical to
 *                           each other; agreement with the reference order is measured statistically
 *                           (tests: relative RMS, bar 1e-4; in practice the films are identical).
 *
 * The reference's LinearBVHNode does not record the split axis, so the order is derived from the child boxes: the
 * axis along which the two children's box centres differ most, and which child is the lower one along it.
 */
#ifndef GPT_TRAVERSAL_H
#define GPT_TRAVERSAL_H

#include "gpt_types.h"

#define GPT_TRAVERSAL_REFERENCE  0
#define GPT_TRAVERSAL_NEAR_FIRST 1

/* For inner node `i`: bits 0-1 = axis, bit 2 = 1 when the LEFT child (i + 1) is the higher one along that axis.
 * With `neg` = sign bit of the ray direction on that axis, the right child is visited first iff neg ^ bit2. */
static inline int gpt_node_order_code(const gpt_bvh_node *nodes, int i)
{
    const gpt_bvh_node *l = &nodes[i + 1];
    const gpt_bvh_node *r = &nodes[nodes[i].second_child_offset];
    const float lc[3] = {l->fmin.x + l->fmax.x, l->fmin.y + l->fmax.y, l->fmin.z + l->fmax.z};
    const float rc[3] = {r->fmin.x + r->fmax.x, r->fmin.y + r->fmax.y, r->fmin.z + r->fmax.z};
    int axis = 0;
    float best = -1.f;
    for (int a = 0; a < 3; ++a) {
        float d = lc[a] - rc[a];
        if (d < 0.f) d = -d;
        if (d > best) { best = d; axis = a; }          /* NaN never wins; ties keep the lower axis */
    }
    return axis | ((lc[axis] > rc[axis]) ? 4 : 0);
}

/* octant of a direction: bit a = sign bit of component a (so -0.0f counts as negative, on every target) */
static inline int gpt_direction_octant(float dx, float dy, float dz)
{
    uint32_t x, y, z;
    __builtin_memcpy(&x, &dx, 4);
    __builtin_memcpy(&y, &dy, 4);
    __builtin_memcpy(&z, &dz, 4);
    return (int)((x >> 31) | ((y >> 31) << 1) | ((z >> 31) << 2));
}

/* does a ray of octant `oct` visit the right child of a node with order code `code` first? */
static inline int gpt_right_child_first(int code, int oct)
{
    return ((oct >> (code & 3)) & 1) ^ ((code >> 2) & 1);
}

#endif
