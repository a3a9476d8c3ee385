/*
 * gpt_traversal.h - traversal orders of the reference's BVH (shared by the host side of libgpt and the oracle).
 *
 * GPT_TRAVERSAL_REFERENCE   the reference's order: left child first, always (src/pathtracer.cu:221-252 pushes the
 *                           right child, then the left one).  Results are the reference's bit for bit.
 * GPT_TRAVERSAL_WIDE4 (2)   include/gpt_wide_bvh.h: the same boxes and triangle tests on a 4-wide tree collapsed from the
 *                           reference's, children entered nearest first.
 *
 * (Value 1 was a nearer-child-first order on the binary tree, rounds 1 - 3.  It was dominated by the 4-wide walk on every
 * configuration - profiles/r03/v6_configs.log - and is gone with its nine threaded node arrays; the value stays unused.)
 */
#ifndef GPT_TRAVERSAL_H
#define GPT_TRAVERSAL_H

#include "gpt_types.h"

#define GPT_TRAVERSAL_REFERENCE  0
#define GPT_TRAVERSAL_AUTO       (-1)     /* gpt_set_traversal_order / oracle_set_traversal: back to the rule below */

/* The default order, the same rule in gpt_begin and in the oracle: a scene whose device records fit the 12 KB of LDS the kernels
 * set aside for it (2 float4 per node, 8 per primitive - triangle + shading record -, 6 per light, 18 floats per material) is walked
 * in the reference's order from LDS; every other scene on the 4-wide tree (when it has one: gpt_wide_build succeeds and the tree is
 * at most 85 wide levels deep), where that is 9 - 74 % faster. */
#define GPT_LDS_SCENE_FLOAT4 768
static inline int gpt_scene_fits_lds(int n_nodes, int n_prims, int n_lights, int n_materials)
{
    return 2 * (int64_t)n_nodes + 8 * (int64_t)n_prims + 6 * (int64_t)n_lights + (18 * (int64_t)n_materials + 3) / 4 <= GPT_LDS_SCENE_FLOAT4;
}

#endif
