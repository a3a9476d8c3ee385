// pt_wavefront.hip — the decoupled scheduler for scenes in global memory (pt_wavefront.h): the reference's Path / Ao /
// Volpath loop bodies (src/pathtracer.cu:880-1021, 830-876, 1025-1242 for homogeneous media) cut at their Intersect /
// IntersectP calls (:905, :942, :960) into a SHADE stage over path slots and a TRACE stage over rays, alternating as kernel
// launches.  What a lane computes is what the persistent per-wave kernel of pt_kernel.hip computes for the same sample (the
// same functions of pt_shade.h in the same order, the same three rays per bounce, the same exact ray culling), so a sample is
// the same float4 whichever scheduler produced it, and the sample planes make the film independent of the order samples
// finish in.  What changes is who a lane works for:
//   * shade: lane i owns path slot i for ONE round; every slot whose rays are all back takes part, nobody waits for a
//     neighbour's rays (the per-wave kernel shades 31 of 64 lanes per round on the config-5 stand-in);
//   * trace: a lane that finishes a ray takes the next ray of the device-wide queue, whatever path it belongs to; the
//     traversal loop no longer shares its registers with the BSDFs (no scratch) nor its instruction cache with them.
// Float contract and draw order as everywhere: no contraction, IEEE divide / sqrt, soft-math transcendentals.

#include "pt_shade.h"
#include "pt_wavefront.h"
#include "../../include/gpt_wide_bvh.h"

namespace pt {

__device__ __forceinline__ float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
__device__ __forceinline__ float4 f4u(V3 v, uint32_t w) { return make_float4(v.x, v.y, v.z, __uint_as_float(w)); }
__device__ __forceinline__ V3 xyz(float4 v) { return V3{v.x, v.y, v.z}; }

// ------------------------------------------------------------------------------------------------ shade stage ------
// INTEG: GPT_IT_PT, GPT_IT_AO, GPT_IT_VPT (homogeneous media, no material-less surfaces: the three-ray form)
// One wave shades the 64 path slots of one chunk (slot = 64 chunk + lane).  Returns the number of rays the chunk put into its segment
// of the queue (wave-uniform) and writes it to *seg_count_lds.
template <int INTEG>
__device__ __forceinline__ int wf_shade_chunk(const DevParams &P, const WfParams &W, uint32_t chunk, unsigned lane, uint32_t *seg_count_lds)
{
    const uint32_t i = chunk * 64u + lane;                         // path slot
    const uint32_t np = W.n_paths;

    // everything the slot may need is requested at once, whatever its flags turn out to be: one memory round trip instead of two in a
    // phase that is bound by them (the BSDF-sampled light ray's planes aside: few paths have one)
    const float4 a3 = W.s3[i];
    const float4 a0 = W.s0[i], a1 = W.s1[i], a2 = W.s2[i], a4 = W.s4[i], ao = W.org[i];
    const float4 ld_rp = W.ray[i], ld_hp = W.hit[i];
    const float ld_hs = W.hit[2u * np + i].x;
    uint32_t flags = __float_as_uint(a3.w);
    bool alive = (flags & kWfAlive) != 0u;

    // ---- per-path state (pt_kernel.hip keeps the same in registers) ----
    Rng rng;
    rng.x = 1;
    V3 Li = v3(0.f), beta = v3(1.f);
    V3 beta_ld = v3(0.f), cand = v3(0.f), mis_fr = v3(0.f);
    float mis_cos = 0.f, mis_pdf = 1.f;
    uint32_t dst = 0;
    int medium = -1, medium_ld = -1;
    V3 org = v3(0.f), dir_p = v3(0.f), dir_m = v3(0.f), dir_s = v3(0.f);
    float tmax_s = 0.f;
    bool specular = (flags & kWfSpecular) != 0u, direct = (flags & kWfDirect) != 0u, ending = (flags & kWfEnding) != 0u;
    bool has_p = alive && (flags & kWfHasP) != 0u, has_m = alive && (flags & kWfHasM) != 0u, has_s = alive && (flags & kWfHasS) != 0u;
    bool mis_any = (flags & kWfMisAny) != 0u, poison_occluded = (flags & kWfPoison) != 0u;
    int bounces = (int)(flags & kWfBouncesMask);
    RayResults res;
    res.occluded = false;
    res.prim_m = res.prim_p = -1;
    res.t_m = res.b1_m = res.b2_m = res.t_p = res.b1_p = res.b2_p = 0.f;

    bool finish = false, waiting = false;
    if (alive) {
        Li = xyz(a0);
        beta = V3{a0.w, a1.x, a1.y};
        mis_cos = a1.z;
        mis_pdf = a1.w;
        cand = xyz(a2);
        rng.x = __float_as_uint(a2.w);
        beta_ld = xyz(a3);
        mis_fr = xyz(a4);
        dst = __float_as_uint(a4.w);
        org = xyz(ao);
        if (INTEG == GPT_IT_VPT) {
            const uint32_t mm = __float_as_uint(ao.w);
            medium = (int)(int16_t)(mm & 0xffffu);
            medium_ld = (int)(int16_t)(mm >> 16);
        }
        if (has_p) {
            const float4 r = ld_rp, h = ld_hp;
            dir_p = xyz(r);
            res.prim_p = __float_as_int(h.x); res.t_p = h.y; res.b1_p = h.z; res.b2_p = h.w;
            waiting = waiting || res.prim_p == kWfPending;
        }
        if (has_m) {
            const float4 r = W.ray[np + i], h = W.hit[np + i];
            dir_m = xyz(r);
            res.prim_m = __float_as_int(h.x); res.t_m = h.y; res.b1_m = h.z; res.b2_m = h.w;
            waiting = waiting || res.prim_m == kWfPending;
        }
        if (has_s) {
            const int prim_s = __float_as_int(ld_hs);
            res.occluded = prim_s >= 0;
            waiting = waiting || prim_s == kWfPending;
        }
      // a path one of whose rays the trace stage has parked sits this round out: nothing of it changes, its results stay where they are
      if (!waiting) {

        // ---- resolve the direct light of the previous bounce (pathtracer.cu:943-994) ------------------
        if (direct) {
            V3 Ld = v3(0.f, 0.f, 0.f);
            if (has_s && !res.occluded) Ld += cand;
            if (INTEG == GPT_IT_VPT && has_s && res.occluded && poison_occluded)
                Ld += v3(__builtin_nanf(""));           // Tr = 0 times a non-finite factor (pathtracer.cu:298-322)
            V3 tr_m = v3(1.f, 1.f, 1.f);                 // Volpath: transmittance along the BSDF-sampled light ray
            if (INTEG == GPT_IT_VPT && has_m && medium_ld >= 0)
                tr_m = hom_tr(P.mediums[medium_ld], res.prim_m >= 0 ? res.t_m : __builtin_inff());
            if (has_m) {
                if (res.prim_m >= 0) {
                    if (!mis_any) {
                        V3 n;
                        int lightIdx;
                        make_light_hit(P, res.prim_m, res.b1_m, res.b2_m, n, lightIdx);
                        V3 radiance = v3(0.f, 0.f, 0.f);
                        if (lightIdx != -1) radiance = area_le(P.lights[lightIdx], n, -dir_m);
                        if (!is_black(radiance)) {
                            V3 p = org + res.t_m * dir_m;
                            float pdfA = 1.f / P.lights[lightIdx].area;              // area.h:28-32
                            float choicePdf = pdf_from_light_distribution(P, lightIdx);
                            float lenSquare = dot(p - org, p - org);
                            float costheta = fabs_(dot(n, dir_m));
                            float lPdf = pdfA * lenSquare / (costheta);
                            float weight = power_heuristic(1, mis_pdf, 1, lPdf * choicePdf);
                            if (INTEG == GPT_IT_VPT) Ld += weight * tr_m * mis_fr * radiance * mis_cos / mis_pdf;
                            else Ld += weight * mis_fr * radiance * mis_cos / mis_pdf;
                        }
                    }
                } else if (P.inf.isvalid) {
                    V3 radiance = inf_le(P.inf, dir_m);
                    float choicePdf = pdf_from_light_distribution(P, P.n_lights);
                    float lightPdf = ONE_OVER_FOUR_PI;                             // infinite.h:38-41
                    float weight = power_heuristic(1, mis_pdf, 1, lightPdf * choicePdf);
                    if (INTEG == GPT_IT_VPT) Ld += weight * tr_m * mis_fr * radiance * mis_cos / mis_pdf;
                    else Ld += weight * mis_fr * radiance * mis_cos / mis_pdf;
                }
            }
            // executed even when both rays were skipped: beta * 0 is NaN for a non-finite throughput, and the
            // reference then discards the sample (pathtracer.cu:994,1019)
            Li += beta_ld * Ld;
            direct = false;
        }
        if (ending) finish = true;

        // ---- the path ray came back: pathtracer.cu:905-1016 ----------------------
        if (!finish && has_p) {
            if (res.prim_p < 0) {
                if (INTEG != GPT_IT_AO && (bounces == 0 || specular) && P.inf.isvalid)
                    Li += beta * inf_le(P.inf, dir_p);
                finish = true;          // Ao: the sample is 0 (pathtracer.cu:852-855)
            } else {
                Ray r;
                r.o = org;
                r.d = dir_p;
                const Hit isect = make_hit(P, r, res.t_p, res.prim_p, res.b1_p, res.b2_p);
                const V3 pos = isect.pos;
                const V3 nor = isect.nor;
                const V2 uv = isect.uv;
                const V3 dpdu = isect.dpdu;
                const V3 wo = -dir_p;
                const gpt_material material = P.materials[isect.matIdx];
                has_s = has_m = has_p = false;

                // Volpath (pathtracer.cu:1062-1070): the medium decides whether the ray gets as far as the surface
                bool scattered = false;
                float scatter_t = 0.f;
                if (INTEG == GPT_IT_VPT && medium >= 0) {
                    float u = rng_uniform(rng);
                    beta *= hom_sample(P.mediums[medium], res.t_p, u, scatter_t, scattered);
                }
                if (INTEG == GPT_IT_VPT && is_black(beta)) {
                    finish = true;
                } else if (INTEG == GPT_IT_VPT && scattered) {
                    // ---- a scattering event inside the medium (pathtracer.cu:1071-1101) ----
                    const DevMedium M = P.mediums[medium];
                    float u = rng_uniform(rng);
                    float choicePdf;
                    int idx = lookup_light_distribution(P, u, choicePdf);
                    bool inf = idx == P.n_lights;
                    V3 samplePos = org + dir_p * scatter_t;
                    float u1x = rng_uniform(rng);
                    float u1y = rng_uniform(rng);
                    V3 radiance = v3(0.f), lightNor;
                    Ray shadowRay;
                    shadowRay.o = samplePos;
                    shadowRay.d = v3(0.f);
                    shadowRay.tmin = P.eps;
                    shadowRay.tmax = 0.f;
                    float lightPdf = 0.f;
                    if (idx >= 0) {
                        if (!inf)
                            area_sample_light(P.lights[idx], samplePos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                        else
                            inf_sample_light(P.inf, samplePos, v2(u1x, u1y), radiance, shadowRay, lightNor, lightPdf, P.eps);
                    }
                    float phase = medium_phase(M, wo, shadowRay.d);
                    poison_occluded = false;
                    if (!is_black(radiance)) {
                        // Li += tr * beta * phase * radiance / (lightPdf * choicePdf), tr = 0 when the light is hidden
                        const V3 tr1 = hom_tr(M, shadowRay.tmax);
                        cand = tr1 * beta * phase * radiance / (lightPdf * choicePdf);
                        poison_occluded = is_nan(v3(0.f) * beta * phase * radiance / (lightPdf * choicePdf));
                        if (!is_black(cand) || poison_occluded) {
                            dir_s = shadowRay.d;
                            tmax_s = shadowRay.tmax;
                            has_s = true;
                        }
                    }
                    beta_ld = v3(1.f, 1.f, 1.f);
                    direct = true;
                    float pux = rng_uniform(rng);
                    float puy = rng_uniform(rng);
                    const V3 dir = medium_sample_phase(M, pux, puy);
                    org = samplePos;
                    specular = false;
                    ending = true;
                    if (bounces + 1 < P.max_depth) {
                        bool kill = false;
                        if (bounces > 3) {
                            float illumate = clamp(1.f - luminance(beta), 0.f, 1.f);
                            if (rng_uniform(rng) < illumate)
                                kill = true;
                            else
                                beta /= (1 - illumate);
                        }
                        if (!kill) {
                            dir_p = dir;
                            has_p = true;
                            ending = false;
                            bounces++;
                        }
                    }
                    if (ending && !has_s) {
                        Li += beta_ld * v3(0.f, 0.f, 0.f);
                        direct = false;
                        finish = true;
                    }
                } else if (INTEG == GPT_IT_AO) {
                    // pathtracer.cu:857-872
                    V3 n = nor;
                    if (dot(wo, nor) < 0.f)
                        n = -n;
                    float u1 = rng_uniform(rng);
                    float u2 = rng_uniform(rng);
                    float pdf;
                    V3 dir = cosine_hemisphere(u1, u2, pdf);
                    V3 uu = dpdu, ww;
                    ww = cross(uu, n);
                    dir = to_world(dir, uu, n, ww);
                    float cosine = dot(dir, n);
                    float v = cosine * ONE_OVER_PI / pdf;
                    cand = v3(v, v, v);                 // L += v if the occlusion ray escapes
                    beta_ld = v3(1.f, 1.f, 1.f);
                    org = pos;
                    if (!is_black(cand)) {               // a zero term adds nothing either way (NaN is traced)
                        dir_s = dir;
                        tmax_s = P.ao_max_dist;
                        has_s = true;
                    }
                    direct = true;
                    ending = true;
                    if (!has_s) {
                        Li += beta_ld * v3(0.f, 0.f, 0.f);
                        direct = false;
                        finish = true;
                    }
                } else if ((bounces == 0 || specular) && isect.lightIdx != -1) {
                    if (INTEG == GPT_IT_VPT) {
                        V3 tr = v3(1.f, 1.f, 1.f);
                        if (medium >= 0) tr = hom_tr(P.mediums[medium], res.t_p);
                        Li += tr * beta * area_le(P.lights[isect.lightIdx], nor, wo);        // pathtracer.cu:1103-1115
                    } else {
                        Li += beta * area_le(P.lights[isect.lightIdx], nor, wo);
                    }
                    finish = true;
                } else {
                    org = pos;
                    // direct light with multiple importance sampling: everything that does not depend on visibility
                    // is evaluated now (pathtracer.cu:925-956); the draw order is the reference's
                    if (!is_delta(PT_MATERIAL_TYPE(material))) {
                        poison_occluded = false;
                        float u = rng_uniform(rng);
                        float choicePdf;
                        int idx = lookup_light_distribution(P, u, choicePdf);
                        bool inf = idx == P.n_lights;
                        float u1x = rng_uniform(rng);
                        float u1y = rng_uniform(rng);
                        V2 u1 = v2(u1x, u1y);
                        V3 radiance = v3(0.f), lightNor;
                        Ray shadowRay;
                        shadowRay.o = pos;
                        shadowRay.d = v3(0.f);
                        shadowRay.tmin = P.eps;
                        shadowRay.tmax = 0.f;
                        float lightPdf = 0.f;
                        if (idx >= 0) {
                            if (!inf)
                                area_sample_light(P.lights[idx], pos, u1, radiance, shadowRay, lightNor, lightPdf, P.eps);
                            else
                                inf_sample_light(P.inf, pos, u1, radiance, shadowRay, lightNor, lightPdf, P.eps);
                        }
                        if (!is_black(radiance)) {
                            V3 fr;
                            float samplePdf;
                            eval_bsdf(P, material, wo, shadowRay.d, nor, uv, dpdu, fr, samplePdf);
                            float weight = power_heuristic(1, lightPdf * choicePdf, 1, samplePdf);
                            if (INTEG == GPT_IT_VPT) {
                                // Ld += weight * tr * fr * radiance * |cos| / pdf with tr = Tr(shadowRay): the medium's
                                // transmittance if the light is visible, 0 if not (pathtracer.cu:1146-1151)
                                V3 tr1 = v3(1.f, 1.f, 1.f);
                                if (medium >= 0) tr1 = hom_tr(P.mediums[medium], shadowRay.tmax);
                                cand = weight * tr1 * fr * radiance * fabs_(dot(nor, shadowRay.d)) / (lightPdf * choicePdf);
                                poison_occluded = is_nan(weight * v3(0.f) * fr * radiance * fabs_(dot(nor, shadowRay.d)) / (lightPdf * choicePdf));
                            } else {
                                cand = weight * fr * radiance * fabs_(dot(nor, shadowRay.d)) / (lightPdf * choicePdf);
                            }
                            // an exactly-zero term (e.g. the light is below the horizon of a lambertian surface: Fr
                            // returns 0) adds nothing whether or not the light is visible
                            if (!is_black(cand) || (INTEG == GPT_IT_VPT && poison_occluded)) {
                                dir_s = shadowRay.d;
                                tmax_s = shadowRay.tmax;
                                has_s = true;
                            }
                        }
                        float usx = rng_uniform(rng);
                        float usy = rng_uniform(rng);
                        float usz = rng_uniform(rng);
                        V3 out, fr;
                        float pdf;
                        sample_bsdf(P, material, wo, nor, uv, dpdu, v3(usx, usy, usz), out, fr, pdf);
                        if (!(is_black(fr) || pdf == 0)) {
                            // The BSDF-sampled light ray contributes only if its CLOSEST hit is an emitter triangle
                            // (pathtracer.cu:964-976) or, with an environment light, if it escapes (:978-990).  With few
                            // emitters, test their triangles first: if Triangle::Intersect would reject all of them, no
                            // traversal order can make an emitter the closest hit.  Then, without an environment light the
                            // ray is not traced at all; with one, only hit / no hit matters and the ray is traced as an
                            // any-hit ray.
                            bool useful = true;
                            mis_any = false;
                            if (P.n_lights <= kEmitterPretestMax) {
                                bool emitter = false;
                                for (int li = 0; li < P.n_lights; ++li)
                                    emitter = emitter || emitter_accepts(P.lights[li], pos, out, P.eps);
                                if (!emitter) {
                                    useful = P.inf.isvalid != 0;
                                    mis_any = true;
                                }
                            }
                            if (useful) {
                                mis_fr = fr;
                                mis_cos = fabs_(dot(out, nor));
                                mis_pdf = pdf;
                                dir_m = out;
                                has_m = true;
                            }
                        }
                        beta_ld = beta;
                        medium_ld = medium;
                        direct = true;
                    }
                    // continuation.  The reference also samples it on the last bounce and then leaves the loop;
                    // nothing of that sample reaches Li, so it is skipped.
                    ending = true;
                    if (bounces + 1 < P.max_depth) {
                        float ux = rng_uniform(rng);
                        float uy = rng_uniform(rng);
                        float uz = rng_uniform(rng);
                        V3 out, fr;
                        float pdf;
                        sample_bsdf(P, material, wo, nor, uv, dpdu, v3(ux, uy, uz), out, fr, pdf);
                        if (!is_black(fr)) {
                            beta *= fr * fabs_(dot(nor, out)) / pdf;
                            specular = is_delta(PT_MATERIAL_TYPE(material));
                            if (INTEG == GPT_IT_VPT) {
                                // the medium on the side the new ray leaves on; a reflection stays where it was
                                // (pathtracer.cu:1223-1227)
                                const int m_in = P.prim_media[2 * res.prim_p], m_out = P.prim_media[2 * res.prim_p + 1];
                                int m2 = dot(out, nor) > 0 ? m_out : m_in;
                                m2 = dot(wo, nor) * dot(out, nor) > 0 ? medium : m2;
                                medium = m2;
                            }
                            bool kill = false;
                            if (bounces > 3) {
                                float illumate = clamp(1.f - luminance(beta), 0.f, 1.f);
                                if (rng_uniform(rng) < illumate)
                                    kill = true;
                                else
                                    beta /= (1 - illumate);
                            }
                            if (!kill) {
                                dir_p = out;
                                has_p = true;
                                ending = false;
                                bounces++;
                            }
                        }
                    }
                    if (ending && !has_s && !has_m) {
                        if (direct) {      // nothing to wait for: Ld = 0
                            Li += beta_ld * v3(0.f, 0.f, 0.f);
                            direct = false;
                        }
                        finish = true;
                    }
                }
            }
        }


      }
    }
    if (finish) {
        // The sample goes to its iteration's plane as is; the finite-guard of pathtracer.cu:1019-1020 and the accumulation
        // run in iteration order in pt_output_kernel.
        reinterpret_cast<float4 *>(P.samples)[dst] = make_float4(Li.x, Li.y, Li.z, 0.f);
        alive = false;
        has_s = has_m = has_p = false;
    }

    // ---- regenerate: pathtracer.cu:881-903.  Samples are handed out as in the per-wave kernel: a WORK ITEM is the samples of one
    // 8x8 tile for a chunk of iterations (sample k -> pixel k % 64, iteration chunk_first + k / 64); the 64 slots of a wave take
    // consecutive samples of the wave's item, and a wave whose item is used up claims the next one with ONE atomic - so lanes
    // that start together start on one tile, and atomics are per item, not per sample.  The wave's item survives between rounds
    // in wave_item[wave].
    const bool was_alive = (flags & kWfAlive) != 0u;
    {
        const uint32_t wave = i >> 6;
        unsigned long long m_need = ballot(!alive);
        if (m_need != 0ull) {
            const uint2 wi = W.wave_item[wave];
            uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int)wi.x), taken = (uint32_t)__builtin_amdgcn_readfirstlane((int)wi.y);
            const uint32_t item0 = item, taken0 = taken;
            const uint32_t n_owned = (uint32_t)(P.plane >> 6);
            bool need = !alive;
            for (;;) {
                uint32_t chunk = 0, tile_seq = 0, item_samples = 0;
                if (item != 0xffffffffu) {
                    chunk = item / n_owned;
                    tile_seq = item - chunk * n_owned;
                    const uint32_t its = (chunk + 1u == W.n_chunks) ? P.iter_count - chunk * W.item_iters : W.item_iters;
                    item_samples = 64u * its;
                }
                if (item == 0xffffffffu || taken >= item_samples) {
                    // the next item (a stale read of the counter can only be too small: then the atomic tells)
                    uint32_t t = 0xffffffffu;
                    if (lane == 0) {
                        const uint32_t seen = __hip_atomic_load(&W.ctrl->next_item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (seen < W.n_items) t = atomicAdd(&W.ctrl->next_item, 1u);
                    }
                    t = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
                    if (t >= W.n_items) {
                        item = 0xffffffffu;
                        taken = 0xffffffffu;
                        break;
                    }
                    item = t;
                    taken = 0u;
                    continue;
                }
                const uint32_t k = taken + (uint32_t)lane_rank(m_need);
                taken += (uint32_t)popc(m_need);
                if (need && k < item_samples) {
                    need = false;
                    uint32_t tile_local = tile_seq;
                    {   // a bijection of [0, n_owned): the owned tiles seen as a grid of gw columns, strip by strip (pt_kernel.hip)
                        constexpr uint32_t TILE_STRIP = 16u;
                        const uint32_t gw = (P.tiles_x + P.n_ranks - 1u) / P.n_ranks, gh = n_owned / gw;
                        if (tile_local < gw * gh) {
                            const uint32_t strip_items = TILE_STRIP * gh;
                            const uint32_t strip = tile_local / strip_items, rr = tile_local - strip * strip_items;
                            const uint32_t left = gw - strip * TILE_STRIP;
                            const uint32_t w = left < TILE_STRIP ? left : TILE_STRIP;
                            const uint32_t ty = rr / w;
                            tile_local = ty * gw + strip * TILE_STRIP + (rr - ty * w);
                        }
                    }
                    const uint32_t tile = P.rank + tile_local * P.n_ranks;
                    const uint32_t pix = k & 63u, iter_rel = chunk * W.item_iters + (k >> 6);
                    const uint32_t x = (tile % P.tiles_x) * 8u + (pix & 7u), y = (tile / P.tiles_x) * 8u + (pix >> 3);
                    if (x < P.stride && y < P.rows) {
                        const uint32_t iter = P.iter_first + iter_rel;
                        dst = iter_rel * (uint32_t)P.plane + tile_local * 64u + pix;
                        const uint32_t pixel = x + y * P.stride;          // pathtracer.cu:881-883
                        rng_seed(rng, wang_hash(pixel) + wang_hash(iter));
                        float offsetx = rng_uniform(rng) - 0.5f;
                        float offsety = rng_uniform(rng) - 0.5f;
                        float du1 = rng_uniform(rng);
                        float du2 = rng_uniform(rng);
                        Ray r0 = primary_ray(P.cam, x + offsetx, y + offsety, du1, du2);
                        org = r0.o;
                        dir_p = r0.d;
                        has_p = true;
                        has_s = has_m = false;
                        Li = v3(0.f, 0.f, 0.f);
                        beta = v3(1.f, 1.f, 1.f);
                        specular = false;
                        bounces = 0;
                        ending = false;
                        direct = false;
                        mis_any = false;
                        poison_occluded = false;
                        medium = INTEG == GPT_IT_VPT ? P.cam.medium : -1;      // pathtracer.cu:1043
                        medium_ld = -1;
                        alive = true;
                    }
                    // (a pixel outside the frame: the sample is not rendered, the slot waits for the next round)
                }
                m_need = ballot(need);                               // lanes the item had no sample for: they take from the next item
                if (m_need == 0ull) break;
            }
            if (lane == 0 && (item != item0 || taken != taken0)) W.wave_item[wave] = make_uint2(item, taken);
        }
    }

    // ---- this round's rays go to the chunk's segment of the queue: path rays first, then light rays, then shadow rays ----
    int n_emitted = 0;
    {
        const bool emit_p = has_p && !waiting, emit_m = has_m && !waiting, emit_s = has_s && !waiting;      // (a waiting path's rays are in flight already)
        const unsigned long long m_p = ballot(emit_p), m_m = ballot(emit_m), m_s = ballot(emit_s);
        const int n_p = popc(m_p), n_m = popc(m_m), n_all = n_p + n_m + popc(m_s);
        const uint32_t wave = i >> 6;
        uint32_t *seg = W.rayq + (size_t)wave * (uint32_t)kWfSegRays;
        if (lane == 0) *seg_count_lds = (uint32_t)n_all;
        n_emitted = n_all;
        if (emit_p) {
            seg[lane_rank(m_p)] = i;
            W.ray[i] = f4(dir_p, __builtin_inff());
        }
        if (emit_m) {
            seg[n_p + lane_rank(m_m)] = i | (1u << kWfKindShift) | (mis_any ? kWfAnyHit : 0u);
            W.ray[np + i] = f4(dir_m, __builtin_inff());
        }
        if (emit_s) {
            seg[n_p + n_m + lane_rank(m_s)] = i | (2u << kWfKindShift) | kWfAnyHit;
            W.ray[2u * np + i] = f4(dir_s, tmax_s);
        }
    }

    // ---- the slot's state for the next round ----
    if (alive && !waiting) {
        flags = (uint32_t)bounces | (specular ? kWfSpecular : 0u) | (direct ? kWfDirect : 0u) | (ending ? kWfEnding : 0u) | kWfAlive |
                (has_p ? kWfHasP : 0u) | (has_m ? kWfHasM : 0u) | (has_s ? kWfHasS : 0u) | (mis_any ? kWfMisAny : 0u) |
                (poison_occluded ? kWfPoison : 0u);
        W.s0[i] = make_float4(Li.x, Li.y, Li.z, beta.x);
        W.s1[i] = make_float4(beta.y, beta.z, mis_cos, mis_pdf);
        W.s2[i] = f4u(cand, rng.x);
        W.s3[i] = f4u(beta_ld, flags);
        W.s4[i] = f4u(mis_fr, dst);
        W.org[i] = f4u(org, INTEG == GPT_IT_VPT ? (((uint32_t)medium & 0xffffu) | ((uint32_t)medium_ld << 16)) : 0u);
    } else if (was_alive && !alive) {
        W.s3[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return n_emitted;
}

// ------------------------------------------------------------------------------------------------ trace phase ------
// The waves of a workgroup take the segments the workgroup's shade phase filled, one segment at a time (an LDS atomic); a lane that
// finishes a ray (its result goes straight to hit[kind][path]) takes the next id of the segment, reads the ray's direction and
// origin from the path's planes and starts at the root.  The walks are those of pt_kernel.hip:
//   WIDE   the 4-wide tree, one lane per ray, per-lane stack in LDS (include/gpt_wide_bvh.h; trace_pool_wide<>)
//   !WIDE  the reference's order on the threaded binary tree (trace_pool<>)
// every box and triangle test in the same arithmetic (bbox.h:77-96, mesh.h:45-67).
#ifndef PT_WF_STACK_LEVELS
#define PT_WF_STACK_LEVELS 24
#endif
#ifndef PT_WF_FETCH_T
#define PT_WF_FETCH_T 8              // idle lanes that trigger a refill
#endif
#ifndef PT_WF_PROBE
#define PT_WF_PROBE 0
#endif
#ifndef PT_WF_WAVES
#define PT_WF_WAVES 4                // waves per SIMD the kernel is compiled for (launch bounds)
#endif
constexpr int kWfStackLevels = PT_WF_STACK_LEVELS;

__device__ __forceinline__ void wf_cex(unsigned &ka, unsigned &ea, unsigned &kb, unsigned &eb)
{
    const bool swap = kb < ka;
    const unsigned k0 = swap ? kb : ka, k1 = swap ? ka : kb, e0 = swap ? eb : ea, e1 = swap ? ea : eb;
    ka = k0; kb = k1; ea = e0; eb = e1;
}

// Shared by the trace walks: what a workgroup keeps in LDS for its rounds
struct WfShared {
    uint32_t seg_count[kWfWgChunks];    // rays in each chunk's segment of the queue, written by the shade phase
    uint32_t shade_next;                // shade phase: next chunk nobody has taken
    uint32_t trace_next;                // trace phase: next segment nobody has taken
    uint32_t emitted;                   // rays the shade phase put into the queue
    uint32_t parked;                    // the trace phase parked rays for the next round
    uint32_t items_left;                // the batch still has work items
};

template <bool WIDE>
__device__ __forceinline__ void wf_trace_cxx(const DevParams &P, const WfParams &W, WfShared &sh, uint32_t chunk0, uint32_t *ids, uint32_t *lds_stack_wave,
                                             unsigned lane)
{
    const uint32_t np = W.n_paths;
    const float tmin_ray = P.eps;
    const char *tris = reinterpret_cast<const char *>(P.tris);
    // the wave's current segment: its ids in ids[0 .. chunk_end); cursor = the next one to hand out
    uint32_t cursor = 0, chunk_end = 0;
    bool exhausted = false;

    // lane state: one ray
#if PT_WF_PROBE
    uint32_t ray_trips = 0;       // probe builds (C++ walk): trips this lane's ray has taken part in -> histogram by log2 in P.counters[0..15]
#endif
    uint32_t id = 0;
    bool has = false;
    V3 o = v3(0.f), d = v3(0.f), inv = v3(0.f);
    float tmax = 0.f;
    // wide walk
    unsigned *stk = lds_stack_wave + lane;
    uint32_t *spill = W.spill + (size_t)(blockIdx.x * (unsigned)kWfWgWaves + (threadIdx.x >> 6)) * 64u * W.spill_levels + lane;
    const char *wnodes = reinterpret_cast<const char *>(P.wide);
    unsigned cur = GPT_WIDE_NONE;
    int sp = 0;
    int bprim = -1;
    float bt = 0.f, bb1 = 0.f, bb2 = 0.f;
    // binary walk (threaded preorder; cursors are byte offsets)
    const char *nodes = reinterpret_cast<const char *>(P.nodes);
    const int end = 32 * P.n_nodes;
    int idx = end, tri = 0, tri_last = -1;

    for (;;) {
        bool fin;
        if (WIDE) fin = has && cur == GPT_WIDE_NONE;
        else fin = has && !(tri <= tri_last) && !(idx < end);
        const unsigned long long m_has = ballot(has), m_fin = ballot(fin);
        if (m_fin != 0ull) {
            if (fin) {
#if PT_WF_PROBE
                atomicAdd(&P.counters[ray_trips < 2u ? 0 : (31 - __builtin_clz(ray_trips) > 15 ? 15 : 31 - __builtin_clz(ray_trips))], 1ull);
#endif
                const uint32_t path = id & kWfPathMask, kind = (id >> kWfKindShift) & 3u;
                float4 r;
                if (WIDE) r = make_float4(__int_as_float(bprim), bprim < 0 ? tmax : bt, bb1, bb2);   // a miss reports the end of the interval
                else r = make_float4(__int_as_float(bprim < 0 ? -1 : (int)((unsigned)bprim / 48u)), tmax, bb1, bb2);
                W.hit[kind * np + path] = r;
                has = false;
            }
        }
        const unsigned long long m_busy = m_has & ~m_fin;
        const int n_idle = 64 - popc(m_busy);
        if (!exhausted && n_idle >= PT_WF_FETCH_T) {
            while (cursor >= chunk_end && !exhausted) {
                // ---- claim the workgroup's next segment (an LDS atomic)
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(&sh.trace_next, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= (uint32_t)kWfWgChunks) { exhausted = true; break; }
                const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.seg_count[c]);
                const uint32_t *seg = W.rayq + (size_t)(chunk0 + c) * (uint32_t)kWfSegRays;
                for (uint32_t j = lane; j < cnt; j += 64u) ids[j] = seg[j];
                cursor = 0;
                chunk_end = cnt;                                   // (an empty segment: claim the next one)
                wave_lds_fence();
            }
            if (!exhausted) {
                // ---- refill: idle lanes take the next rays of the chunk, in lane order
                const uint32_t nth = cursor + (uint32_t)lane_rank(~m_busy);
                if (!has && nth < chunk_end) {
                    id = ids[nth];
                    const uint32_t path = id & kWfPathMask, kind = (id >> kWfKindShift) & 3u;
                    const float4 r0 = W.ray[kind * np + path];
                    const float4 ro = W.org[path];
                    has = true;
#if PT_WF_PROBE
                    ray_trips = 0;
#endif
                    o = xyz(ro);
                    d = xyz(r0);
                    inv = V3{1.f / d.x, 1.f / d.y, 1.f / d.z};     // bbox.h:79 computes 1/d at every node visit: the same quotient
                    tmax = r0.w;
                    bprim = -1;
                    bt = bb1 = bb2 = 0.f;
                    if (WIDE) {
                        cur = 0u;
                        sp = 0;
                    } else {
                        idx = 0;
                        tri = 0;
                        tri_last = -1;
                    }
                }
                cursor += (uint32_t)n_idle;
                wave_lds_fence();
            }
            continue;
        }
        if (m_busy == 0ull) break;
        const bool any_hit = (id & kWfAnyHit) != 0u;
#if PT_WF_PROBE
        if (has) ray_trips++;
#endif

        if (WIDE) {
            bool leaf = has && (cur >> 31) != 0u;
            bool inner = has && (cur >> 31) == 0u;
            {
                const int n_leaf = popc(ballot(leaf)), n_inner = popc(ballot(inner));
                if (n_leaf < 8 && n_inner > 0) leaf = false;                      // the leaves wait until 8 have gathered
            }
            bool pop = false;
            if (inner) {
                // ---- a wide node: four boxes, bbox.h:77-96 each
                const float4 *npn = reinterpret_cast<const float4 *>(wnodes + (size_t)cur);
                const float4 lx = npn[0], ly = npn[1], lz = npn[2], hx = npn[3], hy = npn[4], hz = npn[5];
                const uint4 en = *reinterpret_cast<const uint4 *>(npn + 6);
                const float blx[4] = {lx.x, lx.y, lx.z, lx.w}, bly[4] = {ly.x, ly.y, ly.z, ly.w}, blz[4] = {lz.x, lz.y, lz.z, lz.w};
                const float bhx[4] = {hx.x, hx.y, hx.z, hx.w}, bhy[4] = {hy.x, hy.y, hy.z, hy.w}, bhz[4] = {hz.x, hz.y, hz.z, hz.w};
                unsigned e[4] = {en.x, en.y, en.z, en.w}, key[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float t1 = (blx[k] - o.x) * inv.x;
                    const float t2 = (bhx[k] - o.x) * inv.x;
                    const float t3 = (bly[k] - o.y) * inv.y;
                    const float t4 = (bhy[k] - o.y) * inv.y;
                    const float t5 = (blz[k] - o.z) * inv.z;
                    const float t6 = (bhz[k] - o.z) * inv.z;
                    const float tn = fmax_(fmax_(fmin_(t1, t2), fmin_(t3, t4)), fmin_(t5, t6));
                    const float tf = fmin_(fmin_(fmax_(t1, t2), fmax_(t3, t4)), fmax_(t5, t6));
                    const bool hit = e[k] != GPT_WIDE_NONE && !(tf <= 0.00001f) && !(tn > tf) && !(tn > tmax);
                    key[k] = hit ? ((__float_as_uint(tn > 0.0f ? tn : 0.0f) & ~3u) | (unsigned)k) : 0xffffffffu;
                }
                wf_cex(key[0], e[0], key[1], e[1]);
                wf_cex(key[2], e[2], key[3], e[3]);
                wf_cex(key[0], e[0], key[2], e[2]);
                wf_cex(key[1], e[1], key[3], e[3]);
                wf_cex(key[1], e[1], key[2], e[2]);
                if (key[0] == 0xffffffffu) {
                    pop = true;
                } else {
                    const int top = sp + 3 - (key[1] == 0xffffffffu ? 1 : 0) - (key[2] == 0xffffffffu ? 1 : 0) - (key[3] == 0xffffffffu ? 1 : 0);
#pragma unroll
                    for (int j = 3; j >= 1; --j)
                        if (key[j] != 0xffffffffu) {
                            const int at = top - j;
                            if (at < kWfStackLevels) stk[64 * at] = e[j];
                            else spill[64 * (at - kWfStackLevels)] = e[j];
                        }
                    sp = top;
                    cur = e[0];
                }
            }
            if (leaf) {
                // ---- a leaf: its first triangle, mesh.h:45-67
                const int prim = (int)(cur & 0x07ffffffu), left = (int)((cur >> 27) & 15u);
                const char *tp = tris + (size_t)prim * 48u;
                const float4 q0 = *reinterpret_cast<const float4 *>(tp);
                const float4 q1 = *reinterpret_cast<const float4 *>(tp + 16);
                const float e2z = *reinterpret_cast<const float *>(tp + 32);
                const V3 v1 = V3{q0.x, q0.y, q0.z};
                const V3 e1 = V3{q0.w, q1.x, q1.y};
                const V3 e2 = V3{q1.z, q1.w, e2z};
                const V3 s1 = cross(d, e2);
                const float divisor = dot(s1, e1);
                const float invDivisor = 1.0f / divisor;           // == (float)(1.0 / divisor): 53 >= 2 * 24 + 2 bits
                const V3 s = o - v1;
                const float b1 = dot(s, s1) * invDivisor;
                const V3 s2 = cross(s, e1);
                const float b2 = dot(d, s2) * invDivisor;
                const float tt = dot(e2, s2) * invDivisor;
                const bool accept = !(fabs_(divisor) < 1e-8f) && !(b1 < 0.0f || b1 > 1.0f) &&
                                    !(b2 < 0.0f || b1 + b2 > 1.0f) && !(tt < tmin_ray || tt > tmax);
                bool ended = false;
                if (accept) {
                    if (bprim < 0 || tt < bt || (tt == bt && prim > bprim)) {
                        bprim = prim;
                        bt = tt;
                        bb1 = b1;
                        bb2 = b2;
                    }
                    if (tt < tmax) tmax = tt;                      // (a NaN distance never becomes the interval's end)
                    ended = any_hit;                               // IntersectP: the first accepted triangle ends the ray
                }
                if (ended) {
                    cur = GPT_WIDE_NONE;
                    sp = 0;
                } else if (left > 0) {
                    cur = 0x80000000u | ((unsigned)(left - 1) << 27) | (unsigned)(prim + 1);
                } else {
                    pop = true;
                }
            }
            if (pop) {
                if (sp > 0) {
                    --sp;
                    cur = sp < kWfStackLevels ? stk[64 * sp] : spill[64 * (sp - kWfStackLevels)];
                } else {
                    cur = GPT_WIDE_NONE;
                }
            }
        } else {
            const bool want_tri = has && tri <= tri_last;
            const bool more_nodes = has && idx < end;
            const unsigned long long m_tri = ballot(want_tri), m_node = ballot(more_nodes) & ~m_tri;
            const bool node_trip = popc(m_node) >= (popc(m_tri) << 1);
            if (node_trip && more_nodes && !want_tri) {
                // ---- one node (bbox.h:77-96); threaded preorder == the reference's stack order (pathtracer.cu:221-252)
                const float4 a = *reinterpret_cast<const float4 *>(nodes + (unsigned)idx);
                const float4 b = *reinterpret_cast<const float4 *>(nodes + (unsigned)idx + 16);
                const float t1 = (a.x - o.x) * inv.x;
                const float t2 = (a.w - o.x) * inv.x;
                const float t3 = (a.y - o.y) * inv.y;
                const float t4 = (b.x - o.y) * inv.y;
                const float t5 = (a.z - o.z) * inv.z;
                const float t6 = (b.y - o.z) * inv.z;
                const float tn = fmax_(fmax_(fmin_(t1, t2), fmin_(t3, t4)), fmin_(t5, t6));
                const float tf = fmin_(fmin_(fmax_(t1, t2), fmax_(t3, t4)), fmax_(t5, t6));
                const bool box = !(tf <= 0.00001f) && !(tn > tf) && !(tn > tmax);
                const int link = __float_as_int(b.z);
                const int last = __float_as_int(b.w);
                const bool leaf = last >= 0;
                idx = (box || leaf) ? idx + 32 : link;
                if (box && leaf) {
                    tri = link;
                    tri_last = last;
                }
            }
            if (!node_trip && want_tri) {
                // ---- one triangle (mesh.h:45-67); the later of two equally near hits wins (:63 accepts tt == tmax)
                const int ti = tri;
                tri += 48;
                const float4 q0 = *reinterpret_cast<const float4 *>(tris + (unsigned)ti);
                const float4 q1 = *reinterpret_cast<const float4 *>(tris + (unsigned)ti + 16);
                const float e2z = *reinterpret_cast<const float *>(tris + (unsigned)ti + 32);
                const V3 v1 = V3{q0.x, q0.y, q0.z};
                const V3 e1 = V3{q0.w, q1.x, q1.y};
                const V3 e2 = V3{q1.z, q1.w, e2z};
                const V3 s1 = cross(d, e2);
                const float divisor = dot(s1, e1);
                const float invDivisor = 1.0f / divisor;
                const V3 s = o - v1;
                const float b1 = dot(s, s1) * invDivisor;
                const V3 s2 = cross(s, e1);
                const float b2 = dot(d, s2) * invDivisor;
                const float tt = dot(e2, s2) * invDivisor;
                const bool accept = !(fabs_(divisor) < 1e-8f) && !(b1 < 0.0f || b1 > 1.0f) &&
                                    !(b2 < 0.0f || b1 + b2 > 1.0f) && !(tt < tmin_ray || tt > tmax);
                if (accept) {
                    tmax = tt;
                    bprim = ti;
                    bb1 = b1;
                    bb2 = b2;
                    if (any_hit) {          // IntersectP: the first accepted triangle ends the ray
                        idx = end;
                        tri_last = -1;
                    }
                }
            }
        }
    }
}


// ---- the wide walk of the trace stage as a RAY STREAM ---------------------------------------------------------------------
// One lane per ray wastes the trace stage's lanes on the walk itself: a trip runs its node block for the lanes that are at a wide
// node and its triangle block for the lanes that are at a leaf, and a lane is at one or at the other (measured on the config-5
// stand-in: 45 lanes hold a ray, 22 of 64 are active per VALU instruction).  Here a ray does not belong to a lane.  A wave keeps up to
// kWfR rays in flight with ALL their walk state in its LDS block (origin, direction, 1 / direction, interval end, current entry,
// stack size, best hit, id; the stack's first kWfS levels as columns of the same block), and two lists name the slots whose next
// step is a wide node and those whose next step is a leaf.  A trip takes up to 64 slots off ONE list - every lane runs the same
// block - loads what that step needs, makes the step, writes back what changed and puts the slot on the list of its next step; a
// finished ray's result goes to hit[kind][path], its slot is free again and the next ray of the wave's current segment moves in.
// The walk of a ray is include/gpt_wide_bvh.h's, step for step (wf_trace_cxx above with the state in LDS): same bits.
// Rays still in flight when the segments are used up and few are left simply stay where they are for the next round (their result
// slot says kWfPending meanwhile): parking costs nothing here.
#ifndef PT_WF_STREAM_CXX
#define PT_WF_STREAM_CXX 0              // 1: the stream runs as the C++ specification below instead of its hand-scheduled twin
#endif
#ifndef PT_WF_STREAM_RAYS
#define PT_WF_STREAM_RAYS 96
#endif
#ifndef PT_WF_STREAM_LEVELS
#define PT_WF_STREAM_LEVELS 6
#endif
#ifndef PT_WF_STREAM_REFILL_T
#define PT_WF_STREAM_REFILL_T 16        // free slots that trigger a refill
#endif
#ifndef PT_WF_STREAM_PARK_T
#define PT_WF_STREAM_PARK_T 32          // segments used up and fewer rays than this in flight: they wait for the next round ...
#endif
#ifndef PT_WF_STREAM_DEPTH
#define PT_WF_STREAM_DEPTH 2            // batches a wave has on their way (1: the step follows its loads at once)
#endif
#ifndef PT_WF_STREAM_MIN_TRIPS
#define PT_WF_STREAM_MIN_TRIPS 16       // ... but every round moves its rays on by some trips
#endif
constexpr int kWfR = PT_WF_STREAM_RAYS, kWfS = PT_WF_STREAM_LEVELS;
enum { SF_OX, SF_OY, SF_OZ, SF_DX, SF_DY, SF_DZ, SF_IX, SF_IY, SF_IZ, SF_TMAX, SF_CUR, SF_SP, SF_BPRIM, SF_BT, SF_B1, SF_B2, SF_ID, SF_N };
struct WfStreamWave {
    uint32_t f[SF_N][kWfR];             // field-major: a trip's 64 slots are different words of one row (two-way bank conflicts at most)
    uint32_t stack[kWfS][kWfR];
    uint32_t qn[kWfR], ql[kWfR];        // slots whose next step is a wide node / a leaf (used as stacks)
    uint32_t freel[kWfR];
    uint32_t n_qn, n_ql, n_free, pad;   // (between two rounds)
};

__device__ __forceinline__ uint32_t wf_bperm(uint32_t v, int src_lane) { return (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v); }

// what a wave has asked memory for and not looked at yet: one batch of up to 64 rays, all at a wide node or all at a leaf (seven
// dwordx4 from the node / from the leaf's next two triangles - the copy behind the wide nodes, DevParams::wide_tris_off), and the
// rays that move into free slots (direction and origin)
struct WfFetch {
    uint32_t slot;
    int n;                    // wave-uniform: rays in the batch (lane i < n has one)
    bool node;                // wave-uniform
    float4 d0, d1, d2, d3, d4, d5, d6;
    int take;                 // wave-uniform: rays that move in (lane i < take has one)
    uint32_t new_id, new_slot;
    float4 nr, no;
};

__device__ __forceinline__ void wf_trace_stream(const DevParams &P, const WfParams &W, WfShared &sh, uint32_t chunk0, WfStreamWave &L, unsigned lane)
{
    const uint32_t np = W.n_paths;
    const float tmin_ray = P.eps;
    const char *wnodes = reinterpret_cast<const char *>(P.wide);
    const uint32_t tri_off = P.wide_tris_off;
    uint32_t *spill = W.spill + (size_t)(blockIdx.x * (unsigned)kWfWgWaves + (threadIdx.x >> 6)) * (uint32_t)kWfR * W.spill_levels;
    int qn = __builtin_amdgcn_readfirstlane((int)L.n_qn), ql = __builtin_amdgcn_readfirstlane((int)L.n_ql);
    int nfree = __builtin_amdgcn_readfirstlane((int)L.n_free);
    uint32_t seg0 = 0, seg1 = 0, seg2 = 0;          // the wave's current segment: id j in lane j % 64, register j / 64
    uint32_t cursor = 0, chunk_end = 0;
    bool exhausted = false, parking = false;
    int trips = 0;
#if PT_WF_PROBE == 2
    unsigned long long pr_nt = 0, pr_nl = 0, pr_lt = 0, pr_ll = 0, pr_in = 0;       // node trips, lanes in them, leaf trips, lanes in them, rays in flight
    unsigned long long pr_t = __builtin_readcyclecounter(), pr_pre = 0, pr_wait = 0, pr_proc = 0;      // cycles: up to the loads, until they are back, the step
#define PT_WFS_TICK(acc) { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - pr_t; pr_t = now_; }
#else
#define PT_WFS_TICK(acc)
#endif

    // ---- ask for the next batch (`others` rays are in the batch that is still on its way) and for the rays that move in
    auto issue = [&](WfFetch &F, int others) {
        F.n = 0;
        F.take = 0;
        F.node = true;
        F.slot = 0;
        F.new_id = F.new_slot = 0;
        if (parking) return;
        if (!exhausted && nfree >= PT_WF_STREAM_REFILL_T) {
            while (cursor >= chunk_end && !exhausted) {
                uint32_t c = 0;
                if (lane == 0) c = atomicAdd(&sh.trace_next, 1u);
                c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
                if (c >= (uint32_t)kWfWgChunks) { exhausted = true; break; }
                const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh.seg_count[c]);
                const uint32_t *seg = W.rayq + (size_t)(chunk0 + c) * (uint32_t)kWfSegRays;
                seg0 = lane < cnt ? seg[lane] : 0u;
                seg1 = lane + 64u < cnt ? seg[lane + 64u] : 0u;
                seg2 = lane + 128u < cnt ? seg[lane + 128u] : 0u;
                cursor = 0;
                chunk_end = cnt;
            }
            if (!exhausted) {
                const int left = (int)(chunk_end - cursor);
                int take = nfree < left ? nfree : left;
                if (take > 64) take = 64;
                const uint32_t q = cursor + lane;
                const uint32_t a = wf_bperm(seg0, (int)(q & 63u)), b = wf_bperm(seg1, (int)(q & 63u)), c2 = wf_bperm(seg2, (int)(q & 63u));
                F.new_id = (q >> 6) == 0u ? a : ((q >> 6) == 1u ? b : c2);
                if ((int)lane < take) {
                    F.new_slot = L.freel[nfree - 1 - (int)lane];
                    const uint32_t path = F.new_id & kWfPathMask, kind = (F.new_id >> kWfKindShift) & 3u;
                    F.nr = W.ray[kind * np + path];
                    F.no = W.org[path];
                }
                cursor += (uint32_t)take;
                nfree -= take;
                F.take = take;
            }
        }
        const int n_in = qn + ql;
        if (exhausted && F.take == 0 && n_in + others < PT_WF_STREAM_PARK_T && trips >= PT_WF_STREAM_MIN_TRIPS && n_in + others > 0) {
            parking = true;             // the rest waits for the next round where it is (see below)
            return;
        }
        if (n_in == 0) return;
        ++trips;
        // a full batch of either kind if there is one, else the longer list (a node step costs about twice a leaf step: it goes first)
        F.node = qn >= 64 || (ql < 64 && qn >= ql);
        const int have = F.node ? qn : ql;
        const int n = have < 64 ? have : 64;
#if PT_WF_PROBE == 2
        pr_in += (unsigned)(n_in + others);
        if (F.node) { pr_nt++; pr_nl += (unsigned)n; } else { pr_lt++; pr_ll += (unsigned)n; }
#endif
        F.n = n;
        if ((int)lane < n) {
            F.slot = F.node ? L.qn[qn - n + (int)lane] : L.ql[ql - n + (int)lane];
            const unsigned cur = L.f[SF_CUR][F.slot];
            const uint32_t off = F.node ? cur : tri_off + (cur & 0x07ffffffu) * 48u;
            const float4 *g = reinterpret_cast<const float4 *>(wnodes + (size_t)off);
            F.d0 = g[0]; F.d1 = g[1]; F.d2 = g[2]; F.d3 = g[3]; F.d4 = g[4]; F.d5 = g[5]; F.d6 = g[6];
        }
        if (F.node) qn -= n; else ql -= n;
        PT_WFS_TICK(pr_pre)
    };

    // ---- make the batch's step with what has arrived, put its rays on the list of their next step, let the new rays in
    auto process = [&](const WfFetch &F) {
#if PT_WF_PROBE == 2
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PT_WFS_TICK(pr_wait)
#endif
        if (F.n > 0) {
            const bool act = (int)lane < F.n;
            const uint32_t slot = F.slot;
            bool to_node = false, to_leaf = false, fin = false;
            if (F.node) {
                if (act) {
                    const V3 o = V3{__uint_as_float(L.f[SF_OX][slot]), __uint_as_float(L.f[SF_OY][slot]), __uint_as_float(L.f[SF_OZ][slot])};
                    const V3 inv = V3{__uint_as_float(L.f[SF_IX][slot]), __uint_as_float(L.f[SF_IY][slot]), __uint_as_float(L.f[SF_IZ][slot])};
                    const float tmax = __uint_as_float(L.f[SF_TMAX][slot]);
                    unsigned cur;
                    int sp = (int)L.f[SF_SP][slot];
                    // ---- a wide node: four boxes, bbox.h:77-96 each
                    const float4 lx = F.d0, ly = F.d1, lz = F.d2, hx = F.d3, hy = F.d4, hz = F.d5;
                    const uint4 en = make_uint4(__float_as_uint(F.d6.x), __float_as_uint(F.d6.y), __float_as_uint(F.d6.z), __float_as_uint(F.d6.w));
                    const float blx[4] = {lx.x, lx.y, lx.z, lx.w}, bly[4] = {ly.x, ly.y, ly.z, ly.w}, blz[4] = {lz.x, lz.y, lz.z, lz.w};
                    const float bhx[4] = {hx.x, hx.y, hx.z, hx.w}, bhy[4] = {hy.x, hy.y, hy.z, hy.w}, bhz[4] = {hz.x, hz.y, hz.z, hz.w};
                    unsigned e[4] = {en.x, en.y, en.z, en.w}, key[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float t1 = (blx[k] - o.x) * inv.x;
                        const float t2 = (bhx[k] - o.x) * inv.x;
                        const float t3 = (bly[k] - o.y) * inv.y;
                        const float t4 = (bhy[k] - o.y) * inv.y;
                        const float t5 = (blz[k] - o.z) * inv.z;
                        const float t6 = (bhz[k] - o.z) * inv.z;
                        const float tn = fmax_(fmax_(fmin_(t1, t2), fmin_(t3, t4)), fmin_(t5, t6));
                        const float tf = fmin_(fmin_(fmax_(t1, t2), fmax_(t3, t4)), fmax_(t5, t6));
                        const bool hit = e[k] != GPT_WIDE_NONE && !(tf <= 0.00001f) && !(tn > tf) && !(tn > tmax);
                        key[k] = hit ? ((__float_as_uint(tn > 0.0f ? tn : 0.0f) & ~3u) | (unsigned)k) : 0xffffffffu;
                    }
                    wf_cex(key[0], e[0], key[1], e[1]);
                    wf_cex(key[2], e[2], key[3], e[3]);
                    wf_cex(key[0], e[0], key[2], e[2]);
                    wf_cex(key[1], e[1], key[3], e[3]);
                    wf_cex(key[1], e[1], key[2], e[2]);
                    if (key[0] == 0xffffffffu) {
                        if (sp > 0) {
                            --sp;
                            cur = sp < kWfS ? L.stack[sp][slot] : spill[(size_t)(sp - kWfS) * (uint32_t)kWfR + slot];
                        } else {
                            cur = GPT_WIDE_NONE;
                        }
                    } else {
                        const int top = sp + 3 - (key[1] == 0xffffffffu ? 1 : 0) - (key[2] == 0xffffffffu ? 1 : 0) - (key[3] == 0xffffffffu ? 1 : 0);
#pragma unroll
                        for (int jj = 3; jj >= 1; --jj)
                            if (key[jj] != 0xffffffffu) {
                                const int at = top - jj;
                                if (at < kWfS) L.stack[at][slot] = e[jj];
                                else spill[(size_t)(at - kWfS) * (uint32_t)kWfR + slot] = e[jj];
                            }
                        sp = top;
                        cur = e[0];
                    }
                    L.f[SF_CUR][slot] = cur;
                    L.f[SF_SP][slot] = (uint32_t)sp;
                    fin = cur == GPT_WIDE_NONE;
                    to_leaf = !fin && (cur >> 31) != 0u;
                    to_node = !fin && (cur >> 31) == 0u;
                }
            } else {
                if (act) {
                    const V3 o = V3{__uint_as_float(L.f[SF_OX][slot]), __uint_as_float(L.f[SF_OY][slot]), __uint_as_float(L.f[SF_OZ][slot])};
                    const V3 d = V3{__uint_as_float(L.f[SF_DX][slot]), __uint_as_float(L.f[SF_DY][slot]), __uint_as_float(L.f[SF_DZ][slot])};
                    float tmax = __uint_as_float(L.f[SF_TMAX][slot]);
                    unsigned cur = L.f[SF_CUR][slot];
                    int sp = (int)L.f[SF_SP][slot];
                    const bool any_hit = (L.f[SF_ID][slot] & kWfAnyHit) != 0u;
                    int bprim = (int)L.f[SF_BPRIM][slot];
                    float bt = __uint_as_float(L.f[SF_BT][slot]), bb1 = 0.f, bb2 = 0.f;
                    bool better = false, ended = false;
                    int prim = (int)(cur & 0x07ffffffu), left = (int)((cur >> 27) & 15u);        // left = triangles of the leaf after this one
                    // ---- a leaf: its next triangle and, if the ray goes on, the one after it (mesh.h:45-67 each, each against the
                    // interval it would see in a trip of its own)
#pragma unroll
                    for (int rep = 0; rep < 2; ++rep) {
                        if (rep == 1 && (ended || left <= 0)) break;
                        if (rep == 1) { ++prim; --left; }
                        const float4 q0 = rep ? F.d3 : F.d0, q1 = rep ? F.d4 : F.d1;
                        const float e2z = rep ? F.d5.x : F.d2.x;
                        const V3 v1 = V3{q0.x, q0.y, q0.z};
                        const V3 e1 = V3{q0.w, q1.x, q1.y};
                        const V3 e2 = V3{q1.z, q1.w, e2z};
                        const V3 s1 = cross(d, e2);
                        const float divisor = dot(s1, e1);
                        const float invDivisor = 1.0f / divisor;           // == (float)(1.0 / divisor): 53 >= 2 * 24 + 2 bits
                        const V3 s = o - v1;
                        const float b1 = dot(s, s1) * invDivisor;
                        const V3 s2 = cross(s, e1);
                        const float b2 = dot(d, s2) * invDivisor;
                        const float tt = dot(e2, s2) * invDivisor;
                        const bool accept = !(fabs_(divisor) < 1e-8f) && !(b1 < 0.0f || b1 > 1.0f) &&
                                            !(b2 < 0.0f || b1 + b2 > 1.0f) && !(tt < tmin_ray || tt > tmax);
                        if (accept) {
                            if (bprim < 0 || tt < bt || (tt == bt && prim > bprim)) {
                                bprim = prim;
                                bt = tt;
                                bb1 = b1;
                                bb2 = b2;
                                better = true;
                            }
                            if (tt < tmax) tmax = tt;                      // (a NaN distance never becomes the interval's end)
                            ended = any_hit;                               // IntersectP: the first accepted triangle ends the ray
                        }
                    }
                    if (better) {
                        L.f[SF_BPRIM][slot] = (uint32_t)bprim;
                        L.f[SF_BT][slot] = __float_as_uint(bt);
                        L.f[SF_B1][slot] = __float_as_uint(bb1);
                        L.f[SF_B2][slot] = __float_as_uint(bb2);
                    }
                    L.f[SF_TMAX][slot] = __float_as_uint(tmax);
                    if (ended) {
                        cur = GPT_WIDE_NONE;
                        sp = 0;
                    } else if (left > 0) {
                        cur = 0x80000000u | ((unsigned)(left - 1) << 27) | (unsigned)(prim + 1);
                    } else if (sp > 0) {
                        --sp;
                        cur = sp < kWfS ? L.stack[sp][slot] : spill[(size_t)(sp - kWfS) * (uint32_t)kWfR + slot];
                    } else {
                        cur = GPT_WIDE_NONE;
                    }
                    L.f[SF_CUR][slot] = cur;
                    L.f[SF_SP][slot] = (uint32_t)sp;
                    fin = cur == GPT_WIDE_NONE;
                    to_leaf = !fin && (cur >> 31) != 0u;
                    to_node = !fin && (cur >> 31) == 0u;
                }
            }
            // ---- where the batch's rays go next
            const unsigned long long m_n = ballot(to_node), m_l = ballot(to_leaf), m_f = ballot(fin);
            if (to_node) L.qn[qn + lane_rank(m_n)] = slot;
            if (to_leaf) L.ql[ql + lane_rank(m_l)] = slot;
            qn += popc(m_n);
            ql += popc(m_l);
            if (m_f != 0ull) {
                if (fin) {
                    const uint32_t id = L.f[SF_ID][slot];
                    const int bprim = (int)L.f[SF_BPRIM][slot];
                    // a miss reports the end of the interval
                    W.hit[((id >> kWfKindShift) & 3u) * np + (id & kWfPathMask)] =
                        make_float4(__int_as_float(bprim), __uint_as_float(bprim < 0 ? L.f[SF_TMAX][slot] : L.f[SF_BT][slot]),
                                    __uint_as_float(L.f[SF_B1][slot]), __uint_as_float(L.f[SF_B2][slot]));
                    L.freel[nfree + lane_rank(m_f)] = slot;
                }
                nfree += popc(m_f);
            }
        }
        // ---- the new rays start at the root
        if (F.take > 0) {
            if ((int)lane < F.take) {
                const uint32_t ns = F.new_slot;
                const V3 d = xyz(F.nr);
                L.f[SF_OX][ns] = __float_as_uint(F.no.x); L.f[SF_OY][ns] = __float_as_uint(F.no.y); L.f[SF_OZ][ns] = __float_as_uint(F.no.z);
                L.f[SF_DX][ns] = __float_as_uint(d.x); L.f[SF_DY][ns] = __float_as_uint(d.y); L.f[SF_DZ][ns] = __float_as_uint(d.z);
                L.f[SF_IX][ns] = __float_as_uint(1.f / d.x);      // bbox.h:79 computes 1/d at every node visit: the same quotient
                L.f[SF_IY][ns] = __float_as_uint(1.f / d.y);
                L.f[SF_IZ][ns] = __float_as_uint(1.f / d.z);
                L.f[SF_TMAX][ns] = __float_as_uint(F.nr.w);
                L.f[SF_CUR][ns] = 0u;
                L.f[SF_SP][ns] = 0u;
                L.f[SF_BPRIM][ns] = 0xffffffffu;
                L.f[SF_BT][ns] = 0u; L.f[SF_B1][ns] = 0u; L.f[SF_B2][ns] = 0u;
                L.f[SF_ID][ns] = F.new_id;
                L.qn[qn + (int)lane] = ns;
            }
            qn += F.take;
        }
        wave_lds_fence();
        PT_WFS_TICK(pr_proc)
    };

    // two batches are on their way at any time: while one's step is computed the other's loads are in flight
    WfFetch A, B;
    issue(A, 0);
    for (;;) {
#if PT_WF_STREAM_DEPTH == 1
        process(A);
        if (A.n == 0 && A.take == 0 && qn + ql == 0 && (exhausted || parking)) break;
        if (parking) break;
        issue(A, 0);
#else
        issue(B, A.n);
        process(A);
        if (B.n == 0 && B.take == 0 && ((qn + ql == 0 && exhausted) || parking)) break;
        issue(A, B.n);
        process(B);
        if (A.n == 0 && A.take == 0 && ((qn + ql == 0 && exhausted) || parking)) break;
#endif
    }
    if (parking) {
        // ---- rays left for the next round: the shade phase lets their paths sit the round out
        const int n_in = qn + ql;
        for (int j = (int)lane; j < n_in; j += 64) {
            const uint32_t slot = j < qn ? L.qn[j] : L.ql[j - qn];
            const uint32_t id = L.f[SF_ID][slot];
            reinterpret_cast<int *>(W.hit + ((id >> kWfKindShift) & 3u) * np + (id & kWfPathMask))[0] = kWfPending;
        }
        if (lane == 0 && n_in > 0) sh.parked = 1u;
    }
    if (lane == 0) { L.n_qn = (uint32_t)qn; L.n_ql = (uint32_t)ql; L.n_free = (uint32_t)nfree; }
#if PT_WF_PROBE == 2
    if (lane == 0) {
        atomicAdd(&P.counters[6], pr_nt); atomicAdd(&P.counters[7], pr_nl); atomicAdd(&P.counters[8], pr_lt); atomicAdd(&P.counters[9], pr_ll);
        atomicAdd(&P.counters[10], pr_in);
        atomicAdd(&P.counters[11], pr_pre); atomicAdd(&P.counters[12], pr_wait); atomicAdd(&P.counters[13], pr_proc);
    }
#endif
    wave_lds_fence();
}


// 1 / direction as the IEEE quotient (bbox.h:79 divides): v[8:10] = 1.0 / v[4:6]; temporaries v[33:37], s[66:67], vcc
#define PT_WF_ASM_INV_DIR \
        "v_div_scale_f32 v33, s[66:67], v4, v4, 1.0\n" \
        "v_div_scale_f32 v34, vcc, 1.0, v4, 1.0\n" \
        "v_rcp_f32_e32 v35, v33\n" \
        "s_nop 0\n" \
        "v_fma_f32 v36, -v33, v35, 1.0\n" \
        "v_fmac_f32_e32 v35, v36, v35\n" \
        "v_mul_f32_e32 v37, v34, v35\n" \
        "v_fma_f32 v36, -v33, v37, v34\n" \
        "v_fmac_f32_e32 v37, v36, v35\n" \
        "v_fma_f32 v33, -v33, v37, v34\n" \
        "v_div_fmas_f32 v33, v33, v35, v37\n" \
        "v_div_fixup_f32 v8, v33, v4, 1.0\n" \
        "v_div_scale_f32 v33, s[66:67], v5, v5, 1.0\n" \
        "v_div_scale_f32 v34, vcc, 1.0, v5, 1.0\n" \
        "v_rcp_f32_e32 v35, v33\n" \
        "s_nop 0\n" \
        "v_fma_f32 v36, -v33, v35, 1.0\n" \
        "v_fmac_f32_e32 v35, v36, v35\n" \
        "v_mul_f32_e32 v37, v34, v35\n" \
        "v_fma_f32 v36, -v33, v37, v34\n" \
        "v_fmac_f32_e32 v37, v36, v35\n" \
        "v_fma_f32 v33, -v33, v37, v34\n" \
        "v_div_fmas_f32 v33, v33, v35, v37\n" \
        "v_div_fixup_f32 v9, v33, v5, 1.0\n" \
        "v_div_scale_f32 v33, s[66:67], v6, v6, 1.0\n" \
        "v_div_scale_f32 v34, vcc, 1.0, v6, 1.0\n" \
        "v_rcp_f32_e32 v35, v33\n" \
        "s_nop 0\n" \
        "v_fma_f32 v36, -v33, v35, 1.0\n" \
        "v_fmac_f32_e32 v35, v36, v35\n" \
        "v_mul_f32_e32 v37, v34, v35\n" \
        "v_fma_f32 v36, -v33, v37, v34\n" \
        "v_fmac_f32_e32 v37, v36, v35\n" \
        "v_fma_f32 v33, -v33, v37, v34\n" \
        "v_div_fmas_f32 v33, v33, v35, v37\n" \
        "v_div_fixup_f32 v10, v33, v6, 1.0\n"

// The trip of the hand-scheduled walks as two pieces of text, used by wf_trace_wide_asm (one lane per ray: the lanes at a wide node are
// s[62:63], those at a leaf s[60:61]) and by wf_trace_stream_asm (a batch of rays that are all at a wide node or all at a leaf: one of the two
// masks is empty).  In: v[0:2] origin, v[4:6] direction, v[8:10] 1 / direction, v11 id (bit 31: any hit), v12 current entry, v13 stack size,
// v14 end of the interval, v16 the ray's spill column, v18 its LDS stack column (moved back by three levels), v[20:23] best hit; out: v12,
// v13, v14, v[20:23] and the stack.  PT_WF_LVL(dst, level, column) forms the address of a level of a column, PT_WF_O1..3 are the byte
// offsets of one, two and three levels: the two walks differ in how far apart the levels of a column are.
#define PT_WF_ASM_TRIP \
        "s_mov_b64 exec, s[60:61]\n" \
        "v_and_b32_e32 v53, 0x7ffffff, v12\n"              /* the leaf's first triangle */ \
        "v_lshlrev_b32_e32 v54, 4, v53\n" \
        "v_lshl_add_u32 v54, v53, 5, v54\n"                /* * 48 */ \
        "v_add_u32_e32 v54, %[trioff], v54\n" \
        "s_mov_b64 exec, s[62:63]\n" \
        "v_mov_b32_e32 v54, v12\n" \
        "s_or_b64 exec, s[60:61], s[62:63]\n" \
        "global_load_dwordx4 v[24:27], v54, %[nodes]\n" \
        "global_load_dwordx4 v[36:39], v54, %[nodes] offset:48\n" \
        "global_load_dwordx4 v[28:31], v54, %[nodes] offset:16\n" \
        "global_load_dwordx4 v[40:43], v54, %[nodes] offset:64\n" \
        "global_load_dwordx4 v[32:35], v54, %[nodes] offset:32\n" \
        "global_load_dwordx4 v[44:47], v54, %[nodes] offset:80\n" \
        "global_load_dwordx4 v[48:51], v54, %[nodes] offset:96\n" \
        "s_mov_b64 s[78:79], 0\n" \
        "s_mov_b64 s[82:83], 0\n" \
        "s_mov_b64 exec, s[62:63]\n" \
        "s_cbranch_execz TQ_LEAF_%=\n" \
        "s_waitcnt vmcnt(0)\n" \
        /* ---------------------------------------------------------------- wide node: four boxes (exec = s[62:63]) */ \
        /* child k's six plane values are worked on in place (v24+k, v28+k, ... v44+k); v52 is the one temporary */ \
        "v_sub_f32_e32 v24, v24, v0\n" \
        "v_sub_f32_e32 v36, v36, v0\n" \
        "v_sub_f32_e32 v28, v28, v1\n" \
        "v_sub_f32_e32 v32, v32, v2\n" \
        "v_sub_f32_e32 v40, v40, v1\n" \
        "v_sub_f32_e32 v44, v44, v2\n" \
        "v_mul_f32_e32 v24, v8, v24\n" \
        "v_mul_f32_e32 v36, v8, v36\n" \
        "v_mul_f32_e32 v28, v9, v28\n" \
        "v_mul_f32_e32 v40, v9, v40\n" \
        "v_mul_f32_e32 v32, v10, v32\n" \
        "v_mul_f32_e32 v44, v10, v44\n" \
        "v_min_f32_e32 v52, v24, v36\n" \
        "v_max_f32_e32 v24, v24, v36\n" \
        "v_min_f32_e32 v36, v28, v40\n" \
        "v_max_f32_e32 v28, v28, v40\n" \
        "v_min_f32_e32 v40, v32, v44\n" \
        "v_max_f32_e32 v32, v32, v44\n" \
        "v_min3_f32 v24, v24, v28, v32\n" \
        "v_max3_f32 v52, v52, v36, v40\n" \
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v24\n" \
        "v_min_f32_e32 v24, v24, v14\n" \
        "v_cmp_nlt_f32_e64 s[66:67], v24, v52\n" \
        "v_max_f32_e32 v52, 0, v52\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ne_u32_e32 vcc, -1, v48\n" \
        "v_and_or_b32 v52, v52, -4, 0\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cndmask_b32_e64 v24, -1, v52, s[66:67]\n" \
        "v_sub_f32_e32 v25, v25, v0\n" \
        "v_sub_f32_e32 v37, v37, v0\n" \
        "v_sub_f32_e32 v29, v29, v1\n" \
        "v_sub_f32_e32 v33, v33, v2\n" \
        "v_sub_f32_e32 v41, v41, v1\n" \
        "v_sub_f32_e32 v45, v45, v2\n" \
        "v_mul_f32_e32 v25, v8, v25\n" \
        "v_mul_f32_e32 v37, v8, v37\n" \
        "v_mul_f32_e32 v29, v9, v29\n" \
        "v_mul_f32_e32 v41, v9, v41\n" \
        "v_mul_f32_e32 v33, v10, v33\n" \
        "v_mul_f32_e32 v45, v10, v45\n" \
        "v_min_f32_e32 v52, v25, v37\n" \
        "v_max_f32_e32 v25, v25, v37\n" \
        "v_min_f32_e32 v37, v29, v41\n" \
        "v_max_f32_e32 v29, v29, v41\n" \
        "v_min_f32_e32 v41, v33, v45\n" \
        "v_max_f32_e32 v33, v33, v45\n" \
        "v_min3_f32 v25, v25, v29, v33\n" \
        "v_max3_f32 v52, v52, v37, v41\n" \
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v25\n" \
        "v_min_f32_e32 v25, v25, v14\n" \
        "v_cmp_nlt_f32_e64 s[66:67], v25, v52\n" \
        "v_max_f32_e32 v52, 0, v52\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ne_u32_e32 vcc, -1, v49\n" \
        "v_and_or_b32 v52, v52, -4, 1\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cndmask_b32_e64 v25, -1, v52, s[66:67]\n" \
        "v_sub_f32_e32 v26, v26, v0\n" \
        "v_sub_f32_e32 v38, v38, v0\n" \
        "v_sub_f32_e32 v30, v30, v1\n" \
        "v_sub_f32_e32 v34, v34, v2\n" \
        "v_sub_f32_e32 v42, v42, v1\n" \
        "v_sub_f32_e32 v46, v46, v2\n" \
        "v_mul_f32_e32 v26, v8, v26\n" \
        "v_mul_f32_e32 v38, v8, v38\n" \
        "v_mul_f32_e32 v30, v9, v30\n" \
        "v_mul_f32_e32 v42, v9, v42\n" \
        "v_mul_f32_e32 v34, v10, v34\n" \
        "v_mul_f32_e32 v46, v10, v46\n" \
        "v_min_f32_e32 v52, v26, v38\n" \
        "v_max_f32_e32 v26, v26, v38\n" \
        "v_min_f32_e32 v38, v30, v42\n" \
        "v_max_f32_e32 v30, v30, v42\n" \
        "v_min_f32_e32 v42, v34, v46\n" \
        "v_max_f32_e32 v34, v34, v46\n" \
        "v_min3_f32 v26, v26, v30, v34\n" \
        "v_max3_f32 v52, v52, v38, v42\n" \
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v26\n" \
        "v_min_f32_e32 v26, v26, v14\n" \
        "v_cmp_nlt_f32_e64 s[66:67], v26, v52\n" \
        "v_max_f32_e32 v52, 0, v52\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ne_u32_e32 vcc, -1, v50\n" \
        "v_and_or_b32 v52, v52, -4, 2\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cndmask_b32_e64 v26, -1, v52, s[66:67]\n" \
        "v_sub_f32_e32 v27, v27, v0\n" \
        "v_sub_f32_e32 v39, v39, v0\n" \
        "v_sub_f32_e32 v31, v31, v1\n" \
        "v_sub_f32_e32 v35, v35, v2\n" \
        "v_sub_f32_e32 v43, v43, v1\n" \
        "v_sub_f32_e32 v47, v47, v2\n" \
        "v_mul_f32_e32 v27, v8, v27\n" \
        "v_mul_f32_e32 v39, v8, v39\n" \
        "v_mul_f32_e32 v31, v9, v31\n" \
        "v_mul_f32_e32 v43, v9, v43\n" \
        "v_mul_f32_e32 v35, v10, v35\n" \
        "v_mul_f32_e32 v47, v10, v47\n" \
        "v_min_f32_e32 v52, v27, v39\n" \
        "v_max_f32_e32 v27, v27, v39\n" \
        "v_min_f32_e32 v39, v31, v43\n" \
        "v_max_f32_e32 v31, v31, v43\n" \
        "v_min_f32_e32 v43, v35, v47\n" \
        "v_max_f32_e32 v35, v35, v47\n" \
        "v_min3_f32 v27, v27, v31, v35\n" \
        "v_max3_f32 v52, v52, v39, v43\n" \
        "v_cmp_nge_f32_e32 vcc, 0x3727c5ac, v27\n" \
        "v_min_f32_e32 v27, v27, v14\n" \
        "v_cmp_nlt_f32_e64 s[66:67], v27, v52\n" \
        "v_max_f32_e32 v52, 0, v52\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ne_u32_e32 vcc, -1, v51\n" \
        "v_and_or_b32 v52, v52, -4, 3\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cndmask_b32_e64 v27, -1, v52, s[66:67]\n" \
        /* five-exchange sorting network on (key, entry); registers are renamed from exchange to exchange (a child's key sits in the \
           register its lo.x plane value came in) */ \
        "v_cmp_lt_u32_e32 vcc, v25, v24\n" \
        "v_min_u32_e32 v52, v24, v25\n" \
        "v_max_u32_e32 v25, v24, v25\n" \
        "v_cndmask_b32_e32 v53, v48, v49, vcc\n" \
        "v_cndmask_b32_e32 v49, v49, v48, vcc\n" \
        "v_cmp_lt_u32_e32 vcc, v27, v26\n" \
        "v_min_u32_e32 v24, v26, v27\n" \
        "v_max_u32_e32 v27, v26, v27\n" \
        "v_cndmask_b32_e32 v48, v50, v51, vcc\n" \
        "v_cndmask_b32_e32 v51, v51, v50, vcc\n" \
        "v_cmp_lt_u32_e32 vcc, v24, v52\n" \
        "v_min_u32_e32 v26, v52, v24\n" \
        "v_max_u32_e32 v24, v52, v24\n" \
        "v_cndmask_b32_e32 v50, v53, v48, vcc\n" \
        "v_cndmask_b32_e32 v48, v48, v53, vcc\n" \
        "v_cmp_lt_u32_e32 vcc, v27, v25\n" \
        "v_min_u32_e32 v52, v25, v27\n" \
        "v_max_u32_e32 v27, v25, v27\n" \
        "v_cndmask_b32_e32 v53, v49, v51, vcc\n" \
        "v_cndmask_b32_e32 v51, v51, v49, vcc\n" \
        "v_cmp_lt_u32_e32 vcc, v24, v52\n" \
        "v_min_u32_e32 v25, v52, v24\n" \
        "v_max_u32_e32 v24, v52, v24\n" \
        "v_cndmask_b32_e32 v49, v53, v48, vcc\n" \
        "v_cndmask_b32_e32 v48, v48, v53, vcc\n" \
        /* sorted: keys v26 <= v25 <= v24 <= v27, entries v50, v49, v48, v51 (a child that is not hit: key -1, sorts last) */ \
        "v_cmp_ne_u32_e64 s[66:67], -1, v25\n" \
        "v_cmp_ne_u32_e64 s[68:69], -1, v24\n" \
        "v_cmp_ne_u32_e64 s[72:73], -1, v27\n" \
        "v_cmp_ne_u32_e64 s[80:81], -1, v26\n"             /* the node has a hit child */ \
        "s_nop 0\n" \
        "v_addc_co_u32_e64 v28, s[74:75], v13, 0, s[66:67]\n" \
        "v_addc_co_u32_e64 v28, s[74:75], v28, 0, s[68:69]\n" \
        "v_addc_co_u32_e64 v28, s[74:75], v28, 0, s[72:73]\n"      /* the new stack size: the nearest child is not pushed */ \
        /* the others are pushed farthest first: sorted child j ends at level size' - j */ \
        "v_cmp_lt_u32_e64 s[74:75], %[depth], v28\n" \
        PT_WF_LVL("v29", "v28", "v18")                /* address of level size' - 3 */ \
        "s_cmp_lg_u64 s[74:75], 0\n" \
        "s_cbranch_scc1 TQ_PUSH_SLOW_%=\n" \
        "s_mov_b64 exec, s[72:73]\n" \
        "ds_write_b32 v29, v51\n" \
        "s_mov_b64 exec, s[68:69]\n" \
        "ds_write_b32 v29, v48 offset:" PT_WF_O1 "\n" \
        "s_mov_b64 exec, s[66:67]\n" \
        "ds_write_b32 v29, v49 offset:" PT_WF_O2 "\n" \
        "TQ_PUSHED_%=:\n" \
        "s_mov_b64 exec, s[62:63]\n" \
        "v_cndmask_b32_e64 v13, v13, v28, s[80:81]\n" \
        "v_cndmask_b32_e64 v12, v12, v50, s[80:81]\n"      /* the nearest hit child is the current entry */ \
        "s_andn2_b64 s[78:79], s[62:63], s[80:81]\n"       /* the lanes without a hit child pop */ \
        /* ---------------------------------------------------------------- leaf: one triangle (exec = s[60:61]) */ \
        "s_branch TQ_LEAF_GO_%=\n" \
        "TQ_LEAF_%=:\n"                                     /* no lane at a wide node: the triangle fetches have not been waited for */ \
        "s_waitcnt vmcnt(0)\n" \
        "TQ_LEAF_GO_%=:\n" \
        "s_mov_b64 exec, s[60:61]\n" \
        "s_cbranch_execz TQ_POP_%=\n" \
        "v_mul_f32_e32 v33, v5, v32\n" \
        "v_mul_f32_e32 v50, v6, v31\n" \
        "v_sub_f32_e32 v33, v33, v50\n" \
        "v_mul_f32_e32 v34, v6, v30\n" \
        "v_mul_f32_e32 v50, v4, v32\n" \
        "v_sub_f32_e32 v34, v34, v50\n" \
        "v_mul_f32_e32 v35, v4, v31\n" \
        "v_mul_f32_e32 v50, v5, v30\n" \
        "v_sub_f32_e32 v35, v35, v50\n" \
        "v_mul_f32_e32 v45, v33, v27\n" \
        "v_mul_f32_e32 v50, v34, v28\n" \
        "v_add_f32_e32 v45, v45, v50\n" \
        "v_mul_f32_e32 v50, v35, v29\n" \
        "v_add_f32_e32 v45, v45, v50\n" \
        "v_rcp_f32_e32 v47, v45\n" \
        "v_sub_f32_e32 v24, v0, v24\n" \
        "v_sub_f32_e32 v25, v1, v25\n" \
        "v_sub_f32_e32 v26, v2, v26\n" \
        "v_cmp_nle_f32_e64 s[66:67], abs(v45), s77\n" \
        "v_fma_f32 v49, -v45, v47, 1.0\n" \
        "v_fma_f32 v46, v49, v47, v47\n" \
        "s_cmp_lg_u64 s[66:67], 0\n" \
        "s_cbranch_scc1 TQ_DIV_IEEE_%=\n" \
        "TQ_DIV_DONE_%=:\n" \
        "v_mul_f32_e32 v51, v24, v33\n" \
        "v_mul_f32_e32 v50, v25, v34\n" \
        "v_add_f32_e32 v51, v51, v50\n" \
        "v_mul_f32_e32 v50, v26, v35\n" \
        "v_add_f32_e32 v51, v51, v50\n" \
        "v_mul_f32_e32 v33, v25, v29\n" \
        "v_mul_f32_e32 v50, v26, v28\n" \
        "v_sub_f32_e32 v33, v33, v50\n" \
        "v_mul_f32_e32 v34, v26, v27\n" \
        "v_mul_f32_e32 v50, v24, v29\n" \
        "v_sub_f32_e32 v34, v34, v50\n" \
        "v_mul_f32_e32 v35, v24, v28\n" \
        "v_mul_f32_e32 v50, v25, v27\n" \
        "v_sub_f32_e32 v35, v35, v50\n" \
        "v_mul_f32_e32 v51, v51, v46\n"                    /* b1 */ \
        "v_mul_f32_e32 v47, v4, v33\n" \
        "v_mul_f32_e32 v50, v5, v34\n" \
        "v_add_f32_e32 v47, v47, v50\n" \
        "v_mul_f32_e32 v50, v6, v35\n" \
        "v_add_f32_e32 v47, v47, v50\n" \
        "v_mul_f32_e32 v47, v47, v46\n"                    /* b2 */ \
        "v_cmp_nlt_f32_e64 s[66:67], abs(v45), s76\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v51\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v51\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v47\n" \
        "v_add_f32_e32 v50, v51, v47\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v50\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TQ_TRI_END_%=\n" \
        "v_mul_f32_e32 v48, v30, v33\n" \
        "v_mul_f32_e32 v50, v31, v34\n" \
        "v_add_f32_e32 v48, v48, v50\n" \
        "v_mul_f32_e32 v50, v32, v35\n" \
        "v_add_f32_e32 v48, v48, v50\n" \
        "v_mul_f32_e32 v48, v48, v46\n"                    /* tt */ \
        "v_cmp_ngt_f32_e32 vcc, %[eps], v48\n" \
        "v_cmp_ngt_f32_e64 s[66:67], v48, v14\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TQ_TRI_END_%=\n" \
        /* accepted (exec).  It replaces the best hit when it is nearer, or exactly as near with a larger triangle index; a \
           distance that is not NaN and nearer than the interval's end becomes the interval's end */ \
        "v_cmp_gt_i32_e64 s[68:69], 0, v20\n" \
        "v_cmp_lt_f32_e32 vcc, v48, v21\n" \
        "s_or_b64 s[68:69], s[68:69], vcc\n" \
        "v_cmp_eq_f32_e32 vcc, v48, v21\n" \
        "v_cmp_gt_i32_e64 s[72:73], v53, v20\n" \
        "s_and_b64 vcc, vcc, s[72:73]\n" \
        "s_or_b64 s[68:69], s[68:69], vcc\n" \
        "v_cmp_lt_f32_e32 vcc, v48, v14\n" \
        "s_mov_b64 s[72:73], exec\n" \
        "v_cndmask_b32_e32 v14, v14, v48, vcc\n" \
        "v_and_b32_e32 v50, 0x80000000, v11\n" \
        "s_and_b64 exec, exec, s[68:69]\n" \
        "v_mov_b32_e32 v20, v53\n" \
        "v_mov_b32_e32 v21, v48\n" \
        "v_mov_b32_e32 v22, v51\n" \
        "v_mov_b32_e32 v23, v47\n" \
        "s_mov_b64 exec, s[72:73]\n" \
        "v_cmp_ne_u32_e32 vcc, 0, v50\n"                   /* IntersectP: the first accepted triangle ends the ray */ \
        "v_cndmask_b32_e64 v12, v12, -1, vcc\n" \
        "v_cndmask_b32_e64 v13, v13, 0, vcc\n" \
        "TQ_TRI_END_%=:\n" \
        "s_mov_b64 exec, s[60:61]\n" \
        /* a ray that goes on and whose leaf has another triangle tests it in the same trip (its record came with the first one's): \
           the order of the tests and the interval they see are those of two trips */ \
        "v_cmp_ne_u32_e32 vcc, -1, v12\n" \
        "v_bfe_u32 v50, v12, 27, 4\n" \
        "v_cmp_lt_u32_e64 s[66:67], 0, v50\n" \
        "s_nop 0\n" \
        "s_and_b64 s[84:85], s[66:67], vcc\n"             /* second test */ \
        "s_andn2_b64 s[68:69], vcc, s[66:67]\n"           /* the leaf is exhausted: pop */ \
        "s_or_b64 s[78:79], s[78:79], s[68:69]\n" \
        "s_mov_b64 exec, s[84:85]\n" \
        "s_cbranch_execz TQ_POP_%=\n" \
        "v_add_u32_e32 v53, 1, v53\n" \
        "v_mul_f32_e32 v33, v5, v44\n" \
        "v_mul_f32_e32 v50, v6, v43\n" \
        "v_sub_f32_e32 v33, v33, v50\n" \
        "v_mul_f32_e32 v34, v6, v42\n" \
        "v_mul_f32_e32 v50, v4, v44\n" \
        "v_sub_f32_e32 v34, v34, v50\n" \
        "v_mul_f32_e32 v35, v4, v43\n" \
        "v_mul_f32_e32 v50, v5, v42\n" \
        "v_sub_f32_e32 v35, v35, v50\n" \
        "v_mul_f32_e32 v45, v33, v39\n" \
        "v_mul_f32_e32 v50, v34, v40\n" \
        "v_add_f32_e32 v45, v45, v50\n" \
        "v_mul_f32_e32 v50, v35, v41\n" \
        "v_add_f32_e32 v45, v45, v50\n" \
        "v_rcp_f32_e32 v47, v45\n" \
        "v_sub_f32_e32 v36, v0, v36\n" \
        "v_sub_f32_e32 v37, v1, v37\n" \
        "v_sub_f32_e32 v38, v2, v38\n" \
        "v_cmp_nle_f32_e64 s[66:67], abs(v45), s77\n" \
        "v_fma_f32 v49, -v45, v47, 1.0\n" \
        "v_fma_f32 v46, v49, v47, v47\n" \
        "s_cmp_lg_u64 s[66:67], 0\n" \
        "s_cbranch_scc1 TQ_DIV_IEEE2_%=\n" \
        "TQ_DIV_DONE2_%=:\n" \
        "v_mul_f32_e32 v51, v36, v33\n" \
        "v_mul_f32_e32 v50, v37, v34\n" \
        "v_add_f32_e32 v51, v51, v50\n" \
        "v_mul_f32_e32 v50, v38, v35\n" \
        "v_add_f32_e32 v51, v51, v50\n" \
        "v_mul_f32_e32 v33, v37, v41\n" \
        "v_mul_f32_e32 v50, v38, v40\n" \
        "v_sub_f32_e32 v33, v33, v50\n" \
        "v_mul_f32_e32 v34, v38, v39\n" \
        "v_mul_f32_e32 v50, v36, v41\n" \
        "v_sub_f32_e32 v34, v34, v50\n" \
        "v_mul_f32_e32 v35, v36, v40\n" \
        "v_mul_f32_e32 v50, v37, v39\n" \
        "v_sub_f32_e32 v35, v35, v50\n" \
        "v_mul_f32_e32 v51, v51, v46\n"                    /* b1 */ \
        "v_mul_f32_e32 v47, v4, v33\n" \
        "v_mul_f32_e32 v50, v5, v34\n" \
        "v_add_f32_e32 v47, v47, v50\n" \
        "v_mul_f32_e32 v50, v6, v35\n" \
        "v_add_f32_e32 v47, v47, v50\n" \
        "v_mul_f32_e32 v47, v47, v46\n"                    /* b2 */ \
        "v_cmp_nlt_f32_e64 s[66:67], abs(v45), s76\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v51\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v51\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_ngt_f32_e32 vcc, 0, v47\n" \
        "v_add_f32_e32 v50, v51, v47\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "v_cmp_nlt_f32_e32 vcc, 1.0, v50\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TQ_TRI2_END_%=\n" \
        "v_mul_f32_e32 v48, v42, v33\n" \
        "v_mul_f32_e32 v50, v43, v34\n" \
        "v_add_f32_e32 v48, v48, v50\n" \
        "v_mul_f32_e32 v50, v44, v35\n" \
        "v_add_f32_e32 v48, v48, v50\n" \
        "v_mul_f32_e32 v48, v48, v46\n"                    /* tt */ \
        "v_cmp_ngt_f32_e32 vcc, %[eps], v48\n" \
        "v_cmp_ngt_f32_e64 s[66:67], v48, v14\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_and_b64 exec, exec, s[66:67]\n" \
        "s_cbranch_scc0 TQ_TRI2_END_%=\n" \
        /* accepted (exec).  It replaces the best hit when it is nearer, or exactly as near with a larger triangle index; a \
           distance that is not NaN and nearer than the interval's end becomes the interval's end */ \
        "v_cmp_gt_i32_e64 s[68:69], 0, v20\n" \
        "v_cmp_lt_f32_e32 vcc, v48, v21\n" \
        "s_or_b64 s[68:69], s[68:69], vcc\n" \
        "v_cmp_eq_f32_e32 vcc, v48, v21\n" \
        "v_cmp_gt_i32_e64 s[72:73], v53, v20\n" \
        "s_and_b64 vcc, vcc, s[72:73]\n" \
        "s_or_b64 s[68:69], s[68:69], vcc\n" \
        "v_cmp_lt_f32_e32 vcc, v48, v14\n" \
        "s_mov_b64 s[72:73], exec\n" \
        "v_cndmask_b32_e32 v14, v14, v48, vcc\n" \
        "v_and_b32_e32 v50, 0x80000000, v11\n" \
        "s_and_b64 exec, exec, s[68:69]\n" \
        "v_mov_b32_e32 v20, v53\n" \
        "v_mov_b32_e32 v21, v48\n" \
        "v_mov_b32_e32 v22, v51\n" \
        "v_mov_b32_e32 v23, v47\n" \
        "s_mov_b64 exec, s[72:73]\n" \
        "v_cmp_ne_u32_e32 vcc, 0, v50\n"                   /* IntersectP: the first accepted triangle ends the ray */ \
        "v_cndmask_b32_e64 v12, v12, -1, vcc\n" \
        "v_cndmask_b32_e64 v13, v13, 0, vcc\n" \
        "TQ_TRI2_END_%=:\n" \
        "s_mov_b64 exec, s[84:85]\n" \
        "v_cmp_ne_u32_e32 vcc, -1, v12\n" \
        "v_bfe_u32 v50, v12, 27, 4\n" \
        "v_add_u32_e32 v49, 0xf0000002, v12\n"             /* first + 2, two triangles fewer */ \
        "v_cmp_lt_u32_e64 s[66:67], 1, v50\n" \
        "s_nop 0\n" \
        "s_and_b64 s[66:67], s[66:67], vcc\n" \
        "s_andn2_b64 s[68:69], vcc, s[66:67]\n" \
        "v_cndmask_b32_e64 v12, v12, v49, s[66:67]\n" \
        "s_or_b64 s[78:79], s[78:79], s[68:69]\n" \
        /* ---------------------------------------------------------------- pop (s[78:79]) */ \
        "TQ_POP_%=:\n" \
        "s_mov_b64 exec, s[78:79]\n" \
        "s_cbranch_execz TQ_POP_NONE_%=\n" \
        "v_cmp_lt_i32_e32 vcc, 0, v13\n" \
        "v_mov_b32_e32 v12, -1\n" \
        "s_and_b64 exec, exec, vcc\n" \
        "s_cbranch_execz TQ_POP_NONE_%=\n" \
        "v_add_u32_e32 v13, -1, v13\n" \
        "v_cmp_gt_i32_e32 vcc, %[depth], v13\n" \
        "s_mov_b64 s[72:73], exec\n" \
        "s_and_b64 exec, exec, vcc\n" \
        PT_WF_LVL("v33", "v13", "v18") \
        "ds_read_b32 v12, v33 offset:" PT_WF_O3 "\n" \
        "s_andn2_b64 exec, s[72:73], vcc\n" \
        "s_cbranch_execz TQ_POP_LDS_%=\n" \
        PT_WF_LVL("v33", "v13", "v16") \
        "global_load_dword v12, v33, %[spill]\n" \
        "s_waitcnt vmcnt(0)\n" \
        "TQ_POP_LDS_%=:\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        "TQ_POP_NONE_%=:\n" \
        "s_mov_b64 exec, -1\n"

#define PT_WF_ASM_SLOW \
        "TQ_PUSH_SLOW_%=:\n" \
        "v_add_u32_e32 v29, -3, v28\n" \
        "v_cmp_gt_i32_e32 vcc, %[depth], v29\n" \
        "s_and_b64 exec, s[72:73], vcc\n" \
        PT_WF_LVL("v30", "v29", "v18") \
        "ds_write_b32 v30, v51 offset:" PT_WF_O3 "\n" \
        "s_andn2_b64 exec, s[72:73], vcc\n" \
        PT_WF_LVL("v30", "v29", "v16") \
        "global_store_dword v30, v51, %[spill]\n" \
        "s_mov_b64 exec, s[62:63]\n" \
        "v_add_u32_e32 v29, -2, v28\n" \
        "v_cmp_gt_i32_e32 vcc, %[depth], v29\n" \
        "s_and_b64 exec, s[68:69], vcc\n" \
        PT_WF_LVL("v30", "v29", "v18") \
        "ds_write_b32 v30, v48 offset:" PT_WF_O3 "\n" \
        "s_andn2_b64 exec, s[68:69], vcc\n" \
        PT_WF_LVL("v30", "v29", "v16") \
        "global_store_dword v30, v48, %[spill]\n" \
        "s_mov_b64 exec, s[62:63]\n" \
        "v_add_u32_e32 v29, -1, v28\n" \
        "v_cmp_gt_i32_e32 vcc, %[depth], v29\n" \
        "s_and_b64 exec, s[66:67], vcc\n" \
        PT_WF_LVL("v30", "v29", "v18") \
        "ds_write_b32 v30, v49 offset:" PT_WF_O3 "\n" \
        "s_andn2_b64 exec, s[66:67], vcc\n" \
        PT_WF_LVL("v30", "v29", "v16") \
        "global_store_dword v30, v49, %[spill]\n" \
        "s_waitcnt vmcnt(0)\n" \
        "s_branch TQ_PUSHED_%=\n" \
        /* ---------------------------------------------------------------- IEEE reciprocal for divisors outside the Newton range */ \
        "TQ_DIV_IEEE_%=:\n" \
        "v_div_scale_f32 v46, s[66:67], v45, v45, 1.0\n" \
        "v_div_scale_f32 v48, vcc, 1.0, v45, 1.0\n" \
        "v_rcp_f32_e32 v47, v46\n" \
        "s_nop 0\n" \
        "v_fma_f32 v49, -v46, v47, 1.0\n" \
        "v_fmac_f32_e32 v47, v49, v47\n" \
        "v_mul_f32_e32 v52, v48, v47\n" \
        "v_fma_f32 v49, -v46, v52, v48\n" \
        "v_fmac_f32_e32 v52, v49, v47\n" \
        "v_fma_f32 v46, -v46, v52, v48\n" \
        "v_div_fmas_f32 v46, v46, v47, v52\n" \
        "v_div_fixup_f32 v46, v46, v45, 1.0\n" \
        "s_branch TQ_DIV_DONE_%=\n" \
        "TQ_DIV_IEEE2_%=:\n" \
        "v_div_scale_f32 v46, s[66:67], v45, v45, 1.0\n" \
        "v_div_scale_f32 v48, vcc, 1.0, v45, 1.0\n" \
        "v_rcp_f32_e32 v47, v46\n" \
        "s_nop 0\n" \
        "v_fma_f32 v49, -v46, v47, 1.0\n" \
        "v_fmac_f32_e32 v47, v49, v47\n" \
        "v_mul_f32_e32 v52, v48, v47\n" \
        "v_fma_f32 v49, -v46, v52, v48\n" \
        "v_fmac_f32_e32 v52, v49, v47\n" \
        "v_fma_f32 v46, -v46, v52, v48\n" \
        "v_div_fmas_f32 v46, v46, v47, v52\n" \
        "v_div_fixup_f32 v46, v46, v45, 1.0\n" \
        "s_branch TQ_DIV_DONE2_%=\n"

// ---- the wide walk of the trace stage, hand-scheduled ----------------------------------------------------------------------
// The node block, the triangle block, the sorting network, the pushes and pops are those of trace_pool_wide_asm (pt_kernel.hip:
// the same instructions on the same registers, hence the same bits as the C++ walk above, which stays the specification and can
// be selected with -DPT_WF_WIDE_ASM=0); what differs is where rays come from and where results go:
//   * refill without a stall.  A free lane is handed the next ray index of the wave's current segment and issues the load of
//     its ray id; one trip later (every trip waits for all outstanding loads anyway) it turns the id into the path's offsets and
//     issues the gathers of direction and origin; another trip later it forms 1 / direction (the IEEE quotient: bbox.h:79
//     divides) and starts at the root.  Two trips without work per ray, nobody else waits.
//   * a finished ray stores {primitive, t, b1, b2} straight to hit[kind][path] (one global_store_dwordx4, never waited for).
//   * segments: a claim (one returning atomic on the head of an XCD's queue + one scalar load of the group's four counts)
//     yields up to four segments that are walked one after the other without further memory operations.
//   * the stack: PT_WF_STACK_LEVELS levels per lane in LDS (nothing else is in LDS), deeper levels in the wave's spill slice.
// Registers: v[0:2] origin (v3: its unused fourth word)  v[4:6] direction  v7 tmax as loaded  v[8:10] 1 / direction  v11 ray id
//   v12 current entry (-1: finished)  v13 stack size  v14 end of the interval  v15 byte offset of the ray's result (-1: no ray)
//   v16 spill column  v18 LDS stack column  v19 result offset of a ray being loaded  v[20:23] best hit  v[24:54] as in pt_kernel.hip
//   s[64:65] lanes with a ray  s[86:87] lanes whose id is in flight  s[88:89] lanes whose ray record is in flight
//   s70 cursor in the segment  s90 its ray count  s91 its byte offset in rayq  s94 no segment is left  s95 trips made
#ifndef PT_WF_WIDE_ASM
#define PT_WF_WIDE_ASM 1
#endif
#ifndef PT_WF_LEAF_MIN
#define PT_WF_LEAF_MIN 8
#endif
#ifndef PT_WF_HIT_STORE             // (experiment hook: cache policy of the result store)
#define PT_WF_HIT_STORE "global_store_dwordx4 v15, v[20:23], %[hit]\n"
#endif
#ifndef PT_WF_MIN_TRIPS
#define PT_WF_MIN_TRIPS 32           // ... but not before the wave has made this many trips in the round
#endif
#ifndef PT_WF_STOP_T
#define PT_WF_STOP_T 8               // no group left and at most this many lanes still walk: the wave parks their rays for the next round
#endif
#define PT_WF_LVL(dst, lvl, col) "v_lshl_add_u32 " dst ", " lvl ", 8, " col "\n"      /* one lane per ray: a level is 64 words */
#define PT_WF_O1 "256"
#define PT_WF_O2 "512"
#define PT_WF_O3 "768"
__device__ __forceinline__ void wf_trace_wide_asm(const DevParams &P, const WfParams &W, unsigned stack_lds, unsigned lane, unsigned shared_lds, uint32_t chunk0,
                                                  uint32_t round)
{
    const unsigned s_eps = __builtin_amdgcn_readfirstlane(__float_as_uint(P.eps));
    const unsigned long long s_nodes = uniform64((unsigned long long)P.wide);
    const unsigned s_trioff = __builtin_amdgcn_readfirstlane(P.wide_tris_off);
    // level l of this lane's stack beyond the LDS levels: spill base + column + 256 l (the base is moved back by the LDS levels)
    const unsigned long long s_spill = uniform64((unsigned long long)(W.spill - 64 * kWfStackLevels));
    const unsigned long long s_rayq = uniform64((unsigned long long)W.rayq), s_ray = uniform64((unsigned long long)W.ray),
                             s_org = uniform64((unsigned long long)W.org), s_hit = uniform64((unsigned long long)W.hit);
    const unsigned s_np = __builtin_amdgcn_readfirstlane(W.n_paths);
    const unsigned s_shared = __builtin_amdgcn_readfirstlane(shared_lds);                   // LDS address of the workgroup's WfShared
    const unsigned s_segbase = __builtin_amdgcn_readfirstlane(chunk0 * (unsigned)(kWfSegRays * 4));     // byte offset of the workgroup's first segment in rayq
    const unsigned s_stack = __builtin_amdgcn_readfirstlane(stack_lds) - 768u;      // (the pushes address level size' - 3 .. size' - 1 from one base)
    const unsigned v_spill = ((blockIdx.x * (unsigned)kWfWgWaves + (threadIdx.x >> 6)) * 64u * W.spill_levels + lane) * 4u;
    const unsigned long long s_save = uniform64((unsigned long long)W.save);
    const unsigned long long s_counters = uniform64((unsigned long long)P.counters);         // (probe builds)
    const unsigned v_save = ((blockIdx.x * (unsigned)kWfWgWaves + (threadIdx.x >> 6)) * 64u + lane) * (unsigned)(kWfSaveDwords * 4);
    const unsigned s_tag = __builtin_amdgcn_readfirstlane(round + 1u);                       // records parked FOR this round carry it
    static_assert(kWfStackLevels % 4 == 0 && kWfStackLevels >= 8 && kWfStackLevels <= 28 && 8 + kWfStackLevels <= kWfSaveDwords, "the save record holds the LDS levels");
    asm volatile(
        "s_mov_b32 s76, 0x322bcc77\n"
        "s_mov_b32 s77, 0x71800000\n"
        "s_mov_b64 s[64:65], 0\n"
        "s_mov_b64 s[82:83], 0\n"
        "s_mov_b64 s[86:87], 0\n"
        "s_mov_b64 s[88:89], 0\n"
        "s_mov_b32 s70, 0\n"
        "s_mov_b32 s90, 0\n"
        "s_mov_b32 s91, 0\n"
        "s_mov_b32 s94, 0\n"
        "s_mov_b32 s95, 0\n"
#if PT_WF_PROBE == 3
        "s_mov_b32 s96, 0\n"
        "s_mov_b32 s97, 0\n"
        "s_mov_b32 s98, 0\n"
#endif
        "v_mbcnt_lo_u32_b32 v33, -1, 0\n"
        "v_mbcnt_hi_u32_b32 v33, -1, v33\n"                /* lane */
        "v_lshl_add_u32 v18, v33, 2, %[stack]\n"
        "v_mov_b32_e32 v16, %[vspill]\n"
        "v_mov_b32_e32 v15, -1\n"
        "v_mov_b32_e32 v12, -1\n"
        "v_mov_b32_e32 v13, 0\n"
        /* ---------------------------------------------------------------- rays parked by the last round's trace stage come back to their lanes */
        "v_mov_b32_e32 v17, %[vsave]\n"
        "global_load_dwordx4 v[24:27], v17, %[save]\n"
        "global_load_dwordx4 v[28:31], v17, %[save] offset:16\n"
        "s_waitcnt vmcnt(0)\n"
        "v_lshrrev_b32_e32 v33, 8, v26\n"
        "v_cmp_eq_u32_e64 s[64:65], %[tag], v33\n"
        "s_mov_b64 exec, s[64:65]\n"
        "s_cbranch_execz TQ_NORESUME_%=\n"
        "v_mov_b32_e32 v11, v24\n"
        "v_mov_b32_e32 v12, v25\n"
        "v_and_b32_e32 v13, 0xff, v26\n"
        "v_mov_b32_e32 v14, v27\n"
        "v_mov_b32_e32 v20, v28\n"
        "v_mov_b32_e32 v21, v29\n"
        "v_mov_b32_e32 v22, v30\n"
        "v_mov_b32_e32 v23, v31\n"
        "v_bfe_u32 v33, v11, 28, 2\n"
        "v_and_b32_e32 v34, 0xfffffff, v11\n"
        "v_mul_lo_u32 v33, v33, %[np]\n"
        "v_add_lshl_u32 v15, v33, v34, 4\n"
        "v_lshlrev_b32_e32 v19, 4, v34\n"
        "global_load_dwordx4 v[24:27], v17, %[save] offset:32\n"
        "global_load_dwordx4 v[28:31], v17, %[save] offset:48\n"
#if PT_WF_STACK_LEVELS > 8
        "global_load_dwordx4 v[32:35], v17, %[save] offset:64\n"
#endif
#if PT_WF_STACK_LEVELS > 12
        "global_load_dwordx4 v[36:39], v17, %[save] offset:80\n"
#endif
#if PT_WF_STACK_LEVELS > 16
        "global_load_dwordx4 v[40:43], v17, %[save] offset:96\n"
#endif
#if PT_WF_STACK_LEVELS > 20
        "global_load_dwordx4 v[44:47], v17, %[save] offset:112\n"
#endif
#if PT_WF_STACK_LEVELS > 24
        "global_load_dwordx4 v[48:51], v17, %[save] offset:128\n"
#endif
        "global_load_dwordx4 v[4:7], v15, %[ray]\n"
        "global_load_dwordx4 v[0:3], v19, %[org]\n"
        "s_waitcnt vmcnt(0)\n"
        "ds_write_b32 v18, v24 offset:768\n"
        "ds_write_b32 v18, v25 offset:1024\n"
        "ds_write_b32 v18, v26 offset:1280\n"
        "ds_write_b32 v18, v27 offset:1536\n"
        "ds_write_b32 v18, v28 offset:1792\n"
        "ds_write_b32 v18, v29 offset:2048\n"
        "ds_write_b32 v18, v30 offset:2304\n"
        "ds_write_b32 v18, v31 offset:2560\n"
#if PT_WF_STACK_LEVELS > 8
        "ds_write_b32 v18, v32 offset:2816\n"
        "ds_write_b32 v18, v33 offset:3072\n"
        "ds_write_b32 v18, v34 offset:3328\n"
        "ds_write_b32 v18, v35 offset:3584\n"
#endif
#if PT_WF_STACK_LEVELS > 12
        "ds_write_b32 v18, v36 offset:3840\n"
        "ds_write_b32 v18, v37 offset:4096\n"
        "ds_write_b32 v18, v38 offset:4352\n"
        "ds_write_b32 v18, v39 offset:4608\n"
#endif
#if PT_WF_STACK_LEVELS > 16
        "ds_write_b32 v18, v40 offset:4864\n"
        "ds_write_b32 v18, v41 offset:5120\n"
        "ds_write_b32 v18, v42 offset:5376\n"
        "ds_write_b32 v18, v43 offset:5632\n"
#endif
#if PT_WF_STACK_LEVELS > 20
        "ds_write_b32 v18, v44 offset:5888\n"
        "ds_write_b32 v18, v45 offset:6144\n"
        "ds_write_b32 v18, v46 offset:6400\n"
        "ds_write_b32 v18, v47 offset:6656\n"
#endif
#if PT_WF_STACK_LEVELS > 24
        "ds_write_b32 v18, v48 offset:6912\n"
        "ds_write_b32 v18, v49 offset:7168\n"
        "ds_write_b32 v18, v50 offset:7424\n"
        "ds_write_b32 v18, v51 offset:7680\n"
#endif
        PT_WF_ASM_INV_DIR
        "s_waitcnt lgkmcnt(0)\n"
        "TQ_NORESUME_%=:\n"
        "s_mov_b64 exec, -1\n"
        /* ---------------------------------------------------------------- loop header */
        "TQ_LOOP_%=:\n"
        /* lanes whose ray record was fetched a trip ago: 1 / direction, start at the root */
        "s_cmp_eq_u64 s[88:89], 0\n"
        "s_cbranch_scc1 TQ_NOFINAL_%=\n"
        "s_waitcnt vmcnt(0)\n"
        "s_mov_b64 exec, s[88:89]\n"
        PT_WF_ASM_INV_DIR
        "v_mov_b32_e32 v14, v7\n"
        "v_mov_b32_e32 v12, 0\n"
        "v_mov_b32_e32 v13, 0\n"
        "v_mov_b32_e32 v20, -1\n"
        "v_mov_b32_e32 v21, 0\n"
        "v_mov_b32_e32 v22, 0\n"
        "v_mov_b32_e32 v23, 0\n"
        "v_mov_b32_e32 v15, v19\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b64 s[88:89], 0\n"
        "TQ_NOFINAL_%=:\n"
        /* lanes whose ray id was fetched a trip ago: the gathers of direction (kind * paths + path) and origin (path) */
        "s_cmp_eq_u64 s[86:87], 0\n"
        "s_cbranch_scc1 TQ_NOGATHER_%=\n"
        "s_waitcnt vmcnt(0)\n"
        "s_mov_b64 exec, s[86:87]\n"
        "v_bfe_u32 v33, v11, 28, 2\n"
        "v_and_b32_e32 v34, 0xfffffff, v11\n"
        "v_mul_lo_u32 v33, v33, %[np]\n"
        "v_add_lshl_u32 v19, v33, v34, 4\n"
        "v_lshlrev_b32_e32 v34, 4, v34\n"
        "global_load_dwordx4 v[4:7], v19, %[ray]\n"
        "global_load_dwordx4 v[0:3], v34, %[org]\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b64 s[88:89], s[86:87]\n"
        "s_mov_b64 s[86:87], 0\n"
        "TQ_NOGATHER_%=:\n"
        "v_cmp_lt_i32_e64 s[64:65], -1, v15\n"             /* lanes with a ray */
        "v_cmp_eq_u32_e64 s[66:67], -1, v12\n"             /* ... that is finished */
        "s_and_b64 s[68:69], s[64:65], s[66:67]\n"
        "s_cbranch_scc0 TQ_FILL_%=\n"
        /* ---------------------------------------------------------------- finished rays (s[68:69]): a miss reports the end of the interval */
        "s_mov_b64 exec, s[68:69]\n"
        "v_cmp_gt_i32_e32 vcc, 0, v20\n"
        "v_cndmask_b32_e32 v21, v21, v14, vcc\n"
        PT_WF_HIT_STORE
        "v_mov_b32_e32 v15, -1\n"
        "s_andn2_b64 s[64:65], s[64:65], s[68:69]\n"
        "s_mov_b64 exec, -1\n"
        /* ---------------------------------------------------------------- refill: free lanes take the next rays of the segment */
        "TQ_FILL_%=:\n"
        "s_cmp_lg_u32 s94, 0\n"
        "s_cbranch_scc1 TQ_TRIPCHK_%=\n"
        "s_or_b64 s[66:67], s[64:65], s[88:89]\n"          /* lanes that are taken */
        "s_bcnt1_i32_b64 s71, s[66:67]\n"
        "s_cmp_gt_u32 s71, %[maxbusy]\n"
        "s_cbranch_scc1 TQ_TRIPCHK_%=\n"
        "s_cmp_lt_u32 s70, s90\n"
        "s_cbranch_scc1 TQ_ASSIGN_%=\n"
        /* the workgroup's next segment: one LDS atomic (nobody outside the workgroup touches the counter), then its ray count */
        "TQ_NEXTSEG_%=:\n"
        "s_mov_b64 exec, 1\n"
        "v_mov_b32_e32 v33, %[shared]\n"
        "v_mov_b32_e32 v34, 1\n"
        "ds_add_rtn_u32 v35, v33, v34 offset:%[o_next]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 s73, v35\n"
        "s_cmp_lt_u32 s73, %[nchunks]\n"
        "s_cbranch_scc0 TQ_EXHAUSTED_%=\n"
        "v_lshl_add_u32 v33, s73, 2, v33\n"
        "ds_read_b32 v35, v33\n"                          /* seg_count[c] leads the record */
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 s90, v35\n"
        "s_mov_b64 exec, -1\n"
        "s_mul_i32 s91, s73, 768\n"
        "s_add_u32 s91, s91, %[segbase]\n"
        "s_mov_b32 s70, 0\n"
        "s_cmp_lg_u32 s90, 0\n"
        "s_cbranch_scc1 TQ_ASSIGN_%=\n"
        "s_branch TQ_NEXTSEG_%=\n"
        "TQ_EXHAUSTED_%=:\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b32 s94, 1\n"
        "s_branch TQ_TRIPCHK_%=\n"
        "TQ_ASSIGN_%=:\n"
        "s_not_b64 s[66:67], s[66:67]\n"                   /* free lanes */
        "v_mbcnt_lo_u32_b32 v33, s66, 0\n"
        "v_mbcnt_hi_u32_b32 v33, s67, v33\n"
        "v_add_u32_e32 v33, s70, v33\n"                    /* index in the segment */
        "v_cmp_gt_u32_e32 vcc, s90, v33\n"
        "s_and_b64 s[86:87], vcc, s[66:67]\n"
        "s_bcnt1_i32_b64 s71, s[66:67]\n"
        "s_add_u32 s70, s70, s71\n"
        "s_mov_b64 exec, s[86:87]\n"
        "v_lshl_add_u32 v33, v33, 2, s91\n"
        "global_load_dword v11, v33, %[rayq]\n"
        "s_mov_b64 exec, -1\n"
        "TQ_TRIPCHK_%=:\n"
        "s_cmp_lg_u64 s[64:65], 0\n"
        "s_cbranch_scc0 TQ_NOBUSY_%=\n"
        /* no group is left, nothing is being loaded and only a few lanes still walk: the wave parks those rays for the next round */
        "s_cmp_eq_u32 s94, 0\n"
        "s_cbranch_scc1 TQ_TRIP_%=\n"
        "s_or_b64 s[66:67], s[86:87], s[88:89]\n"
        "s_cmp_lg_u64 s[66:67], 0\n"
        "s_cbranch_scc1 TQ_TRIP_%=\n"
        "s_bcnt1_i32_b64 s71, s[64:65]\n"
        "s_cmp_gt_u32 s71, %[tstop]\n"
        "s_cbranch_scc1 TQ_TRIP_%=\n"
        "s_cmp_lt_u32 s95, %[mintrips]\n"                /* ... but every round moves its rays on by some trips (a wave that resumes parked rays would park them again at once) */
        "s_cbranch_scc1 TQ_TRIP_%=\n"
        "s_mov_b64 exec, s[64:65]\n"
        "v_mov_b32_e32 v33, -2\n"
        "global_store_dword v15, v33, %[hit]\n"          /* the result slot says "not yet" */
        "v_mov_b32_e32 v4, v11\n"                         /* (the ray's own registers are free now) */
        "v_mov_b32_e32 v5, v12\n"
        "v_mov_b32_e32 v6, %[tag]\n"
        "v_add_u32_e32 v6, 1, v6\n"
        "v_lshl_or_b32 v6, v6, 8, v13\n"
        "v_mov_b32_e32 v7, v14\n"
        "global_store_dwordx4 v17, v[4:7], %[save]\n"
        "global_store_dwordx4 v17, v[20:23], %[save] offset:16\n"
        "ds_read_b32 v24, v18 offset:768\n"
        "ds_read_b32 v25, v18 offset:1024\n"
        "ds_read_b32 v26, v18 offset:1280\n"
        "ds_read_b32 v27, v18 offset:1536\n"
        "ds_read_b32 v28, v18 offset:1792\n"
        "ds_read_b32 v29, v18 offset:2048\n"
        "ds_read_b32 v30, v18 offset:2304\n"
        "ds_read_b32 v31, v18 offset:2560\n"
#if PT_WF_STACK_LEVELS > 8
        "ds_read_b32 v32, v18 offset:2816\n"
        "ds_read_b32 v33, v18 offset:3072\n"
        "ds_read_b32 v34, v18 offset:3328\n"
        "ds_read_b32 v35, v18 offset:3584\n"
#endif
#if PT_WF_STACK_LEVELS > 12
        "ds_read_b32 v36, v18 offset:3840\n"
        "ds_read_b32 v37, v18 offset:4096\n"
        "ds_read_b32 v38, v18 offset:4352\n"
        "ds_read_b32 v39, v18 offset:4608\n"
#endif
#if PT_WF_STACK_LEVELS > 16
        "ds_read_b32 v40, v18 offset:4864\n"
        "ds_read_b32 v41, v18 offset:5120\n"
        "ds_read_b32 v42, v18 offset:5376\n"
        "ds_read_b32 v43, v18 offset:5632\n"
#endif
#if PT_WF_STACK_LEVELS > 20
        "ds_read_b32 v44, v18 offset:5888\n"
        "ds_read_b32 v45, v18 offset:6144\n"
        "ds_read_b32 v46, v18 offset:6400\n"
        "ds_read_b32 v47, v18 offset:6656\n"
#endif
#if PT_WF_STACK_LEVELS > 24
        "ds_read_b32 v48, v18 offset:6912\n"
        "ds_read_b32 v49, v18 offset:7168\n"
        "ds_read_b32 v50, v18 offset:7424\n"
        "ds_read_b32 v51, v18 offset:7680\n"
#endif
        "s_waitcnt lgkmcnt(0)\n"
        "global_store_dwordx4 v17, v[24:27], %[save] offset:32\n"
        "global_store_dwordx4 v17, v[28:31], %[save] offset:48\n"
#if PT_WF_STACK_LEVELS > 8
        "global_store_dwordx4 v17, v[32:35], %[save] offset:64\n"
#endif
#if PT_WF_STACK_LEVELS > 12
        "global_store_dwordx4 v17, v[36:39], %[save] offset:80\n"
#endif
#if PT_WF_STACK_LEVELS > 16
        "global_store_dwordx4 v17, v[40:43], %[save] offset:96\n"
#endif
#if PT_WF_STACK_LEVELS > 20
        "global_store_dwordx4 v17, v[44:47], %[save] offset:112\n"
#endif
#if PT_WF_STACK_LEVELS > 24
        "global_store_dwordx4 v17, v[48:51], %[save] offset:128\n"
#endif
        "s_mov_b64 exec, 1\n"
        "v_mov_b32_e32 v33, %[shared]\n"
        "v_mov_b32_e32 v34, 1\n"
        "ds_write_b32 v33, v34 offset:%[o_parked]\n"
        "s_branch TQ_DONE_%=\n"
        "TQ_NOBUSY_%=:\n"
        /* no lane has a ray: the loop goes on while loads are in flight or groups are left */
        "s_or_b64 s[66:67], s[86:87], s[88:89]\n"
        "s_cmp_lg_u64 s[66:67], 0\n"
        "s_cbranch_scc1 TQ_LOOP_%=\n"
        "s_cmp_eq_u32 s94, 0\n"
        "s_cbranch_scc1 TQ_LOOP_%=\n"
        "s_branch TQ_DONE_%=\n"
        /* ---------------------------------------------------------------- one trip (pt_kernel.hip, trace_pool_wide_asm) */
        "TQ_TRIP_%=:\n"
        "s_add_u32 s95, s95, 1\n"
#if PT_WF_PROBE == 3
        "s_bcnt1_i32_b64 s71, s[64:65]\n"
        "s_add_u32 s96, s96, s71\n"
#endif
        "v_cmp_gt_i32_e64 s[60:61], 0, v12\n"
        "s_and_b64 s[60:61], s[60:61], s[64:65]\n"         /* at a leaf (bit 31 set, not -1: busy lanes only) */
        "s_andn2_b64 s[62:63], s[64:65], s[60:61]\n"       /* at a wide node */
        /* too few lanes at a leaf: they wait (unless nobody is at a wide node); else too few at a wide node: those wait */
        "s_bcnt1_i32_b64 s71, s[60:61]\n"
        "s_bcnt1_i32_b64 s72, s[62:63]\n"
        "s_cmp_ge_u32 s71, %[leafmin]\n"
        "s_cbranch_scc1 TQ_VOTE_NODE_%=\n"
        "s_cmp_eq_u32 s72, 0\n"
        "s_cbranch_scc1 TQ_VOTED_%=\n"
        "s_mov_b64 s[60:61], 0\n"
        "s_branch TQ_VOTED_%=\n"
        "TQ_VOTE_NODE_%=:\n"
        "s_cmp_ge_u32 s72, %[nodemin]\n"
        "s_cbranch_scc1 TQ_VOTED_%=\n"
        "s_mov_b64 s[62:63], 0\n"
        "TQ_VOTED_%=:\n"
#if PT_WF_PROBE == 3
        "s_bcnt1_i32_b64 s71, s[62:63]\n"
        "s_add_u32 s97, s97, s71\n"
        "s_bcnt1_i32_b64 s71, s[60:61]\n"
        "s_add_u32 s98, s98, s71\n"
#endif
        /* ---- fetches of both kinds (a vector-memory instruction whose exec is empty is not counted by vmcnt: both blocks
           wait for everything) */
        /* one set of fetches serves both kinds: a lane's offset from the base of the wide nodes is its node's, or that of its
           triangle in the copy behind the nodes; a node lane gets its 112-byte record, a leaf lane its triangle in v[24:32] and the
           one after it in v[36:44] (the rest of what it reads is not used) */
        PT_WF_ASM_TRIP
        "s_branch TQ_LOOP_%=\n"
        /* ---------------------------------------------------------------- a push beyond the LDS levels (rare): level by level */
        PT_WF_ASM_SLOW
        "TQ_DONE_%=:\n"
#if PT_WF_PROBE == 3
        "s_mov_b64 exec, 1\n"
        "v_mov_b32_e32 v33, 64\n"
        "v_mov_b32_e32 v34, s95\n"
        "global_atomic_add v33, v34, %[counters]\n"
        "v_mov_b32_e32 v34, s96\n"
        "global_atomic_add v33, v34, %[counters] offset:8\n"
        "v_mov_b32_e32 v34, s97\n"
        "global_atomic_add v33, v34, %[counters] offset:16\n"
        "v_mov_b32_e32 v34, s98\n"
        "global_atomic_add v33, v34, %[counters] offset:24\n"
#endif
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_mov_b64 exec, -1\n"
        :
        : [eps] "s"(s_eps), [nodes] "s"(s_nodes), [trioff] "s"(s_trioff), [spill] "s"(s_spill), [stack] "s"(s_stack), [vspill] "v"(v_spill),
          [rayq] "s"(s_rayq), [ray] "s"(s_ray), [org] "s"(s_org), [hit] "s"(s_hit), [np] "s"(s_np), [shared] "s"(s_shared), [segbase] "s"(s_segbase),
          [save] "s"(s_save), [vsave] "v"(v_save), [tag] "s"(s_tag), [counters] "s"(s_counters),
          [depth] "n"(kWfStackLevels), [maxbusy] "n"(64 - PT_WF_FETCH_T), [leafmin] "n"(PT_WF_LEAF_MIN), [nodemin] "n"(1),
          [tstop] "n"(PT_WF_STOP_T), [mintrips] "n"(PT_WF_MIN_TRIPS), [nchunks] "n"(kWfWgChunks),
          [o_next] "n"(offsetof(WfShared, trace_next)), [o_parked] "n"(offsetof(WfShared, parked))
        : "memory", "vcc", "scc", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
          "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91",
          "s94", "s95", "s96", "s97", "s98",
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37",
          "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54");
}

#undef PT_WF_LVL
#undef PT_WF_O1
#undef PT_WF_O2
#undef PT_WF_O3

// ---- the ray stream, hand-scheduled ------------------------------------------------------------------------------------------
// wf_trace_stream above is the specification of the scheduling (and, with -DPT_WF_STREAM_CXX=1, runs); this is the same loop around the SAME
// node block, triangle block, sorting network, pushes and pops as wf_trace_wide_asm (PT_WF_ASM_TRIP: the batch's lanes are s[62:63]
// for a batch at wide nodes, s[60:61] for a batch at leaves).  A wave's block in LDS, R = kWfR slots, one row of R words per field:
//   rows 0-16  ox oy oz b1 | dx dy dz b2 | 1/dx 1/dy 1/dz id | cur sp tmax bprim | bt        rows 17 .. 17 + kWfS - 1  the stack's first levels
//   then three byte lists of R entries - slots whose next step is a wide node / a leaf / free slots - and their three counts.
// Registers beyond those of the trip: v55 lane, v56 address of the slot's column (block + 4 slot), v57 slot; a group of rays moving in:
// v58 id, v59 slot, v[60:63] direction and interval end, v[64:67] origin, s[86:87] ids on their way, s[88:89] records on their way (one
// group at a time, one stage per trip: id -> record -> 1 / direction and the slot's rows).  s96 / s97 / s98 entries of the three lists,
// s70 s90 s91 s94 s95 as in wf_trace_wide_asm, s71 s92 s99 temporaries.
constexpr int kWfsRowBytes = 4 * kWfR;
constexpr int kWfsRows = 17 + kWfS;
constexpr int kWfsLists = kWfsRows * kWfsRowBytes;              // byte offset of the three lists
constexpr int kWfsCounts = kWfsLists + 3 * kWfR;                // ... of the three counts
constexpr int kWfsBlockBytes = (kWfsCounts + 12 + 15) & ~15;
static_assert(kWfR % 4 == 0 && kWfR <= 255 && kWfsBlockBytes < 65536, "slots are bytes, offsets are 16-bit immediates");

#define PT_WF_LVL(dst, lvl, col) "v_mad_u32_u24 " dst ", " lvl ", %[lstride], " col "\n"      /* a level is R words */
#define PT_WF_O1 "%[o1]"
#define PT_WF_O2 "%[o2]"
#define PT_WF_O3 "%[o3]"
__device__ __forceinline__ void wf_trace_stream_asm(const DevParams &P, const WfParams &W, unsigned block_lds, unsigned shared_lds, uint32_t chunk0)
{
    const unsigned s_eps = __builtin_amdgcn_readfirstlane(__float_as_uint(P.eps));
    const unsigned long long s_nodes = uniform64((unsigned long long)P.wide);
    const unsigned s_trioff = __builtin_amdgcn_readfirstlane(P.wide_tris_off);
    // level l >= kWfS of slot s: spill + 4 (R l + s), with the base of the wave's slice moved back by the LDS levels
    const unsigned long long s_spill = uniform64((unsigned long long)(W.spill + (size_t)(blockIdx.x * (unsigned)kWfWgWaves + (threadIdx.x >> 6)) * (uint32_t)kWfR * W.spill_levels)
                                                 - (unsigned long long)(4 * kWfR * kWfS));
    const unsigned long long s_rayq = uniform64((unsigned long long)W.rayq), s_ray = uniform64((unsigned long long)W.ray),
                             s_org = uniform64((unsigned long long)W.org), s_hit = uniform64((unsigned long long)W.hit);
    const unsigned s_np = __builtin_amdgcn_readfirstlane(W.n_paths);
    const unsigned s_shared = __builtin_amdgcn_readfirstlane(shared_lds);
    const unsigned s_segbase = __builtin_amdgcn_readfirstlane(chunk0 * (unsigned)(kWfSegRays * 4));
    const unsigned s_blk = __builtin_amdgcn_readfirstlane(block_lds);
    const unsigned s_lstride = __builtin_amdgcn_readfirstlane((unsigned)kWfsRowBytes);
    asm volatile(
        "s_mov_b32 s76, 0x322bcc77\n"
        "s_mov_b32 s77, 0x71800000\n"
        "s_mov_b64 s[86:87], 0\n"
        "s_mov_b64 s[88:89], 0\n"
        "s_mov_b32 s70, 0\n"
        "s_mov_b32 s90, 0\n"
        "s_mov_b32 s91, 0\n"
        "s_mov_b32 s94, 0\n"
        "s_mov_b32 s95, 0\n"
        "v_mbcnt_lo_u32_b32 v55, -1, 0\n"
        "v_mbcnt_hi_u32_b32 v55, -1, v55\n"                /* lane */
        "v_mov_b32_e32 v33, %[blk]\n"
        "ds_read_b32 v34, v33 offset:%[o_cnt0]\n"
        "ds_read_b32 v35, v33 offset:%[o_cnt1]\n"
        "ds_read_b32 v36, v33 offset:%[o_cnt2]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 s96, v34\n"                   /* slots on the node list */
        "v_readfirstlane_b32 s97, v35\n"                   /* ... on the leaf list */
        "v_readfirstlane_b32 s98, v36\n"                   /* free slots */
        /* ---------------------------------------------------------------- loop header */
        "TS_LOOP_%=:\n"
        /* rays whose record was fetched a trip ago: 1 / direction, the slot's rows, onto the node list (they start at the root) */
        "s_cmp_eq_u64 s[88:89], 0\n"
        "s_cbranch_scc1 TS_NOFINAL_%=\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_mov_b64 exec, s[88:89]\n"
        "v_mov_b32_e32 v4, v60\n"
        "v_mov_b32_e32 v5, v61\n"
        "v_mov_b32_e32 v6, v62\n"
        PT_WF_ASM_INV_DIR
        "v_lshl_add_u32 v56, v59, 2, %[blk]\n"
        "v_mov_b32_e32 v33, 0\n"
        "v_mov_b32_e32 v34, -1\n"
        "ds_write_b32 v56, v64 offset:%[f0]\n"
        "ds_write_b32 v56, v65 offset:%[f1]\n"
        "ds_write_b32 v56, v66 offset:%[f2]\n"
        "ds_write_b32 v56, v33 offset:%[f3]\n"
        "ds_write_b32 v56, v4 offset:%[f4]\n"
        "ds_write_b32 v56, v5 offset:%[f5]\n"
        "ds_write_b32 v56, v6 offset:%[f6]\n"
        "ds_write_b32 v56, v33 offset:%[f7]\n"
        "ds_write_b32 v56, v8 offset:%[f8]\n"
        "ds_write_b32 v56, v9 offset:%[f9]\n"
        "ds_write_b32 v56, v10 offset:%[f10]\n"
        "ds_write_b32 v56, v58 offset:%[f11]\n"
        "ds_write_b32 v56, v33 offset:%[f12]\n"            /* wide node 0 */
        "ds_write_b32 v56, v33 offset:%[f13]\n"
        "ds_write_b32 v56, v63 offset:%[f14]\n"
        "ds_write_b32 v56, v34 offset:%[f15]\n"
        "ds_write_b32 v56, v33 offset:%[f16]\n"
        "v_add_u32_e32 v35, s96, v55\n"                    /* (the group's lanes are 0 .. take - 1) */
        "v_add_u32_e32 v35, %[blk], v35\n"
        "ds_write_b8 v35, v59 offset:%[o_qn]\n"
        "s_bcnt1_i32_b64 s71, s[88:89]\n"
        "s_add_u32 s96, s96, s71\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b64 s[88:89], 0\n"
        "TS_NOFINAL_%=:\n"
        /* rays whose id was fetched a trip ago: the gathers of direction (kind * paths + path) and origin (path) */
        "s_cmp_eq_u64 s[86:87], 0\n"
        "s_cbranch_scc1 TS_NOGATHER_%=\n"
        "s_waitcnt vmcnt(0)\n"
        "s_mov_b64 exec, s[86:87]\n"
        "v_bfe_u32 v33, v58, 28, 2\n"
        "v_and_b32_e32 v34, 0xfffffff, v58\n"
        "v_mul_lo_u32 v33, v33, %[np]\n"
        "v_add_lshl_u32 v33, v33, v34, 4\n"
        "v_lshlrev_b32_e32 v34, 4, v34\n"
        "global_load_dwordx4 v[60:63], v33, %[ray]\n"
        "global_load_dwordx4 v[64:67], v34, %[org]\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b64 s[88:89], s[86:87]\n"
        "s_mov_b64 s[86:87], 0\n"
        "s_branch TS_TRIPCHK_%=\n"                          /* (one group at a time) */
        "TS_NOGATHER_%=:\n"
        /* ---------------------------------------------------------------- a new group: free slots take the next rays of the segment */
        "s_cmp_lg_u32 s94, 0\n"
        "s_cbranch_scc1 TS_TRIPCHK_%=\n"
        "s_cmp_lt_u32 s98, %[refill_t]\n"
        "s_cbranch_scc1 TS_TRIPCHK_%=\n"
        "s_cmp_lt_u32 s70, s90\n"
        "s_cbranch_scc1 TS_ASSIGN_%=\n"
        /* the workgroup's next segment: one LDS atomic (nobody outside the workgroup touches the counter), then its ray count */
        "TS_NEXTSEG_%=:\n"
        "s_mov_b64 exec, 1\n"
        "v_mov_b32_e32 v33, %[shared]\n"
        "v_mov_b32_e32 v34, 1\n"
        "ds_add_rtn_u32 v35, v33, v34 offset:%[o_next]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 s73, v35\n"
        "s_cmp_lt_u32 s73, %[nchunks]\n"
        "s_cbranch_scc0 TS_EXHAUSTED_%=\n"
        "v_lshl_add_u32 v33, s73, 2, v33\n"
        "ds_read_b32 v35, v33\n"                          /* seg_count[c] leads the record */
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 s90, v35\n"
        "s_mov_b64 exec, -1\n"
        "s_mul_i32 s91, s73, 768\n"
        "s_add_u32 s91, s91, %[segbase]\n"
        "s_mov_b32 s70, 0\n"
        "s_cmp_lg_u32 s90, 0\n"
        "s_cbranch_scc1 TS_ASSIGN_%=\n"
        "s_branch TS_NEXTSEG_%=\n"
        "TS_EXHAUSTED_%=:\n"
        "s_mov_b64 exec, -1\n"
        "s_mov_b32 s94, 1\n"
        "s_branch TS_TRIPCHK_%=\n"
        "TS_ASSIGN_%=:\n"
        "s_sub_u32 s71, s90, s70\n"                        /* rays left in the segment */
        "s_min_u32 s71, s71, s98\n"
        "s_min_u32 s71, s71, 64\n"                         /* the group */
        "v_cmp_gt_u32_e64 s[86:87], s71, v55\n"
        "s_sub_u32 s98, s98, s71\n"
        "s_mov_b64 exec, s[86:87]\n"
        "v_add_u32_e32 v33, s98, v55\n"                    /* the topmost entries of the free list */
        "v_add_u32_e32 v33, %[blk], v33\n"
        "ds_read_u8 v59, v33 offset:%[o_free]\n"
        "v_add_u32_e32 v33, s70, v55\n"
        "v_lshl_add_u32 v33, v33, 2, s91\n"
        "global_load_dword v58, v33, %[rayq]\n"
        "s_add_u32 s70, s70, s71\n"
        "s_mov_b64 exec, -1\n"
        "TS_TRIPCHK_%=:\n"
        "s_add_u32 s99, s96, s97\n"                        /* rays on the two lists */
        "s_cmp_lg_u32 s99, 0\n"
        "s_cbranch_scc1 TS_HAVE_%=\n"
        /* none: the loop goes on while a group is moving in or segments are left */
        "s_or_b64 s[66:67], s[86:87], s[88:89]\n"
        "s_cmp_lg_u64 s[66:67], 0\n"
        "s_cbranch_scc1 TS_LOOP_%=\n"
        "s_cmp_eq_u32 s94, 0\n"
        "s_cbranch_scc1 TS_LOOP_%=\n"
        "s_branch TS_DONE_%=\n"
        "TS_HAVE_%=:\n"
        /* no segment is left, no group is moving in and only a few rays are left: they wait for the next round where they are ... */
        "s_cmp_eq_u32 s94, 0\n"
        "s_cbranch_scc1 TS_TRIP_%=\n"
        "s_or_b64 s[66:67], s[86:87], s[88:89]\n"
        "s_cmp_lg_u64 s[66:67], 0\n"
        "s_cbranch_scc1 TS_TRIP_%=\n"
        "s_cmp_ge_u32 s99, %[park_t]\n"
        "s_cbranch_scc1 TS_TRIP_%=\n"
        "s_cmp_lt_u32 s95, %[mintrips]\n"                  /* ... but every round moves its rays on by some trips */
        "s_cbranch_scc1 TS_TRIP_%=\n"
        "s_branch TS_PARK_%=\n"
        /* ---------------------------------------------------------------- one trip: a batch off one list */
        "TS_TRIP_%=:\n"
        "s_add_u32 s95, s95, 1\n"
        /* a full batch of either kind if there is one, else the longer list (a node step costs about twice a leaf step: it goes first) */
        "s_cmp_ge_u32 s96, 64\n"
        "s_cbranch_scc1 TS_KNODE_%=\n"
        "s_cmp_ge_u32 s97, 64\n"
        "s_cbranch_scc1 TS_KLEAF_%=\n"
        "s_cmp_ge_u32 s96, s97\n"
        "s_cbranch_scc1 TS_KNODE_%=\n"
        "TS_KLEAF_%=:\n"
        "s_min_u32 s99, s97, 64\n"
        "s_sub_u32 s97, s97, s99\n"
        "v_cmp_gt_u32_e64 s[64:65], s99, v55\n"
        "s_add_u32 s92, s97, %[o_ql]\n"
        "s_mov_b64 s[62:63], 0\n"
        "s_mov_b64 s[60:61], s[64:65]\n"
        "s_branch TS_POPQ_%=\n"
        "TS_KNODE_%=:\n"
        "s_min_u32 s99, s96, 64\n"
        "s_sub_u32 s96, s96, s99\n"
        "v_cmp_gt_u32_e64 s[64:65], s99, v55\n"
        "s_add_u32 s92, s96, %[o_qn]\n"
        "s_mov_b64 s[60:61], 0\n"
        "s_mov_b64 s[62:63], s[64:65]\n"
        "TS_POPQ_%=:\n"
        "s_add_u32 s92, s92, %[blk]\n"
        "s_mov_b64 exec, s[64:65]\n"
        "v_add_u32_e32 v33, s92, v55\n"
        "ds_read_u8 v57, v33\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_lshl_add_u32 v56, v57, 2, %[blk]\n"
        "v_lshlrev_b32_e32 v16, 2, v57\n"
        "v_add_u32_e32 v18, %[o_stk], v56\n"
        "ds_read_b32 v12, v56 offset:%[f12]\n"
        "ds_read_b32 v13, v56 offset:%[f13]\n"
        "ds_read_b32 v14, v56 offset:%[f14]\n"
        "ds_read_b32 v0, v56 offset:%[f0]\n"
        "ds_read_b32 v1, v56 offset:%[f1]\n"
        "ds_read_b32 v2, v56 offset:%[f2]\n"
        "s_cmp_lg_u64 s[62:63], 0\n"
        "s_cbranch_scc0 TS_LDLEAF_%=\n"
        "ds_read_b32 v8, v56 offset:%[f8]\n"
        "ds_read_b32 v9, v56 offset:%[f9]\n"
        "ds_read_b32 v10, v56 offset:%[f10]\n"
        "s_branch TS_LDDONE_%=\n"
        "TS_LDLEAF_%=:\n"
        "ds_read_b32 v4, v56 offset:%[f4]\n"
        "ds_read_b32 v5, v56 offset:%[f5]\n"
        "ds_read_b32 v6, v56 offset:%[f6]\n"
        "ds_read_b32 v11, v56 offset:%[f11]\n"
        "ds_read_b32 v20, v56 offset:%[f15]\n"
        "ds_read_b32 v21, v56 offset:%[f16]\n"
        "ds_read_b32 v22, v56 offset:%[f3]\n"
        "ds_read_b32 v23, v56 offset:%[f7]\n"
        "TS_LDDONE_%=:\n"
        "s_waitcnt lgkmcnt(0)\n"
        PT_WF_ASM_TRIP
        /* ---------------------------------------------------------------- what changed goes back to the slot's rows */
        "s_mov_b64 exec, s[64:65]\n"
        "ds_write_b32 v56, v12 offset:%[f12]\n"
        "ds_write_b32 v56, v13 offset:%[f13]\n"
        "s_cmp_lg_u64 s[60:61], 0\n"
        "s_cbranch_scc0 TS_WBDONE_%=\n"
        "ds_write_b32 v56, v14 offset:%[f14]\n"
        "ds_write_b32 v56, v20 offset:%[f15]\n"
        "ds_write_b32 v56, v21 offset:%[f16]\n"
        "ds_write_b32 v56, v22 offset:%[f3]\n"
        "ds_write_b32 v56, v23 offset:%[f7]\n"
        "TS_WBDONE_%=:\n"
        /* ---------------------------------------------------------------- the batch's rays go onto the list of their next step */
        "v_cmp_eq_u32_e64 s[66:67], -1, v12\n"             /* finished */
        "v_cmp_gt_i32_e64 s[68:69], 0, v12\n"
        "s_andn2_b64 s[68:69], s[68:69], s[66:67]\n"       /* at a leaf */
        "s_andn2_b64 s[72:73], s[64:65], s[66:67]\n"
        "s_andn2_b64 s[72:73], s[72:73], s[68:69]\n"       /* at a wide node */
        "s_mov_b64 exec, s[72:73]\n"
        "v_mbcnt_lo_u32_b32 v33, s72, 0\n"
        "v_mbcnt_hi_u32_b32 v33, s73, v33\n"
        "v_add_u32_e32 v33, s96, v33\n"
        "v_add_u32_e32 v33, %[blk], v33\n"
        "ds_write_b8 v33, v57 offset:%[o_qn]\n"
        "s_bcnt1_i32_b64 s71, s[72:73]\n"
        "s_add_u32 s96, s96, s71\n"
        "s_mov_b64 exec, s[68:69]\n"
        "v_mbcnt_lo_u32_b32 v33, s68, 0\n"
        "v_mbcnt_hi_u32_b32 v33, s69, v33\n"
        "v_add_u32_e32 v33, s97, v33\n"
        "v_add_u32_e32 v33, %[blk], v33\n"
        "ds_write_b8 v33, v57 offset:%[o_ql]\n"
        "s_bcnt1_i32_b64 s71, s[68:69]\n"
        "s_add_u32 s97, s97, s71\n"
        "s_cmp_lg_u64 s[66:67], 0\n"
        "s_cbranch_scc0 TS_NOFIN_%=\n"
        /* ---------------------------------------------------------------- finished rays: the result (a miss reports the end of the interval), the slot is free */
        "s_mov_b64 exec, s[66:67]\n"
        "s_cmp_lg_u64 s[62:63], 0\n"
        "s_cbranch_scc0 TS_FINHAVE_%=\n"
        "ds_read_b32 v11, v56 offset:%[f11]\n"            /* (a batch at wide nodes has not loaded them) */
        "ds_read_b32 v20, v56 offset:%[f15]\n"
        "ds_read_b32 v21, v56 offset:%[f16]\n"
        "ds_read_b32 v22, v56 offset:%[f3]\n"
        "ds_read_b32 v23, v56 offset:%[f7]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "TS_FINHAVE_%=:\n"
        "v_cmp_gt_i32_e32 vcc, 0, v20\n"
        "v_cndmask_b32_e32 v21, v21, v14, vcc\n"
        "v_bfe_u32 v33, v11, 28, 2\n"
        "v_and_b32_e32 v34, 0xfffffff, v11\n"
        "v_mul_lo_u32 v33, v33, %[np]\n"
        "v_add_lshl_u32 v15, v33, v34, 4\n"
        "global_store_dwordx4 v15, v[20:23], %[hit]\n"
        "v_mbcnt_lo_u32_b32 v33, s66, 0\n"
        "v_mbcnt_hi_u32_b32 v33, s67, v33\n"
        "v_add_u32_e32 v33, s98, v33\n"
        "v_add_u32_e32 v33, %[blk], v33\n"
        "ds_write_b8 v33, v57 offset:%[o_free]\n"
        "s_bcnt1_i32_b64 s71, s[66:67]\n"
        "s_add_u32 s98, s98, s71\n"
        "TS_NOFIN_%=:\n"
        "s_mov_b64 exec, -1\n"
        "s_branch TS_LOOP_%=\n"
        PT_WF_ASM_SLOW
        /* ---------------------------------------------------------------- the rays on the lists wait for the next round: their result slots say "not yet" */
        "TS_PARK_%=:\n"
        "v_mov_b32_e32 v35, -2\n"
        "s_mov_b32 s71, 0\n"
        "TS_PK1_%=:\n"
        "s_cmp_ge_u32 s71, s96\n"
        "s_cbranch_scc1 TS_PK1E_%=\n"
        "v_add_u32_e32 v33, s71, v55\n"
        "v_cmp_gt_u32_e32 vcc, s96, v33\n"
        "s_mov_b64 exec, vcc\n"
        "v_add_u32_e32 v33, %[blk], v33\n"
        "ds_read_u8 v34, v33 offset:%[o_qn]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_lshl_add_u32 v34, v34, 2, %[blk]\n"
        "ds_read_b32 v36, v34 offset:%[f11]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_bfe_u32 v33, v36, 28, 2\n"
        "v_and_b32_e32 v34, 0xfffffff, v36\n"
        "v_mul_lo_u32 v33, v33, %[np]\n"
        "v_add_lshl_u32 v33, v33, v34, 4\n"
        "global_store_dword v33, v35, %[hit]\n"
        "s_mov_b64 exec, -1\n"
        "s_add_u32 s71, s71, 64\n"
        "s_branch TS_PK1_%=\n"
        "TS_PK1E_%=:\n"
        "s_mov_b32 s71, 0\n"
        "TS_PK2_%=:\n"
        "s_cmp_ge_u32 s71, s97\n"
        "s_cbranch_scc1 TS_PK2E_%=\n"
        "v_add_u32_e32 v33, s71, v55\n"
        "v_cmp_gt_u32_e32 vcc, s97, v33\n"
        "s_mov_b64 exec, vcc\n"
        "v_add_u32_e32 v33, %[blk], v33\n"
        "ds_read_u8 v34, v33 offset:%[o_ql]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_lshl_add_u32 v34, v34, 2, %[blk]\n"
        "ds_read_b32 v36, v34 offset:%[f11]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "v_bfe_u32 v33, v36, 28, 2\n"
        "v_and_b32_e32 v34, 0xfffffff, v36\n"
        "v_mul_lo_u32 v33, v33, %[np]\n"
        "v_add_lshl_u32 v33, v33, v34, 4\n"
        "global_store_dword v33, v35, %[hit]\n"
        "s_mov_b64 exec, -1\n"
        "s_add_u32 s71, s71, 64\n"
        "s_branch TS_PK2_%=\n"
        "TS_PK2E_%=:\n"
        "s_mov_b64 exec, 1\n"
        "v_mov_b32_e32 v33, %[shared]\n"
        "v_mov_b32_e32 v34, 1\n"
        "ds_write_b32 v33, v34 offset:%[o_parked]\n"
        "TS_DONE_%=:\n"
        "s_mov_b64 exec, 1\n"
        "v_mov_b32_e32 v33, %[blk]\n"
        "v_mov_b32_e32 v34, s96\n"
        "v_mov_b32_e32 v35, s97\n"
        "v_mov_b32_e32 v36, s98\n"
        "ds_write_b32 v33, v34 offset:%[o_cnt0]\n"
        "ds_write_b32 v33, v35 offset:%[o_cnt1]\n"
        "ds_write_b32 v33, v36 offset:%[o_cnt2]\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"
        "s_mov_b64 exec, -1\n"
        :
        : [eps] "s"(s_eps), [nodes] "s"(s_nodes), [trioff] "s"(s_trioff), [spill] "s"(s_spill), [lstride] "s"(s_lstride), [blk] "s"(s_blk),
          [rayq] "s"(s_rayq), [ray] "s"(s_ray), [org] "s"(s_org), [hit] "s"(s_hit), [np] "s"(s_np), [shared] "s"(s_shared), [segbase] "s"(s_segbase),
          [depth] "n"(kWfS), [o1] "n"(kWfsRowBytes), [o2] "n"(2 * kWfsRowBytes), [o3] "n"(3 * kWfsRowBytes), [o_stk] "n"(14 * kWfsRowBytes),
          [f0] "n"(0), [f1] "n"(kWfsRowBytes), [f2] "n"(2 * kWfsRowBytes), [f3] "n"(3 * kWfsRowBytes), [f4] "n"(4 * kWfsRowBytes), [f5] "n"(5 * kWfsRowBytes),
          [f6] "n"(6 * kWfsRowBytes), [f7] "n"(7 * kWfsRowBytes), [f8] "n"(8 * kWfsRowBytes), [f9] "n"(9 * kWfsRowBytes), [f10] "n"(10 * kWfsRowBytes),
          [f11] "n"(11 * kWfsRowBytes), [f12] "n"(12 * kWfsRowBytes), [f13] "n"(13 * kWfsRowBytes), [f14] "n"(14 * kWfsRowBytes), [f15] "n"(15 * kWfsRowBytes),
          [f16] "n"(16 * kWfsRowBytes),
          [o_qn] "n"(kWfsLists), [o_ql] "n"(kWfsLists + kWfR), [o_free] "n"(kWfsLists + 2 * kWfR),
          [o_cnt0] "n"(kWfsCounts), [o_cnt1] "n"(kWfsCounts + 4), [o_cnt2] "n"(kWfsCounts + 8),
          [refill_t] "n"(PT_WF_STREAM_REFILL_T), [park_t] "n"(PT_WF_STREAM_PARK_T), [mintrips] "n"(PT_WF_STREAM_MIN_TRIPS), [nchunks] "n"(kWfWgChunks),
          [o_next] "n"(offsetof(WfShared, trace_next)), [o_parked] "n"(offsetof(WfShared, parked))
        : "memory", "vcc", "scc", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73",
          "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91",
          "s94", "s95", "s96", "s97", "s98", "s99", "s92",
          "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19",
          "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37",
          "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54",
          "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67");
}
#undef PT_WF_LVL
#undef PT_WF_O1
#undef PT_WF_O2
#undef PT_WF_O3

// ------------------------------------------------------------------------------------------- the render kernel ------
// Persistent workgroups of four waves; a workgroup owns kWfWgChunks chunks of 64 path slots (its pool) and alternates, for as long as
// it has paths or the batch has work items:
//   shade phase   the waves take the pool's chunks one at a time (LDS counter) and shade them: every slot whose rays are back
//   trace phase   the waves take the segments the shade phase filled and trace them; when the segments are used up and only a few
//                 rays of a wave are still walking, the wave parks them for the next round
// Between the phases stands a workgroup barrier - nothing device-wide: no kernel boundary per bounce, no atomic on a word another
// workgroup wants (the work-item counter aside: one claim per 256 samples), and a round waits for the slowest ray among the ~1 500
// of its own pool, not among the frame's millions.  The four workgroups of a CU are in different phases at any time: the shade
// phase's arithmetic runs while another workgroup's trace phase waits for memory.
template <int INTEG, bool WIDE, bool STREAM_MODE>
__global__ void __launch_bounds__(64 * kWfWgWaves, PT_WF_WAVES) wf_render_kernel(const DevParams P, const WfParams W)
{
    constexpr bool STREAM = WIDE && STREAM_MODE;           // (the binary tree is walked one lane per ray in both modes)
    __shared__ uint32_t lds_stack[(WIDE && !STREAM) ? 256 + kWfWgWaves * 64 * kWfStackLevels : 1];      // (768 bytes ahead of the first stack stay addressable: see s_stack)
    __shared__ uint32_t lds_ids[(STREAM || (WIDE && PT_WF_WIDE_ASM)) ? 1 : kWfWgWaves * kWfSegRays];   // the C++ walks stage a segment's ids
    __shared__ WfStreamWave lds_stream[(STREAM && PT_WF_STREAM_CXX) ? kWfWgWaves : 1];
    __shared__ __attribute__((aligned(16))) unsigned char lds_stream_asm[(STREAM && !PT_WF_STREAM_CXX) ? kWfWgWaves * kWfsBlockBytes : 16];
    __shared__ WfShared sh;
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t chunk0 = blockIdx.x * (uint32_t)kWfWgChunks;
    if (threadIdx.x == 0) {
        sh.shade_next = sh.trace_next = sh.emitted = sh.parked = 0u;
        sh.items_left = 1u;
    }
    if (STREAM && PT_WF_STREAM_CXX) {
        WfStreamWave &L = lds_stream[wv];
        for (unsigned j = lane; j < (unsigned)kWfR; j += 64u) L.freel[j] = j;
        if (lane == 0) { L.n_qn = L.n_ql = 0u; L.n_free = (uint32_t)kWfR; }
    }
    if (STREAM && !PT_WF_STREAM_CXX) {          // every slot is free
        unsigned char *blk = lds_stream_asm + wv * kWfsBlockBytes;
        for (unsigned j = lane; j < (unsigned)kWfR; j += 64u) blk[kWfsLists + 2 * kWfR + j] = (unsigned char)j;
        if (lane == 0) {
            uint32_t *cnt = reinterpret_cast<uint32_t *>(blk + kWfsCounts);
            cnt[0] = cnt[1] = 0u;
            cnt[2] = (uint32_t)kWfR;
        }
    }
    // no ray is parked for this lane (the record's third word carries the round a parked ray resumes in)
    if (WIDE && !STREAM) W.save[(size_t)((blockIdx.x * (unsigned)kWfWgWaves + wv) * 64u + lane) * (uint32_t)kWfSaveDwords + 2u] = 0u;
    __syncthreads();
#if PT_WF_PROBE == 2      // probe builds: where a wave's time goes (shader-clock cycles, lane 0 of every wave) -> P.counters[0..5]
    unsigned long long pr_shade = 0, pr_trace = 0, pr_wait = 0, pr_rounds = 0, pr_chunks = 0, pr_t0 = __builtin_readcyclecounter();
#define PT_WF_TICK(acc) { const unsigned long long now_ = __builtin_readcyclecounter(); acc += now_ - pr_t0; pr_t0 = now_; }
#else
#define PT_WF_TICK(acc)
#endif
    for (uint32_t round = 0;; ++round) {
        // ---- shade phase
        int emitted = 0;
        for (;;) {
            uint32_t c = 0;
            if (lane == 0) c = atomicAdd(&sh.shade_next, 1u);
            c = (uint32_t)__builtin_amdgcn_readfirstlane((int)c);
            if (c >= (uint32_t)kWfWgChunks) break;
            emitted += wf_shade_chunk<INTEG>(P, W, chunk0 + c, lane, &sh.seg_count[c]);
#if PT_WF_PROBE == 2
            pr_chunks++;
#endif
        }
        PT_WF_TICK(pr_shade)
        if (lane == 0 && emitted > 0) atomicAdd(&sh.emitted, (uint32_t)emitted);
        if (threadIdx.x == 0) sh.items_left = __hip_atomic_load(&W.ctrl->next_item, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < W.n_items ? 1u : 0u;
        __syncthreads();
        const bool work = sh.emitted != 0u || sh.parked != 0u;
        const bool more = sh.items_left != 0u;
        __syncthreads();
        // no ray went out, none is parked and no work item is left: every path of the pool has ended (a wave that still held samples of
        // its item would have started them: see wf_shade_chunk)
        if (!work && !more) break;
        const bool resume = sh.parked != 0u;
        if (threadIdx.x == 0) sh.shade_next = sh.trace_next = sh.emitted = sh.parked = 0u;
        __syncthreads();
        PT_WF_TICK(pr_wait)
        // ---- trace phase
        if (work) {
            if (STREAM && !PT_WF_STREAM_CXX) wf_trace_stream_asm(P, W, lds_address(lds_stream_asm + wv * kWfsBlockBytes), lds_address(&sh), chunk0);
            else if (STREAM) wf_trace_stream(P, W, sh, chunk0, lds_stream[wv], lane);
            else if (WIDE && PT_WF_WIDE_ASM) wf_trace_wide_asm(P, W, lds_address(lds_stack + 256 + wv * 64 * kWfStackLevels), lane, lds_address(&sh), chunk0, round);
            else wf_trace_cxx<WIDE>(P, W, sh, chunk0, lds_ids + wv * kWfSegRays, lds_stack + (WIDE ? 256 + wv * 64 * kWfStackLevels : 0), lane);
        }
        (void)resume;
        PT_WF_TICK(pr_trace)
        __syncthreads();
        PT_WF_TICK(pr_wait)
#if PT_WF_PROBE == 2
        pr_rounds++;
#endif
    }
#if PT_WF_PROBE == 2
    if (lane == 0) {
        atomicAdd(&P.counters[0], pr_shade); atomicAdd(&P.counters[1], pr_trace); atomicAdd(&P.counters[2], pr_wait);
        atomicAdd(&P.counters[3], pr_rounds); atomicAdd(&P.counters[4], pr_chunks); atomicAdd(&P.counters[5], 1ull);
    }
#endif
}

// ---------------------------------------------------------------------------------------------------- launchers ------
hipError_t launch_wf_render(const DevParams &P, const WfParams &W, int n_blocks, bool stream_trace, hipStream_t stream)
{
    const dim3 grid(n_blocks), block(64 * kWfWgWaves);
    const bool wide = P.traversal == GPT_TRAVERSAL_WIDE4;
#define PT_WF_LAUNCH(I) do { if (wide && stream_trace) hipLaunchKernelGGL((wf_render_kernel<I, true, true>), grid, block, 0, stream, P, W); \
                             else if (wide) hipLaunchKernelGGL((wf_render_kernel<I, true, false>), grid, block, 0, stream, P, W); \
                             else hipLaunchKernelGGL((wf_render_kernel<I, false, false>), grid, block, 0, stream, P, W); } while (0)
    if (P.integrator == GPT_IT_AO) PT_WF_LAUNCH(GPT_IT_AO);
    else if (P.integrator == GPT_IT_VPT) PT_WF_LAUNCH(GPT_IT_VPT);
    else PT_WF_LAUNCH(GPT_IT_PT);
#undef PT_WF_LAUNCH
    return hipGetLastError();
}

int wf_blocks_per_cu(int integrator, bool wide, bool stream_trace)
{
    int n = 0;
    hipError_t e;
#define PT_WF_OCC(I) ((wide && stream_trace) ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_render_kernel<I, true, true>, 64 * kWfWgWaves, 0) \
                      : wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_render_kernel<I, true, false>, 64 * kWfWgWaves, 0) \
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_render_kernel<I, false, false>, 64 * kWfWgWaves, 0))
    if (integrator == GPT_IT_AO) e = PT_WF_OCC(GPT_IT_AO);
    else if (integrator == GPT_IT_VPT) e = PT_WF_OCC(GPT_IT_VPT);
    else e = PT_WF_OCC(GPT_IT_PT);
#undef PT_WF_OCC
    if (e != hipSuccess || n < 1) { (void)hipGetLastError(); n = 2; }
    return n > 32 / kWfWgWaves ? 32 / kWfWgWaves : n;
}

// stack levels a ray keeps in LDS, and the dwords of one level in a wave's slice of WfParams::spill (one per lane, or one per slot of the stream)
int wf_lds_stack_levels(bool stream_trace) { return stream_trace ? kWfS : kWfStackLevels; }
int wf_spill_columns(bool stream_trace) { return stream_trace ? kWfR : 64; }
int wf_paths_per_block() { return 64 * kWfWgChunks; }
int wf_waves_per_block() { return kWfWgWaves; }

}  // namespace pt
