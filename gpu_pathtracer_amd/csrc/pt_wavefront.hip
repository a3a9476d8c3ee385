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
template <int INTEG>
__global__ void __launch_bounds__(256) wf_shade_kernel(const DevParams P, const WfParams W)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;           // path slot (n_paths is a multiple of 256)
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t np = W.n_paths;
    const unsigned par = W.round & 1u;
    if (i == 0) W.ctrl->head = 0u;                                 // the trace stage of this round starts at ray 0

    const float4 a3 = W.s3[i];
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

    bool finish = false;
    if (alive) {
        const float4 a0 = W.s0[i], a1 = W.s1[i], a2 = W.s2[i], a4 = W.s4[i], ao = W.org[i];
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
            const float4 r = W.ray[i], h = W.hit[i];
            dir_p = xyz(r);
            res.prim_p = __float_as_int(h.x); res.t_p = h.y; res.b1_p = h.z; res.b2_p = h.w;
        }
        if (has_m) {
            const float4 r = W.ray[np + i], h = W.hit[np + i];
            dir_m = xyz(r);
            res.prim_m = __float_as_int(h.x); res.t_m = h.y; res.b1_m = h.z; res.b2_m = h.w;
        }
        if (has_s) res.occluded = __float_as_int(W.hit[2u * np + i].x) >= 0;

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

    if (finish) {
        // The sample goes to its iteration's plane as is; the finite-guard of pathtracer.cu:1019-1020 and the accumulation
        // run in iteration order in pt_output_kernel.
        reinterpret_cast<float4 *>(P.samples)[dst] = make_float4(Li.x, Li.y, Li.z, 0.f);
        alive = false;
        has_s = has_m = has_p = false;
    }

    // ---- regenerate: pathtracer.cu:881-903.  Sample s of the batch = pixel s % 64 of iteration (s / 64) % iter_count of
    // owned tile s / (64 iter_count) (tiles in strips, like the per-wave kernel's items): the 64 lanes of a wave that start
    // together start one 8x8 tile, and the paths in flight cover a compact block of the frame.
    const bool was_alive = (flags & kWfAlive) != 0u;
    bool start = false;
    {
        const unsigned long long m_idle = ballot(!alive);
        if (m_idle != 0ull) {
            unsigned long long base = 0;
            if (lane == 0) {       // (every lane is here: the branch is wave-uniform)
                // (a stale read can only be too small: then the atomic tells)
                const unsigned long long seen = __hip_atomic_load(&W.ctrl->next_sample, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                base = seen >= W.n_samples ? seen : atomicAdd(&W.ctrl->next_sample, (unsigned long long)popc(m_idle));
            }
            base = uniform64(base);
            const unsigned long long s64 = base + (unsigned)lane_rank(m_idle);
            if (!alive && s64 < W.n_samples) {
                const uint32_t s = (uint32_t)s64;
                const uint32_t per = 64u * P.iter_count;
                const uint32_t tile_seq = s / per, r = s - tile_seq * per;
                const uint32_t iter_rel = r >> 6, pix = r & 63u;
                const uint32_t n_owned = (uint32_t)(P.plane >> 6);
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
                const uint32_t x = (tile % P.tiles_x) * 8u + (pix & 7u), y = (tile / P.tiles_x) * 8u + (pix >> 3);
                if (x < P.stride && y < P.rows) {
                    start = true;
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
            }
        }
    }
    (void)start;

    // ---- this round's rays go to the queue: one atomic per wave; path rays first, then light rays, then shadow rays ----
    {
        const unsigned long long m_p = ballot(has_p), m_m = ballot(has_m), m_s = ballot(has_s);
        const int n_p = popc(m_p), n_m = popc(m_m), n_all = n_p + n_m + popc(m_s);
        if (n_all > 0) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&W.ctrl->n_rays[par], (uint32_t)n_all);
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (has_p) {
                W.rayq[base + (uint32_t)lane_rank(m_p)] = i;
                W.ray[i] = f4(dir_p, __builtin_inff());
            }
            if (has_m) {
                W.rayq[base + (uint32_t)(n_p + lane_rank(m_m))] = i | (1u << kWfKindShift) | (mis_any ? kWfAnyHit : 0u);
                W.ray[np + i] = f4(dir_m, __builtin_inff());
            }
            if (has_s) {
                W.rayq[base + (uint32_t)(n_p + n_m + lane_rank(m_s))] = i | (2u << kWfKindShift) | kWfAnyHit;
                W.ray[2u * np + i] = f4(dir_s, tmax_s);
            }
        }
    }

    // ---- the slot's state for the next round ----
    if (alive) {
        flags = (uint32_t)bounces | (specular ? kWfSpecular : 0u) | (direct ? kWfDirect : 0u) | (ending ? kWfEnding : 0u) | kWfAlive |
                (has_p ? kWfHasP : 0u) | (has_m ? kWfHasM : 0u) | (has_s ? kWfHasS : 0u) | (mis_any ? kWfMisAny : 0u) |
                (poison_occluded ? kWfPoison : 0u);
        W.s0[i] = make_float4(Li.x, Li.y, Li.z, beta.x);
        W.s1[i] = make_float4(beta.y, beta.z, mis_cos, mis_pdf);
        W.s2[i] = f4u(cand, rng.x);
        W.s3[i] = f4u(beta_ld, flags);
        W.s4[i] = f4u(mis_fr, dst);
        W.org[i] = f4u(org, INTEG == GPT_IT_VPT ? (((uint32_t)medium & 0xffffu) | ((uint32_t)medium_ld << 16)) : 0u);
    } else if (was_alive) {
        W.s3[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------------------------------------ trace stage ------
// Persistent waves over the round's ray queue.  A wave claims kWfChunk consecutive ray ids with one atomic and keeps them in
// LDS; a lane that finishes a ray (its result goes straight to hit[kind][path]) takes the next id of the chunk, reads the
// ray's direction and origin from the path's planes and starts at the root.  The walks are those of pt_kernel.hip:
//   WIDE   the 4-wide tree, one lane per ray, per-lane stack in LDS (include/gpt_wide_bvh.h; trace_pool_wide<>)
//   !WIDE  the reference's order on the threaded binary tree (trace_pool<>)
// every box and triangle test in the same arithmetic (bbox.h:77-96, mesh.h:45-67).
#ifndef PT_WF_CHUNK
#define PT_WF_CHUNK 256
#endif
#ifndef PT_WF_STACK_LEVELS
#define PT_WF_STACK_LEVELS 24
#endif
#ifndef PT_WF_FETCH_T
#define PT_WF_FETCH_T 8              // idle lanes that trigger a refill
#endif
#ifndef PT_WF_TRACE_WAVES
#define PT_WF_TRACE_WAVES 4
#endif
constexpr int kWfChunk = PT_WF_CHUNK, kWfStackLevels = PT_WF_STACK_LEVELS;

__device__ __forceinline__ void wf_cex(unsigned &ka, unsigned &ea, unsigned &kb, unsigned &eb)
{
    const bool swap = kb < ka;
    const unsigned k0 = swap ? kb : ka, k1 = swap ? ka : kb, e0 = swap ? eb : ea, e1 = swap ? ea : eb;
    ka = k0; kb = k1; ea = e0; eb = e1;
}

template <bool WIDE>
__global__ void __launch_bounds__(256, PT_WF_TRACE_WAVES) wf_trace_kernel(const DevParams P, const WfParams W)
{
    __shared__ uint32_t lds_ids[4 * kWfChunk];
    __shared__ uint32_t lds_stack[WIDE ? 4 * 64 * kWfStackLevels : 1];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    uint32_t *ids = lds_ids + wv * kWfChunk;
    const unsigned par = W.round & 1u;
    const uint32_t n_rays = __hip_atomic_load(&W.ctrl->n_rays[par], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        W.ctrl->n_rays[par ^ 1u] = 0u;                            // the next round's shade stage counts from 0
        // progress for the host's round loop: this round had no rays and no sample is left = the batch is complete
        const unsigned long long next = __hip_atomic_load(&W.ctrl->next_sample, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long done = (n_rays == 0u && next >= W.n_samples) ? 1ull : 0ull;
        __hip_atomic_store(W.host_flag, ((unsigned long long)W.seq << 32) | ((unsigned long long)(W.round & 0x3fffffffu) << 1) | done,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const uint32_t np = W.n_paths;
    const float tmin_ray = P.eps;
    const char *tris = reinterpret_cast<const char *>(P.tris);

    // the wave's chunk of the queue: ids [cursor, chunk_end) of which ids[0 ..] holds [chunk_base, chunk_end)
    uint32_t cursor = 0, chunk_base = 0, chunk_end = 0;
    bool exhausted = n_rays == 0u;

    // lane state: one ray
    uint32_t id = 0;
    bool has = false;
    V3 o = v3(0.f), d = v3(0.f), inv = v3(0.f);
    float tmax = 0.f;
    // wide walk
    unsigned *stk = lds_stack + (WIDE ? wv * 64 * kWfStackLevels : 0) + lane;
    uint32_t *spill = W.spill + (size_t)(blockIdx.x * 4u + wv) * 64u * W.spill_levels + lane;
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
            if (cursor >= chunk_end) {
                // ---- claim the next chunk of the queue (one atomic per kWfChunk rays)
                uint32_t b = 0;
                if (lane == 0) b = atomicAdd(&W.ctrl->head, (uint32_t)kWfChunk);
                b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                if (b >= n_rays) {
                    exhausted = true;
                } else {
                    chunk_base = cursor = b;
                    chunk_end = b + (uint32_t)kWfChunk < n_rays ? b + (uint32_t)kWfChunk : n_rays;
#pragma unroll
                    for (int k = 0; k < kWfChunk / 64; ++k) {
                        const uint32_t at = b + (uint32_t)(k * 64) + lane;
                        if (at < chunk_end) ids[k * 64 + (int)lane] = W.rayq[at];
                    }
                    wave_lds_fence();
                }
            }
            if (!exhausted) {
                // ---- refill: idle lanes take the next rays of the chunk, in lane order
                const uint32_t nth = cursor + (uint32_t)lane_rank(~m_busy);
                if (!has && nth < chunk_end) {
                    id = ids[nth - chunk_base];
                    const uint32_t path = id & kWfPathMask, kind = (id >> kWfKindShift) & 3u;
                    const float4 r0 = W.ray[kind * np + path];
                    const float4 ro = W.org[path];
                    has = true;
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

// ---------------------------------------------------------------------------------------------------- launchers ------
hipError_t launch_wf_shade(const DevParams &P, const WfParams &W, hipStream_t stream)
{
    const dim3 grid(W.n_paths / 256u), block(256);
    if (P.integrator == GPT_IT_AO) hipLaunchKernelGGL((wf_shade_kernel<GPT_IT_AO>), grid, block, 0, stream, P, W);
    else if (P.integrator == GPT_IT_VPT) hipLaunchKernelGGL((wf_shade_kernel<GPT_IT_VPT>), grid, block, 0, stream, P, W);
    else hipLaunchKernelGGL((wf_shade_kernel<GPT_IT_PT>), grid, block, 0, stream, P, W);
    return hipGetLastError();
}

hipError_t launch_wf_trace(const DevParams &P, const WfParams &W, int n_blocks, hipStream_t stream)
{
    if (P.traversal == GPT_TRAVERSAL_WIDE4) hipLaunchKernelGGL((wf_trace_kernel<true>), dim3(n_blocks), dim3(256), 0, stream, P, W);
    else hipLaunchKernelGGL((wf_trace_kernel<false>), dim3(n_blocks), dim3(256), 0, stream, P, W);
    return hipGetLastError();
}

int wf_trace_blocks_per_cu(bool wide)
{
    int n = 0;
    const hipError_t e = wide ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_trace_kernel<true>, 256, 0)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, wf_trace_kernel<false>, 256, 0);
    if (e != hipSuccess || n < 1) { (void)hipGetLastError(); n = 2; }
    return n > 8 ? 8 : n;
}

int wf_lds_stack_levels() { return kWfStackLevels; }

}  // namespace pt
