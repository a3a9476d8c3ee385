// pt_bsdf.h — what a surface does with light, for the render kernels of pt_kernel.hip.
//
// A bounce of the reference's Path() puts three questions to the same hit (src/pathtracer.cu:942-1005): the value towards the
// sampled light (Fr, :698-826), a direction for the MIS light ray and a direction for the continuation (SampleBSDF, :491-695,
// twice).  There each question is a six-way switch that derives the geometry of the hit again.  On a 64-wide wave a switch
// costs the sum of the cases its lanes hold, and a hit's three answers sit in the same lane, so the work is cut the other way:
//
//   Surface           what the three questions share, once per hit: which side the path arrives on, the tangent frame of that
//                     side, the texel, the view-side factor of the substrate's diffuse term, the masking of the arriving direction
//   surface_respond   question "value and density towards wi"
//   surface_scatter   question "a direction from three draws": proposes wi, then runs the SAME closing code as surface_respond
//
// Inside a question the lanes part by OPERATION rather than by material - lambertian and the two delta lobes leave first (a
// wave of the Cornell box never sees the rest), the rough kinds share one cosine lobe, one GGX micro-normal draw, one normal
// distribution, one masking term and meet again in weigh_rough() - so a wave that holds rough conductor, substrate and rough
// dielectric lanes runs each of these once per question instead of once per material.
//
// Float contract: every VALUE below is produced by the IEEE operations, in the order, of the reference line named
// beside it, which is what keeps the film bit-equal to oracle/pt_oracle.c; what is shared is only ever a whole subexpression
// (sharing cannot change its bits).  Three identities are used and are exact in round-to-nearest: dot(a, -b) = -dot(a, b),
// cross(a, -b) = -cross(a, b), (-x)(-y) = xy.
#pragma once

#include "pt_device.h"

// (tests/cxx/bsdf_host.cpp compiles this header for the HOST as well, to compare it with the oracle where there is no GPU)
#ifndef PT_FN
#define PT_FN __device__ __forceinline__
#endif

namespace pt {

struct Scatter {       // one answer
    V3 wi;             // direction the light arrives from / the path leaves in
    V3 f;              // BSDF value
    float pdf;         // solid-angle density of wi (1 for a delta lobe)
};

// ---------------------------------------------------------------- angles ------
// sin and cos of one angle from ONE argument reduction and one pair of kernels: the same doubles gpt_sinf / gpt_cosf
// (include/gpt_softmath.h) round, so the same floats
PT_FN void sincos_soft(float x, float &s, float &c)
{
    int quad;
    const double r = gpt_rem_pio2((double)x, &quad);
    const double ks = gpt_ksin(r), kc = gpt_kcos(r);
    const double sv = (quad & 1) ? kc : ks, cv = (quad & 1) ? ks : kc;
    s = (float)((quad & 2) ? -sv : sv);
    c = (float)(((quad + 1) & 2) ? -cv : cv);
}
// a unit vector of the y-up frame from the sine / cosine of its polar angle and of its azimuth (wrap.h:36,60)
PT_FN V3 polar_y_up(float sin_t, float cos_t, float sin_p, float cos_p)
{
    return v3(sin_t * cos_p, cos_t, sin_t * sin_p);
}
// local (x, y, z) -> x * tangent + y * normal + z * bitangent, summed left to right (wrap.h:18-20)
PT_FN V3 frame_to_world(V3 local, V3 tangent, V3 normal, V3 bitangent)
{
    return local.x * tangent + local.y * normal + local.z * bitangent;
}
// a cosine-weighted direction about +y (wrap.h:51-62) and its density cos / pi
PT_FN V3 cosine_lobe(float u1, float u2, float &density)
{
    float sin_p, cos_p;
    sincos_soft(TWOPI * u2, sin_p, cos_p);
    const float up = sqrt_rn(1.f - u1);
    density = up * ONE_OVER_PI;
    return polar_y_up(sqrt_rn(u1), up, sin_p, cos_p);
}
PT_FN V3 mirror_about(V3 w, V3 axis) { return 2.f * dot(w, axis) * axis - w; }      // pathtracer.cu:140-142
PT_FN float pow5(float x) { return x * x * x * x * x; }
// balance of two sampling densities with exponent 2 (pathtracer.cu:166-169, always called with one sample each)
PT_FN float mis_weight(float p_this, float p_other)
{
    const float a = p_this * p_this;
    return a / (a + p_other * p_other);
}
PT_FN float rr_luminance(V3 c) { return dot(c, v3(0.212671f, 0.715160f, 0.072169f)); }   // pathtracer.cu:206-208

// -------------------------------------------------------------- textures ------
// One axis of the bilinear lookup (pathtracer.cu:324-333, 346-351): the two texel indices after the repeat wrap and the
// weight of the second.  The four corners of a lookup share two of these instead of wrapping eight coordinates.
struct TexelSpan {
    int i0, i1;
    float w1;
};
PT_FN int wrap_texel(int i, int n)
{
    const float r = (float)(i - (i / n) * n);                  // C remainder, through float as the reference stores it
    int k = (int)(r < 0 ? r + n : r);
    k = k < 0 ? 0 : k;
    return k > n - 1 ? n - 1 : k;
}
PT_FN TexelSpan texel_span(float coord, int n)
{
    const float scaled = n * coord;
    const int cell = (int)__builtin_floorf(scaled);
    TexelSpan s;
    s.i0 = wrap_texel(cell, n);
    s.i1 = wrap_texel(cell + 1, n);
    s.w1 = fabs_(scaled - cell);
    return s;
}
PT_FN V3 texel_rgb(const DevTexture &t, int row, int col)
{
    const gpt_uchar4 c = t.data[row * t.width + col];
    const float to_unit = 1.f / 255.f;
    return v3(c.x * to_unit, c.y * to_unit, c.z * to_unit);
}
// GetTexel (pathtracer.cu:341-359): the material's colour, or its texture filtered bilinearly (alpha is never read)
PT_FN V3 surface_colour(const DevParams &P, const gpt_material &m, V2 uv)
{
    if (m.textureIdx == -1) return V3{m.diffuse.x, m.diffuse.y, m.diffuse.z};
    const DevTexture t = P.textures[m.textureIdx];
    const TexelSpan sx = texel_span(uv.x, t.width), sy = texel_span(uv.y, t.height);
    const V3 lower = (1 - sx.w1) * texel_rgb(t, sy.i0, sx.i0) + sx.w1 * texel_rgb(t, sy.i0, sx.i1);
    const V3 upper = (1 - sx.w1) * texel_rgb(t, sy.i1, sx.i0) + sx.w1 * texel_rgb(t, sy.i1, sx.i1);
    return (1 - sy.w1) * lower + sy.w1 * upper;
}

// ----------------------------------------------------- Fresnel reflectance ------
// Unpolarised reflectance of a smooth interface from the two cosines and the two indices.  The reference's call sites hand
// DielectricFresnel (pathtracer.cu:51-56) the TRANSMITTED cosine first and the indices swapped (:536, :679, :802); the two
// amplitude ratios below are its expressions after that substitution.
PT_FN float interface_reflectance(float cos_in, float cos_tr, float n_in, float n_tr)
{
    const float r_a = (n_in * cos_tr - n_tr * cos_in) / (n_in * cos_tr + n_tr * cos_in);
    const float r_b = (n_tr * cos_tr - n_in * cos_in) / (n_tr * cos_tr + n_in * cos_in);
    return (r_a * r_a + r_b * r_b) * 0.5f;
}
// the approximate conductor reflectance per colour channel (pathtracer.cu:58-66)
PT_FN V3 conductor_reflectance(float c, const gpt_material &m)
{
    const V3 eta = V3{m.eta.x, m.eta.y, m.eta.z}, k = V3{m.k.x, m.k.y, m.k.z};
    const V3 n2k2 = eta * eta + k * k;
    const V3 two_eta_c = eta * c * 2.f;
    const float cc = c * c;
    const V3 grazing = n2k2 * c * c;
    const V3 r_par = (grazing - two_eta_c + 1.f) / (grazing + two_eta_c + 1.f);
    const V3 r_perp = (n2k2 - two_eta_c + cc) / (n2k2 + two_eta_c + cc);
    return (r_par + r_perp) * 0.5f;
}

// --------------------------------------------------------------- Surface ------
PT_FN bool kind_is_delta(int kind) { return kind == GPT_MT_MIRROR || kind == GPT_MT_DIELECTRIC; }
PT_FN bool kind_is_rough(int kind)
{
    return kind == GPT_MT_ROUGHCONDUCTOR || kind == GPT_MT_SUBSTRATE || kind == GPT_MT_ROUGHDIELECTRIC;
}
constexpr float kSubstrateDiffuse = 28.f / (23.f * PI);      // Ashikhmin-Shirley (pathtracer.cu:623, :774)

struct Surface {
    int kind;          // GPT_MT_*
    V3 wo;             // unit vector back along the arriving ray ("in" of the reference)
    V3 ng;             // the interpolated normal as mesh.h:87-90 gives it
    V3 tu;             // dp/du, the frame's first axis
    float wo_ng;       // wo . ng: its sign is the side the path arrives on
    // The lobes are built about ng turned to wo's side (pathtracer.cu:494-496, 555-557, 582-584, 724-726, 749-751); the rough
    // dielectric keeps ng as it is (:644, :789).  Only the decision is stored - a lane mask, no vector register - and the turned
    // normal, the frame's third axis and wo's cosine are formed where a question needs them: what stays in registers from one
    // question to the next is what was expensive to get.
    bool turn;
    V3 base;           // lambertian: albedo / pi;  substrate: (28 / 23 pi) Rd (1 - Rs) (1 - (1 - |wo.n| / 2)^5)
    float mask_wo;     // rough conductor / dielectric: Smith masking of wo, before its micro-facet side test

    PT_FN V3 nm() const { return turn ? -ng : ng; }
    PT_FN V3 tb() const { return cross(tu, nm()); }
    PT_FN float wo_nm() const { return turn ? -wo_ng : wo_ng; }
};

// Smith masking of a direction w with w . n = w_n about the normal n, anisotropic GGX (pathtracer.cu:86-101) - all of it but the
// test against the micro-normal, which is the only part that changes from question to question (mask_side)
PT_FN float mask_shape(V3 w, float w_n, V3 n, V3 tu, float a_uu, float a_vv)
{
    const float sin_t = sqrt_rn(clamp(1.f - w_n * w_n, 0.f, 1.f));
    const float tan_t = sin_t / w_n;
    const V3 flat = normalize(w - w_n * n);              // w's direction within the tangent plane
    const float cos_p = dot(flat, tu);
    const float cos_p2 = cos_p * cos_p;
    const float a2 = cos_p2 * a_uu + (1.f - cos_p2) * a_vv;
    const float g = 2.f / (1.f + sqrt_rn(1 + a2 * tan_t * tan_t));
    return gpt_isinff(tan_t) ? 0.f : g;
}
PT_FN float mask_side(float shape, float w_n, float w_wh) { return w_n * w_wh < 0.f ? 0.f : shape; }

// The two terms a Surface keeps for its three questions.  Measured on the MI355X (profiles/r06/a2_variants.log): with them formed once
// per hit the config-3 stand-in runs 9 % faster than with each question forming its own.
PT_FN V3 surface_base(const DevParams &P, const Surface &S, const gpt_material &m, V2 uv)
{
    if (S.kind == GPT_MT_LAMBERTIAN) return surface_colour(P, m, uv) * ONE_OVER_PI;                                      // :503, :707
    if (S.kind == GPT_MT_SUBSTRATE) {
        const V3 rs = V3{m.specular.x, m.specular.y, m.specular.z};
        const float k0 = 1 - 0.5f * fabs_(S.wo_nm());
        return kSubstrateDiffuse * surface_colour(P, m, uv) * (v3(1.f, 1.f, 1.f) - rs) * (1 - pow5(k0));             // :623-624
    }
    return v3(0.f);
}
PT_FN float surface_mask_wo(const Surface &S, const gpt_material &m)
{
    return mask_shape(S.wo, S.wo_nm(), S.nm(), S.tu, m.alphaU * m.alphaU, m.alphaV * m.alphaV);
}
// `uv` is only read by the kinds that have a colour (lambertian, substrate)
PT_FN Surface surface_prepare(const DevParams &P, const gpt_material &m, V3 wo, V3 nor, V3 dpdu, V2 uv)
{
    Surface S;
    S.kind = m.type;
    S.wo = wo;
    S.ng = nor;
    S.tu = dpdu;
    S.wo_ng = dot(nor, wo);
    S.turn = S.wo_ng < 0 && S.kind != GPT_MT_ROUGHDIELECTRIC;
    S.base = v3(0.f);
    S.mask_wo = 0.f;
    S.base = surface_base(P, S, m, uv);
    if (S.kind == GPT_MT_ROUGHCONDUCTOR || S.kind == GPT_MT_ROUGHDIELECTRIC) S.mask_wo = surface_mask_wo(S, m);
    return S;
}

// GGX normal distribution about the normal nm, anisotropic along S.tu (pathtracer.cu:68-84)
PT_FN float ndf(const Surface &S, const gpt_material &m, V3 nm, V3 wh, float wh_nm)
{
    const float c = clamp(wh_nm, 0.f, 1.f);
    const float c2 = c * c;
    const float tan2 = (1.f - c2) / c2;
    const V3 flat = normalize(wh - c * nm);
    const float cos_p = dot(flat, S.tu);
    const float cos_p2 = cos_p * cos_p;
    const float stretch = 1.f + tan2 * (cos_p2 / (m.alphaU * m.alphaU) + (1.f - cos_p2) / (m.alphaV * m.alphaV));
    const float d = 1.f / (PI * m.alphaU * m.alphaV * (c2 * c2) * stretch * stretch);
    return wh_nm <= 0.f ? 0.f : d;
}

// A micro-normal of the GGX distribution in the y-up frame (pathtracer.cu:107-138).  Isotropic: closed form in u1 with the
// azimuth 2 pi u2; anisotropic: the azimuth is stretched by alphaV / alphaU quadrant by quadrant and the polar angle follows.
PT_FN V3 micro_normal(const gpt_material &m, float u1, float u2)
{
    float sin_t, cos_t, sin_p, cos_p;
    if (m.alphaU == m.alphaV) {
        cos_t = sqrt_rn((1.f - u1) / (u1 * (m.alphaU * m.alphaV - 1.f) + 1.f));
        sin_t = sqrt_rn(1.f - cos_t * cos_t);
        sincos_soft(TWOPI * u2, sin_p, cos_p);             // the reference writes 2 * PI * u2: the same float
    } else {
        float phi = gpt_atanf(m.alphaV / m.alphaU * gpt_tanf(TWOPI * u2));
        if (!(u2 <= 0.25f)) phi = phi + (u2 >= 0.75f ? TWOPI : PI);     // (adding 0 would turn a -0 into +0)
        sincos_soft(phi, sin_p, cos_p);
        const float sin_p2 = sin_p * sin_p;
        const float spread = 1.0f / ((1.0f - sin_p2) / (m.alphaU * m.alphaU) + sin_p2 / (m.alphaV * m.alphaV));
        sincos_soft(gpt_atanf(sqrt_rn(spread * u1 / (1.0f - u1))), sin_t, cos_t);
    }
    return polar_y_up(sin_t, cos_t, sin_p, cos_p);
}

// which closing formula a rough answer takes
enum : int {
    kCloseMirrorLike = 0,     // rough conductor; rough dielectric beyond the critical angle (no Fresnel factor)
    kCloseReflectDrawn = 1,   // rough dielectric, reflection chosen by a draw: density = D |wh.n| / (4 |wh.wo|) * F   (:691)
    kCloseReflectAsked = 2,   // rough dielectric, reflection evaluated:        density = F * D |wh.n| / (4 |wh.wo|)   (:820)
    kCloseRefract = 3,        // rough dielectric, transmission
    kCloseSubstrate = 4
};

// The two sides of a dielectric boundary as the arriving path meets them (pathtracer.cu:516-523, 650-657, 793-800)
struct Boundary {
    float n_here, n_there;
    bool entering;
};
PT_FN Boundary boundary_of(const Surface &S, const gpt_material &m)
{
    Boundary b;
    b.entering = -S.wo_ng < 0;                              // dot(-wo, ng) < 0
    b.n_here = b.entering ? m.outsideIOR : m.insideIOR;
    b.n_there = b.entering ? m.insideIOR : m.outsideIOR;
    return b;
}
// ... and what a facet with normal `axis` (wo . axis = wo_axis) does there: the transmitted cosine, whether anything is
// transmitted at all, the reflectance
struct Crossing {
    float ratio, cos_i, sin_t2, cos_t, reflectance;
};
PT_FN Crossing crossing_of(const Boundary &b, float wo_axis)
{
    Crossing x;
    x.ratio = b.n_here / b.n_there;
    x.cos_i = -wo_axis;                                     // dot(-wo, axis)
    x.sin_t2 = x.ratio * x.ratio * (1.f - x.cos_i * x.cos_i);
    const float under = 1.f - x.sin_t2;
    x.cos_t = sqrt_rn(under < 0.f ? 0.f : under);
    x.reflectance = interface_reflectance(fabs_(x.cos_i), fabs_(x.cos_t), b.n_here, b.n_there);
    return x;
}

// The closing code of every rough answer: wi is known, wh is the micro-normal that links it to wo.  One normal distribution,
// one masking term of wi (wo's comes from the Surface), then the formula of the lobe.
PT_FN void weigh_rough(const Surface &S, const gpt_material &m, V3 wh, int close, float reflectance, Scatter &r)
{
    const V3 spec = V3{m.specular.x, m.specular.y, m.specular.z};
    const V3 nm = S.nm();
    const float wo_nm = S.wo_nm();
    const float wh_nm = dot(wh, nm);
    const float D = ndf(S, m, nm, wh, wh_nm);
    const float wo_wh = dot(S.wo, wh), wi_wh = dot(r.wi, wh);
    const float wi_nm = dot(r.wi, nm);
    if (close == kCloseSubstrate) {
        // pathtracer.cu:617-634, 758-781: the diffuse term's light-side factor, the Schlick-weighted specular term
        const float c0 = fabs_(wo_nm), c1 = fabs_(wi_nm);
        const float k1 = 1 - 0.5f * c1;
        const V3 matte = S.base * (1 - pow5(k1));
        const V3 white_minus_rs = v3(1.f, 1.f, 1.f) - spec;
        const V3 schlick = spec + pow5(1.f - wi_wh) * white_minus_rs;
        const V3 gloss = D / (4.f * fabs_(wi_wh) * (c0 > c1 ? c0 : c1)) * schlick;
        r.f = matte + gloss;
        r.pdf = 0.5f * (c1 * ONE_OVER_PI + D * fabs_(wh_nm) / (4.f * wo_wh));
        return;
    }
    const float G = mask_side(S.mask_wo, wo_nm, wo_wh) * mask_side(mask_shape(r.wi, wi_nm, nm, S.tu, m.alphaU * m.alphaU, m.alphaV * m.alphaV), wi_nm, wi_wh);
    if (close == kCloseRefract) {
        // pathtracer.cu:680-688, 806-813 (radiance transport)
        const Boundary b = boundary_of(S, m);
        const float n_i = b.n_here, n_t = b.n_there;
        const float c = n_t * wi_wh + n_i * wo_wh;
        const float ratio = n_i / n_t;
        r.f = spec * n_i * n_i * D * G * (1.f - reflectance) * fabs_(wo_wh) * fabs_(wi_wh) / (fabs_(wi_nm) * fabs_(wo_nm) * c * c);
        r.f *= (1.f / (ratio * ratio));
        r.pdf = (1.f - reflectance) * D * fabs_(wh_nm) * n_t * n_t * fabs_(wi_wh) / (c * c);
        return;
    }
    // the reflecting lobes: rho F D G / (4 |wo.n| |wi.n|) with F per channel (conductor, :569-577, 733-739), the scalar
    // reflectance of the boundary (:690, :817) or nothing (total reflection, :672-673: multiplying by 1 is exact)
    V3 F = v3(1.f, 1.f, 1.f);
    if (S.kind == GPT_MT_ROUGHCONDUCTOR) F = conductor_reflectance(fabs_(wi_wh), m);
    else if (close != kCloseMirrorLike) F = v3(reflectance, reflectance, reflectance);
    r.f = spec * F * D * G / (4.f * fabs_(wo_nm) * fabs_(wi_nm));
    const float spread = 4.f * fabs_(wo_wh);
    if (close == kCloseReflectAsked) r.pdf = reflectance * D * fabs_(wh_nm) / spread;
    else {
        r.pdf = D * fabs_(wh_nm) / spread;
        if (close == kCloseReflectDrawn) r.pdf = r.pdf * reflectance;
    }
}

// ---------------------------------------------------------------- respond ------
// Fr (pathtracer.cu:698-826): value and density of the surface towards wi.  Delta lobes and unknown kinds answer zero.
PT_FN Scatter surface_respond(const Surface &S, const gpt_material &m, V3 wi)
{
    Scatter r;
    r.wi = wi;
    r.f = v3(0.f, 0.f, 0.f);
    r.pdf = 0.f;
    const float wi_ng = dot(wi, S.ng);
    const bool same_side = S.wo_ng * wi_ng > 0;
    if (S.kind == GPT_MT_LAMBERTIAN) {
        if (same_side) {
            r.f = S.base;
            r.pdf = fabs_(wi_ng) * ONE_OVER_PI;
        }
    } else if (kind_is_rough(S.kind)) {
        const bool boundary = S.kind == GPT_MT_ROUGHDIELECTRIC;
        if (boundary || same_side) {
            Boundary b = boundary_of(S, m);
            // the micro-normal that links wo and wi: their bisector, or across a boundary the index-weighted one (:802)
            const V3 wh = normalize(boundary ? -(b.n_here * S.wo + b.n_there * wi) : S.wo + wi);
            float reflectance = 1.f;
            int close = S.kind == GPT_MT_SUBSTRATE ? kCloseSubstrate : kCloseMirrorLike;
            if (boundary) {
                reflectance = crossing_of(b, dot(S.wo, wh)).reflectance;
                close = same_side ? kCloseReflectAsked : kCloseRefract;
            }
            weigh_rough(S, m, wh, close, reflectance, r);
        }
    }
    return r;
}

// ---------------------------------------------------------------- scatter ------
// SampleBSDF (pathtracer.cu:491-695): a direction from the draws (u1, u2, u3) with its value and density.
PT_FN Scatter surface_scatter(const Surface &S, const gpt_material &m, float u1, float u2, float u3)
{
    Scatter r;
    r.wi = v3(0.f, 0.f, 0.f);
    r.f = v3(0.f, 0.f, 0.f);
    r.pdf = 0.f;
    const V3 spec = V3{m.specular.x, m.specular.y, m.specular.z};
    if (S.kind == GPT_MT_LAMBERTIAN) {
        // cosine-weighted about the arriving side's normal (wrap.h:51-62); the value does not look at the direction
        r.wi = frame_to_world(cosine_lobe(u1, u2, r.pdf), S.tu, S.nm(), S.tb());
        r.f = S.base;
    } else if (S.kind == GPT_MT_MIRROR) {
        r.wi = mirror_about(S.wo, S.ng);                                       // :506-510
        r.f = spec / fabs_(dot(r.wi, S.ng));
        r.pdf = 1.f;
    } else if (S.kind == GPT_MT_DIELECTRIC) {
        // :512-551.  The transmitted direction is Refract()'s (:144-158), whose sin^2 multiplies in a different order from the
        // one the total-reflection test uses - both are kept, they can differ in the last bit at the critical angle.
        const Boundary b = boundary_of(S, m);
        const Crossing x = crossing_of(b, S.wo_ng);
        const V3 bounce = mirror_about(S.wo, S.ng);
        const float sin_t2_refract = (1.f - S.wo_ng * S.wo_ng) * x.ratio * x.ratio;
        const float cos_t_refract = sqrt_rn(1.f - sin_t2_refract);
        const V3 through = normalize((S.ng * S.wo_ng - S.wo) * x.ratio + (b.entering ? -cos_t_refract : cos_t_refract) * S.ng);
        const bool total = x.sin_t2 > 1.f;
        const bool transmit = !total && u1 > x.reflectance;
        r.wi = transmit ? through : bounce;
        const V3 per_cos = spec / fabs_(dot(r.wi, S.ng));
        if (total) {
            r.f = per_cos;
            r.pdf = 1.f;
        } else if (transmit) {
            r.f = per_cos * (1.f - x.reflectance);
            r.f *= x.ratio * x.ratio;                                          // radiance transport
            r.pdf = 1.f - x.reflectance;
        } else {
            r.f = per_cos * x.reflectance;
            r.pdf = x.reflectance;
        }
    } else if (kind_is_rough(S.kind)) {
        const bool boundary = S.kind == GPT_MT_ROUGHDIELECTRIC, layered = S.kind == GPT_MT_SUBSTRATE;
        // the substrate spends its first draw on the choice of lobe and stretches the half it fell in back to [0, 1) (:585-598)
        const bool matte_lobe = layered && u1 < 0.5f;
        const float v1 = layered ? (matte_lobe ? u1 * 2.f : (u1 - 0.5f) * 2.f) : u1;
        Boundary b = boundary_of(S, m);
        float reflectance = 1.f;
        int close = layered ? kCloseSubstrate : kCloseMirrorLike;
        V3 wh = v3(0.f, 0.f, 0.f);
        if (matte_lobe) {
            float lobe_density;                              // (the answer's density covers both lobes: weigh_rough)
            r.wi = frame_to_world(cosine_lobe(v1, u2, lobe_density), S.tu, S.nm(), S.tb());
        } else {
            wh = frame_to_world(micro_normal(m, v1, u2), S.tu, S.nm(), S.tb());
            const float wo_wh = dot(S.wo, wh);
            r.wi = mirror_about(S.wo, wh);
            if (boundary) {
                // :659-693: the facet reflects or transmits by the third draw; beyond the critical angle it can only reflect
                const Crossing x = crossing_of(b, wo_wh);
                reflectance = x.reflectance;
                if (!(x.sin_t2 > 1.f)) {
                    close = kCloseReflectDrawn;
                    if (u3 > x.reflectance) {
                        close = kCloseRefract;
                        r.wi = normalize((-S.wo - wh * x.cos_i) * x.ratio + (b.entering ? -x.cos_t : x.cos_t) * wh);
                    }
                }
            }
        }
        // a proposal that lands on the other side of the surface is void (:563-567, 599-603); the boundary has both sides
        if (boundary || S.wo_ng * dot(r.wi, S.ng) > 0) {
            if (layered) wh = normalize(S.wo + r.wi);                                        // :625
            weigh_rough(S, m, wh, close, reflectance, r);
        }
    }
    return r;
}

}  // namespace pt
