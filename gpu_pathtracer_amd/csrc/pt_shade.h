// pt_shade.h — what a path evaluates around a surface hit besides the surface's own scattering (pt_bsdf.h): the hit record, the camera,
// the lights, the film's tone curve, participating media.  Each function names the reference lines whose values it reproduces; the
// operation order is the one oracle/pt_oracle.c follows (bit-exact contract, DESIGN.md "Float contract").
#pragma once

#include "pt_device.h"
#include "pt_bsdf.h"

namespace pt {

// mesh.h:68-95 evaluated once for the final hit
__device__ __forceinline__ Hit make_hit(const DevParams &P, const Ray &ray, float tt, int prim, float b1, float b2)
{
    const float4 *__restrict__ sp = reinterpret_cast<const float4 *>(P.shade) + 5 * prim;
    const float4 s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4];
    const V3 n1 = V3{s0.x, s0.y, s0.z}, n2 = V3{s0.w, s1.x, s1.y}, n3 = V3{s1.z, s1.w, s2.x};
    const V2 uv1 = V2{s2.y, s2.z}, uv2 = V2{s2.w, s3.x}, uv3 = V2{s3.y, s3.z};
    const V3 ndpdv = V3{s3.w, s4.x, s4.y};
    Hit h;
    h.pos = ray.o + tt * ray.d;
    h.nor = normalize(n1 * (1.f - b1 - b2) + n2 * b1 + n3 * b2);
    h.uv = uv1 * (1.f - b1 - b2) + uv2 * b1 + uv3 * b2;
    h.matIdx = __float_as_int(s4.z);
    h.lightIdx = __float_as_int(s4.w);
    h.dpdu = normalize(cross(h.nor, ndpdv));
    return h;
}

// ------------------------------------------------------------- samplers ------
// a direction uniform over the sphere, y-up (wrap.h:26-36); its density is the constant 1 / 4 pi
__device__ __forceinline__ V3 sphere_direction(float u1, float u2)
{
    float sin_p, cos_p;
    sincos_soft(TWOPI * u2, sin_p, cos_p);
    const float cos_t = 1.f - 2.f * u1;
    return polar_y_up(sqrt_rn(1.f - cos_t * cos_t), cos_t, sin_p, cos_p);
}

// --------------------------------------------------------------- camera ------
// `du1, du2` are the two draws of UniformDisk (pathtracer.cu:895, wrap.h:78-85).  The reference always
// evaluates the disk sample; only the thin-lens branch reads it, so its sin/cos is evaluated there.
__device__ __forceinline__ Ray primary_ray(const gpt_camera &c, float x, float y, float du1, float du2)   // camera.h:48-84
{
    const V3 cu = V3{c.u.x, c.u.y, c.u.z}, cv = V3{c.v.x, c.v.y, c.v.z}, cw = V3{c.w.x, c.w.y, c.w.z};
    Ray ray;
    ray.tmin = 0.001f;
    ray.tmax = __builtin_inff();
    V3 orig = V3{c.position.x, c.position.y, c.position.z};
    if (c.environment) {
        float theta = PI * (1.f - y / c.resolution.y);
        float phi = TWOPI * (1.f - x / c.resolution.x);
        V3 dir = v3(gpt_sinf(theta) * gpt_cosf(phi), gpt_cosf(theta), gpt_sinf(theta) * gpt_sinf(phi));
        dir = dir.x * cu + dir.y * cv - dir.z * cw;
        ray.o = orig;
        ray.d = dir;
        return ray;
    }
    float xx = x * c.pixel2screen.x - c.width;
    float yy = y * c.pixel2screen.y - c.height;
    V3 dir;
    if (c.apertureRadius > 0.00001f) {
        float rr = sqrt_rn(du1), sin_p, cos_p;    // UniformDisk
        sincos_soft(TWOPI * du2, sin_p, cos_p);
        V2 xy = v2(rr * cos_p, rr * sin_p);
        V2 aperture_xy = xy * c.apertureRadius;
        float focal_x = c.ratio * xx;
        float focal_y = c.ratio * yy;
        V3 aperture = v3(aperture_xy.x, aperture_xy.y, 0);
        V3 focal = v3(focal_x, focal_y, -c.focalDistance);
        dir = focal - aperture;
        dir = dir.x * cu + dir.y * cv + dir.z * cw;
        orig += (aperture.x * cu + aperture.y * cv);
    } else {
        dir = xx * cu + yy * cv + -c.distance * cw;
    }
    dir = normalize(dir);
    ray.o = orig;
    ray.d = dir;
    return ray;
}

// --------------------------------------------------------------- lights ------
// Area::SampleLight (area.h:14-19) -> Triangle::SampleShape (mesh.h:100-109)
__device__ __forceinline__ void area_sample_light(const DevLight &L, V3 pos, V2 u, V3 &rad, Ray &ray, V3 &nor,
                                                  float &pdf, float eps)
{
    float su1 = sqrt_rn(u.x);                         // UniformTriangle, wrap.h:110-115
    V2 uv = v2(1.f - su1, u.y * su1);
    V3 p = uv.x * ld3(L.v1) + uv.y * ld3(L.v2) + (1 - uv.x - uv.y) * ld3(L.v3);
    V3 normal = normalize(uv.x * ld3(L.n1) + uv.y * ld3(L.n2) + (1 - uv.x - uv.y) * ld3(L.n3));
    V3 dir = p - pos;
    nor = normal;
    pdf = 1.f / (L.area * fabs_(dot(normal, normalize(dir)))) * dot(dir, dir);
    if (dot(normal, dir) >= 0.f)
        pdf = 0.f;
    rad = pdf != 0.f ? ld3(L.radiance) : v3(0.f, 0.f, 0.f);
    ray.o = pos;
    ray.d = normalize(dir);
    ray.tmin = eps;
    ray.tmax = sqrt_rn(dot(dir, dir) - eps);
}
__device__ __forceinline__ V3 area_le(const DevLight &L, V3 nor, V3 dir)   // area.h:38-41
{
    if (dot(nor, dir) > 0.f) return ld3(L.radiance);
    return v3(0.f, 0.f, 0.f);
}

__device__ __forceinline__ V3 inf_texel(const DevInfinite &I, int x, int y)   // infinite.h:79-94
{
    int width = I.width, height = I.height;
    float rx = x - (x / width) * width;
    float ry = y - (y / height) * height;
    x = (rx < 0) ? rx + width : rx;
    y = (ry < 0) ? ry + height : ry;
    if (x < 0) x = 0;
    if (x > width - 1) x = width - 1;
    if (y < 0) y = 0;
    if (y > height - 1) y = height - 1;
    const float *c = I.data + 3 * (size_t)(y * width + x);
    return V3{c[0], c[1], c[2]};
}
__device__ __forceinline__ V3 inf_texel_bilinear(const DevInfinite &I, V2 uv)   // infinite.h:66-77
{
    float xx = I.width * uv.x;
    float yy = I.height * uv.y;
    int x = (int)__builtin_floorf(xx);
    int y = (int)__builtin_floorf(yy);
    float dx = fabs_(xx - x);
    float dy = fabs_(yy - y);
    V3 c00 = inf_texel(I, x, y);
    V3 c10 = inf_texel(I, x + 1, y);
    V3 c01 = inf_texel(I, x, y + 1);
    V3 c11 = inf_texel(I, x + 1, y + 1);
    return (1 - dy) * ((1 - dx) * c00 + dx * c10) + dy * ((1 - dx) * c01 + dx * c11);
}
// Infinite::Le (infinite.h:47-59); SampleLight's lookup (:22-36) is the same arithmetic
__device__ __forceinline__ V3 inf_le(const DevInfinite &I, V3 dir)
{
    const V3 iu = ld3(I.u), iv = ld3(I.v), iw = ld3(I.w);
    float costheta = dot(dir, iv);
    float theta = gpt_acosf(costheta);
    V3 d = normalize(dir - costheta * iv);
    float cosphi = dot(d, iu);
    float phi = gpt_acosf(cosphi);
    float c = dot(d, iw);
    phi = c > 0 ? TWOPI - phi : phi;
    float uu = phi / TWOPI;
    float vv = theta / PI;
    return inf_texel_bilinear(I, v2(1.f - uu, vv));
}
__device__ __forceinline__ void inf_sample_light(const DevInfinite &I, V3 pos, V2 uniform, V3 &rad, Ray &ray, V3 &nor,
                                                 float &pdf, float eps)   // infinite.h:17-36
{
    V3 dir = sphere_direction(uniform.x, uniform.y);
    nor = -dir;
    ray.o = pos;
    ray.d = dir;
    ray.tmin = eps;
    ray.tmax = 2.f * I.radius - eps;
    pdf = ONE_OVER_FOUR_PI;
    rad = inf_le(I, dir);
}

// Would Triangle::Intersect (mesh.h:45-67) accept this ray on the emitter's triangle, with tmax = inf?
// Same operations, same order as the traversal's triangle step.
__device__ __forceinline__ bool emitter_accepts(const DevLight &L, V3 o, V3 d, float tmin)
{
    const V3 v1 = ld3(L.v1);
    const V3 e1 = ld3(L.v2) - v1;
    const V3 e2 = ld3(L.v3) - v1;
    const V3 s1 = cross(d, e2);
    const float divisor = dot(s1, e1);
    const float invDivisor = 1.0f / divisor;
    const V3 s = o - v1;
    const float b1 = dot(s, s1) * invDivisor;
    const V3 s2 = cross(s, e1);
    const float b2 = dot(d, s2) * invDivisor;
    const float tt = dot(e2, s2) * invDivisor;
    return !(fabs_(divisor) < 1e-8f) && !(b1 < 0.0f || b1 > 1.0f) && !(b2 < 0.0f || b1 + b2 > 1.0f) &&
           !(tt < tmin || tt > __builtin_inff());
}
constexpr int kEmitterPretestMax = 8;

// pathtracer.cu:172-181 (no match — only for NaN u — returns -1 here; the
// reference falls off the end of a non-void function)
__device__ __forceinline__ int lookup_light_distribution(const DevParams &P, float u, float &pdf)
{
    for (int i = 0; i + 1 < P.n_cdf; ++i) {
        float s = P.light_cdf[i];
        float e = P.light_cdf[i + 1];
        if (u >= s && u <= e) {
            pdf = e - s;
            return i;
        }
    }
    pdf = 0.f;
    return -1;
}
__device__ __forceinline__ float pdf_from_light_distribution(const DevParams &P, int idx)
{
    return P.light_cdf[idx + 1] - P.light_cdf[idx];
}

// ----------------------------------------------------------------- film ------
__device__ __forceinline__ V3 tonemap(V3 in, bool filmic)   // pathtracer.cu:187-204
{
    if (filmic) {
        V3 c = in - v3(0.004f, 0.004f, 0.004f);
        c = V3{fmax_(0.f, c.x), fmax_(0.f, c.y), fmax_(0.f, c.z)};
        c = (c * (6.2f * c + 0.5f)) / (c * (6.2f * c + 1.7f) + 0.06f);
        return c;
    }
    float one_over_gamma = 1.f / 2.2f;
    float exposure = 1.41421356f;
    in = V3{fmax_(in.x, 1e-5f), fmax_(in.y, 1e-5f), fmax_(in.z, 1e-5f)};
    in.x = gpt_powf(in.x * exposure, one_over_gamma);
    in.y = gpt_powf(in.y * exposure, one_over_gamma);
    in.z = gpt_powf(in.z * exposure, one_over_gamma);
    return in;
}

// ------------------------------------------------ participating media -----
// Homogeneous media and the Henyey-Greenstein phase function (src/medium.h:9-51,196-233), as oracle/pt_oracle.c
// restates them (hom_tr, hom_sample, medium_phase, medium_sample_phase).
__device__ __forceinline__ V3 exp3(V3 c) { return V3{gpt_expf(c.x), gpt_expf(c.y), gpt_expf(c.z)}; }       // common.h:81-86
__device__ __forceinline__ V3 hom_tr(const DevMedium &m, float tmax)                                          // medium.h:14-17
{
    return exp3(V3{m.sigmaT[0], m.sigmaT[1], m.sigmaT[2]} * (-tmax));
}
__device__ __forceinline__ V3 hom_sample(const DevMedium &m, float ray_tmax, float u, float &t, bool &sampled)  // medium.h:19-50
{
    const V3 sigmaT = V3{m.sigmaT[0], m.sigmaT[1], m.sigmaT[2]}, sigmaS = V3{m.sigmaS[0], m.sigmaS[1], m.sigmaS[2]};
    float sigma = dot(sigmaT, v3(0.212671f, 0.715160f, 0.072169f));
    float dist = -gpt_logf(u) / sigma;                                  // wrap.h:158-160
    V3 Tr = exp3(sigmaT * (-dist));
    float pdf = sigma * gpt_expf(sigma * -dist);
    bool sampledMedium = dist < ray_tmax;
    sampled = sampledMedium;
    t = dist;
    return sampledMedium ? (Tr * sigmaS) / pdf : (sigmaT * Tr) / pdf;
}
__device__ __forceinline__ float medium_phase(const DevMedium &m, V3 in, V3 out)                               // medium.h:222-233
{
    float g = m.g;
    if (g == 0) return ONE_OVER_FOUR_PI;
    float costheta = dot(in, out);
    float cubicTerm = (1.f + g * g - 2.f * g * costheta);
    return ONE_OVER_FOUR_PI * (1.f - g * g) / sqrt_rn(cubicTerm * cubicTerm * cubicTerm);
}
__device__ __forceinline__ V3 medium_sample_phase(const DevMedium &m, float ux, float uy)                      // medium.h:196-220
{
    float g = m.g;
    if (g == 0) return sphere_direction(ux, uy);
    float costheta;
    if (fabs_(g) < 1e-3f)
        costheta = 1.f - 2.f * ux;
    else {
        float sqrtTerm = (1.f - g * g) / (1.f - g + 2.f * g * ux);
        costheta = (1.f + g * g - sqrtTerm * sqrtTerm) / (2.f * g);
    }
    float sintheta = sqrt_rn(1.f - costheta * costheta);
    float sin_p, cos_p;
    sincos_soft(TWOPI * uy, sin_p, cos_p);
    return polar_y_up(sintheta, costheta, sin_p, cos_p);
}

// Heterogeneous media (src/medium.h:53-182), as oracle/pt_oracle.c restates them (het_d, het_density; het_tr and
// het_sample are the shared tracking loop of the PT_IT_VPT_WALK kernel): a density grid in the box p0..p1, sampled by
// delta tracking; transmittance by delta (0), ratio (1) or residual ratio (2) tracking.  Every loop draws from the
// path's generator and ends after iterMax steps at the latest.
__device__ __forceinline__ int f2i_sat(float f)        // float -> int: truncate, saturate, NaN -> 0 - which is v_cvt_i32_f32
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    asm("v_cvt_i32_f32_e32 %0, %1" : "=v"(r) : "v"(f));      // (a C++ cast leaves the out-of-range cases undefined)
    return r;
#else
    return (int)f;
#endif
}
__device__ __forceinline__ float lerp_(float a, float b, float t) { return a + t * (b - a); }               // cutil_math.h:1008-1011
// getDensity (medium.h:160-174) with d() (:176-181) folded in: the eight corners share six coordinate conversions and
// six range tests (per axis, for psi and psi + 1) instead of doing 24 of each; a corner outside the grid reads as 0.
__device__ __forceinline__ float het_density(const DevMedium &m, V3 p)
{
    const V3 ps = v3(p.x * m.nx, p.y * m.ny, p.z * m.nz);
    const V3 psi = v3(__builtin_floorf(ps.x), __builtin_floorf(ps.y), __builtin_floorf(ps.z));
    const V3 delta = ps - psi;
    const int x0 = f2i_sat(psi.x), x1 = f2i_sat(psi.x + 1), y0 = f2i_sat(psi.y), y1 = f2i_sat(psi.y + 1),
              z0 = f2i_sat(psi.z), z1 = f2i_sat(psi.z + 1);
    // !(x < 0 || x > nx - 1) as one unsigned comparison (nx > 0)
    const bool vx0 = (unsigned)x0 < (unsigned)m.nx, vx1 = (unsigned)x1 < (unsigned)m.nx, vy0 = (unsigned)y0 < (unsigned)m.ny,
               vy1 = (unsigned)y1 < (unsigned)m.ny, vz0 = (unsigned)z0 < (unsigned)m.nz, vz1 = (unsigned)z1 < (unsigned)m.nz;
    const int sy = m.nx, sz = m.ny * m.nx;               // int idx = z*ny*nx + y*nx + x
    const int r00 = z0 * sz + y0 * sy, r10 = z0 * sz + y1 * sy, r01 = z1 * sz + y0 * sy, r11 = z1 * sz + y1 * sy;
    const float *g = m.density;
    // branch-free: a corner outside the grid reads cell 0 and is then replaced by 0
    const bool v000 = vx0 && vy0 && vz0, v100 = vx1 && vy0 && vz0, v010 = vx0 && vy1 && vz0, v110 = vx1 && vy1 && vz0,
               v001 = vx0 && vy0 && vz1, v101 = vx1 && vy0 && vz1, v011 = vx0 && vy1 && vz1, v111 = vx1 && vy1 && vz1;
    const float g000 = g[v000 ? r00 + x0 : 0], g100 = g[v100 ? r00 + x1 : 0];
    const float g010 = g[v010 ? r10 + x0 : 0], g110 = g[v110 ? r10 + x1 : 0];
    const float g001 = g[v001 ? r01 + x0 : 0], g101 = g[v101 ? r01 + x1 : 0];
    const float g011 = g[v011 ? r11 + x0 : 0], g111 = g[v111 ? r11 + x1 : 0];
    const float d000 = v000 ? g000 : 0.f, d100 = v100 ? g100 : 0.f, d010 = v010 ? g010 : 0.f, d110 = v110 ? g110 : 0.f;
    const float d001 = v001 ? g001 : 0.f, d101 = v101 ? g101 : 0.f, d011 = v011 ? g011 : 0.f, d111 = v111 ? g111 : 0.f;
    const float d00 = lerp_(d000, d100, delta.x);
    const float d10 = lerp_(d010, d110, delta.x);
    const float d01 = lerp_(d001, d101, delta.x);
    const float d11 = lerp_(d011, d111, delta.x);
    const float d0 = lerp_(d00, d10, delta.y);
    const float d1 = lerp_(d01, d11, delta.y);
    return lerp_(d0, d1, delta.z);
}
__device__ __forceinline__ V3 het_local(const DevMedium &m, V3 o, V3 d, float dist)      // (r(dist) - p0) / (p1 - p0)
{
    const V3 p0 = V3{m.p0[0], m.p0[1], m.p0[2]}, ext = V3{m.p1[0], m.p1[1], m.p1[2]} - p0;
    const V3 p = (o + d * dist) - p0;
    return v3(p.x / ext.x, p.y / ext.y, p.z / ext.z);
}

// ----------------------------------------------------- the render kernel -----
// normal + light index of a hit, for the MIS light ray (mesh.h:87-90): the
// other Intersection fields are not read at pathtracer.cu:960-976
__device__ __forceinline__ void make_light_hit(const DevParams &P, int prim, float b1, float b2, V3 &nor, int &lightIdx)
{
    const float4 *__restrict__ sp = reinterpret_cast<const float4 *>(P.shade) + 5 * prim;
    const float4 s0 = sp[0], s1 = sp[1];
    const float n3z = reinterpret_cast<const float *>(sp + 2)[0];
    lightIdx = __float_as_int(reinterpret_cast<const float *>(sp + 4)[3]);
    const V3 n1 = V3{s0.x, s0.y, s0.z}, n2 = V3{s0.w, s1.x, s1.y}, n3 = V3{s1.z, s1.w, n3z};
    nor = normalize(n1 * (1.f - b1 - b2) + n2 * b1 + n3 * b2);
}

}  // namespace pt
