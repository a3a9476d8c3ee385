// pt_vec.h — float2/float3 value types with the operator semantics the
// reference's math header gives CUDA's float2/float3 (reference
// src/cutil_math.h: component-wise + - * /, scalar broadcast, dot :1126,
// cross :1298, length :1169, normalize = v * rsqrt(dot) :1187, clamp =
// fmaxf(a, fminf(f, b)) :1030).  Re-derived here, not copied: only the
// arithmetic contract matters, and it is what keeps expression evaluation order
// identical to the reference when its formulas are written with operators.
//
// Shared by host code (scene packing, BVH build) and the HIP kernel; every
// translation unit using it is compiled with -ffp-contract=off.
#pragma once

#include <stdint.h>
#include "../../include/gpt_softmath.h"

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define PT_HD inline __attribute__((always_inline))
#endif

namespace pt {

// truncated-literal constants, reference src/common.h:22-27
constexpr float PI = 3.14159265358f;
constexpr float TWOPI = 6.28318530716f;
constexpr float FOURPI = 12.56637061432f;
constexpr float ONE_OVER_PI = 0.3183098861847f;
constexpr float ONE_OVER_TWO_PI = 0.1591549430923f;
constexpr float ONE_OVER_FOUR_PI = 0.0795774715461f;

struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct V4 { float x, y, z, w; };

PT_HD V2 v2(float x, float y) { return V2{x, y}; }
PT_HD V3 v3(float x, float y, float z) { return V3{x, y, z}; }
PT_HD V3 v3(float s) { return V3{s, s, s}; }
PT_HD V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }

PT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
PT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
PT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
PT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
PT_HD V3 operator/(V3 a, V3 b) { return V3{a.x / b.x, a.y / b.y, a.z / b.z}; }
PT_HD V3 operator+(V3 a, float b) { return V3{a.x + b, a.y + b, a.z + b}; }
PT_HD V3 operator-(V3 a, float b) { return V3{a.x - b, a.y - b, a.z - b}; }
PT_HD V3 operator*(V3 a, float b) { return V3{a.x * b, a.y * b, a.z * b}; }
PT_HD V3 operator*(float b, V3 a) { return V3{b * a.x, b * a.y, b * a.z}; }
PT_HD V3 operator/(V3 a, float b) { return V3{a.x / b, a.y / b, a.z / b}; }
PT_HD void operator+=(V3 &a, V3 b) { a.x += b.x; a.y += b.y; a.z += b.z; }
PT_HD void operator*=(V3 &a, V3 b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; }
PT_HD void operator*=(V3 &a, float b) { a.x *= b; a.y *= b; a.z *= b; }
PT_HD void operator/=(V3 &a, float b) { a.x /= b; a.y /= b; a.z /= b; }

PT_HD V2 operator+(V2 a, V2 b) { return V2{a.x + b.x, a.y + b.y}; }
PT_HD V2 operator-(V2 a, V2 b) { return V2{a.x - b.x, a.y - b.y}; }
PT_HD V2 operator*(V2 a, float b) { return V2{a.x * b, a.y * b}; }
PT_HD V2 operator*(float b, V2 a) { return V2{b * a.x, b * a.y}; }

PT_HD V4 operator+(V4 a, V4 b) { return V4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
PT_HD V4 operator*(float b, V4 a) { return V4{b * a.x, b * a.y, b * a.z, b * a.w}; }

PT_HD float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_HD V3 cross(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// IEEE sqrt and divide, correctly rounded on both targets
// (-fhip-fp32-correctly-rounded-divide-sqrt on the device).
PT_HD float sqrt_rn(float x) { return __builtin_sqrtf(x); }
// the reference's host rsqrtf (cutil_math.h:55-58); the device intrinsic is not reproducible
PT_HD float rsqrt_rn(float x) { return 1.0f / sqrt_rn(x); }
PT_HD float length(V3 v) { return sqrt_rn(dot(v, v)); }
PT_HD V3 normalize(V3 v) { float invLen = rsqrt_rn(dot(v, v)); return v * invLen; }

// minNum / maxNum (a NaN operand loses), like CUDA's fminf/fmaxf.  On the device this is one v_min_f32 /
// v_max_f32; on the host the builtin would be a libm call, so the same rule is spelled out inline.
#if defined(__HIP_DEVICE_COMPILE__)
PT_HD float fmin_(float a, float b) { return __builtin_fminf(a, b); }
PT_HD float fmax_(float a, float b) { return __builtin_fmaxf(a, b); }
#else
PT_HD float fmin_(float a, float b) { return gpt_fminf(a, b); }
PT_HD float fmax_(float a, float b) { return gpt_fmaxf(a, b); }
#endif
PT_HD float clamp(float f, float a, float b) { return fmax_(a, fmin_(f, b)); }
PT_HD float fabs_(float x) { return __builtin_fabsf(x); }
PT_HD bool is_black(V3 c) { return c.x == 0 && c.y == 0 && c.z == 0; }
PT_HD bool is_nan(V3 c) { return gpt_isnanf(c.x) || gpt_isnanf(c.y) || gpt_isnanf(c.z); }
PT_HD bool is_inf(V3 c) { return gpt_isinff(c.x) || gpt_isinff(c.y) || gpt_isinff(c.z); }

}  // namespace pt
