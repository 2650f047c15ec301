// tbrm_device_math.h — gfx950 device-side scalar arithmetic of the raymarch + illumination path.
//
// This header IS the "arithmetic spec" of DESIGN.md in code form: fp32, no contraction (the translation
// unit is compiled with -ffp-contract=off), fused multiply-add only where spelled __builtin_fmaf, IEEE
// division/sqrt (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt). The CPU oracle restates the
// same sequence independently; the UNORM8 light volume is compared bit-for-bit against it.
//
// Reference functions restated here (paths relative to Source/Raymarcher/Shaders/Private/):
//   GetTransferFuncPosition / SampleWindowedTransferFunction   WindowedSampling.usf:14-37
//   IsCurPosClipped / GetUVW                                   RaymarcherCommon.usf:22-43
//   AccumulateLightEnergy                                      RaymarchMaterialCommon.usf:82-88
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tbrm {

__device__ __forceinline__ float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ float lerp_(float a, float b, float f) { return fma_(f, b - a, a); }
__device__ __forceinline__ float saturate_(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); } // NaN -> 0

// ---- pow(x,y) = exp2(y*log2(x)); polynomial coefficients from tools/gen_pow_coeffs.py -----------------
// HLSL pow (WindowedSampling.usf:35) has no bit-level definition; v_log_f32/v_exp_f32 are not reproducible
// on a CPU, so the build defines pow through these two polynomials (max rel. error 2.7e-8 / 5.2e-9 before
// fp32 rounding).
__device__ __forceinline__ float log2_poly(float x) // x normal, > 0
{
    const uint32_t ix = __float_as_uint(x);
    int e = (int) (ix >> 23) - 127;
    float m = __uint_as_float((ix & 0x007fffffu) | 0x3f800000u);
    if (m >= 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; }
    const float r = m - 1.0f;
    float q = 0x1.025a2p-3f;
    q = fma_(q, r, -0x1.a8cc5cp-3f);
    q = fma_(q, r, 0x1.b9b11ep-3f);
    q = fma_(q, r, -0x1.e94f12p-3f);
    q = fma_(q, r, 0x1.26d41p-2f);
    q = fma_(q, r, -0x1.715c9cp-2f);
    q = fma_(q, r, 0x1.ec73d4p-2f);
    q = fma_(q, r, -0x1.71547p-1f);
    q = fma_(q, r, 0x1.715476p+0f);
    return fma_(r, q, (float) e);
}

__device__ __forceinline__ float exp2_poly(float p)
{
    if (!(p >= -150.0f)) return (p != p) ? p : 0.0f;
    if (p >= 128.0f) return __builtin_inff();
    const float n = floorf(p + 0.5f);
    const float g = p - n;
    float r = 0x1.444004p-13f;
    r = fma_(r, g, 0x1.5f0896p-10f);
    r = fma_(r, g, 0x1.3b2a1cp-7f);
    r = fma_(r, g, 0x1.c6af6cp-5f);
    r = fma_(r, g, 0x1.ebfbep-3f);
    r = fma_(r, g, 0x1.62e43p-1f);
    float v = fma_(g, r, 1.0f);
    int ni = (int) n;
    if (ni > 127) { v = v * 2.0f; ni -= 1; }
    if (ni < -126) { v = v * 0x1p-64f; ni += 64; }
    return v * __uint_as_float((uint32_t) (ni + 127) << 23);
}

// exp2_poly(p) for p <= 0 (or NaN), bit for bit: the results' exponent is never positive, so the overflow branch and the
// 2^128 rescue drop out (the opacity correction raises 1 - a in [0, 1] to a step size >= 0: its exponent p = y log2 x is never positive)
__device__ __forceinline__ float exp2_poly_nonpos(float p)
{
    if (!(p >= -150.0f)) return (p != p) ? p : 0.0f;
    const float n = floorf(p + 0.5f);
    const float g = p - n;
    float r = 0x1.444004p-13f;
    r = fma_(r, g, 0x1.5f0896p-10f);
    r = fma_(r, g, 0x1.3b2a1cp-7f);
    r = fma_(r, g, 0x1.c6af6cp-5f);
    r = fma_(r, g, 0x1.ebfbep-3f);
    r = fma_(r, g, 0x1.62e43p-1f);
    float v = fma_(g, r, 1.0f);
    int ni = (int) n;
    if (ni < -126) { v = v * 0x1p-64f; ni += 64; }
    return v * __uint_as_float((uint32_t) (ni + 127) << 23);
}

// pow_(x, y) for x in [0, 1] and y >= 0 (both NaN-free by construction: x = 1 - saturate(a), y = a step size the host checked), bit
// for bit
__device__ __forceinline__ float pow01_(float x, float y)
{
    if (!(x >= 0x1p-126f)) return (y > 0.0f) ? 0.0f : 1.0f; // (x == 0: pow_'s first branch with y >= 0)
    return exp2_poly_nonpos(y * log2_poly(x));
}
__device__ __forceinline__ void pow01_2_(float x, float y0, float y1, float& r0, float& r1)
{
    if (!(x >= 0x1p-126f)) { r0 = (y0 > 0.0f) ? 0.0f : 1.0f; r1 = (y1 > 0.0f) ? 0.0f : 1.0f; return; }
    const float l = log2_poly(x);
    r0 = exp2_poly_nonpos(y0 * l);
    r1 = exp2_poly_nonpos(y1 * l);
}

// 1 - exp2_poly_nonpos(p), bit for bit, for p <= 0 (-inf allowed, NaN not: pow01_'s domain). The opacity correction only ever uses
// 1 - pow: a power r <= 2^-25 is gone in 1 - r (RNE: 1), so everything exp2_poly_nonpos does for tiny results — the p < -150 test,
// the two-step scaling into the denormals — cannot show. Clamped at p = -100 the scaling 2^n v is exact with v in [0.70, 1.42] and
// n in [-100, 0]: an addition to the exponent field. 14 vector instructions instead of 24 and no branch; compared with the long form
// over every float x in [0, 1] on the device (tbrm_selftest_opacity_correction).
__device__ __forceinline__ float one_minus_exp2_nonpos(float p)
{
    p = fmaxf(p, -100.0f);
    const float n = floorf(p + 0.5f);
    const float g = p - n;
    float r = 0x1.444004p-13f;
    r = fma_(r, g, 0x1.5f0896p-10f);
    r = fma_(r, g, 0x1.3b2a1cp-7f);
    r = fma_(r, g, 0x1.c6af6cp-5f);
    r = fma_(r, g, 0x1.ebfbep-3f);
    r = fma_(r, g, 0x1.62e43p-1f);
    const float v = fma_(g, r, 1.0f);
    return 1.0f - __uint_as_float(__float_as_uint(v) + ((uint32_t) (int) n << 23));
}
// 1 - pow01_(x, y) and the pair 1 - pow01_(x, y0), 1 - pow01_(x, y1) (one logarithm), bit for bit
__device__ __forceinline__ float one_minus_pow01_(float x, float y)
{
    if (!(x >= 0x1p-126f)) return (y > 0.0f) ? 1.0f : 0.0f;
    return one_minus_exp2_nonpos(y * log2_poly(x));
}
__device__ __forceinline__ void one_minus_pow01_2_(float x, float y0, float y1, float& r0, float& r1)
{
    if (!(x >= 0x1p-126f)) { r0 = (y0 > 0.0f) ? 1.0f : 0.0f; r1 = (y1 > 0.0f) ? 1.0f : 0.0f; return; }
    const float l = log2_poly(x);
    r0 = one_minus_exp2_nonpos(y0 * l);
    r1 = one_minus_exp2_nonpos(y1 * l);
}

__device__ __forceinline__ float pow_(float x, float y)
{
    if (!(x >= 0x1p-126f)) {
        if (x != x) return x;
        return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : __builtin_inff());
    }
    if (x == __builtin_inff()) return (y > 0.0f) ? x : ((y == 0.0f) ? 1.0f : 0.0f);
    return exp2_poly(y * log2_poly(x));
}

// pow_(x, y0) and pow_(x, y1), bit for bit, with the logarithm computed once (the two axis passes of a light correct the same
// opacity for two step sizes: k_light_occlusion<..., DUAL>)
__device__ __forceinline__ void pow2_(float x, float y0, float y1, float& r0, float& r1)
{
    if (!(x >= 0x1p-126f) || x == __builtin_inff()) { r0 = pow_(x, y0); r1 = pow_(x, y1); return; }
    const float l = log2_poly(x);
    r0 = exp2_poly(y0 * l);
    r1 = exp2_poly(y1 * l);
}

// ---- UNORM conversion (D3D11 functional spec: load c/(2^n-1); store trunc(clamp(x,0,1)*255+0.5), NaN->0)
// Load: c/d as fma(c, r, c*r2) with r = RN(1/d) and r2 = RN(1/d - r), the reciprocal split in two floats: 2 instructions
// instead of the ~10 of a correctly rounded v_div sequence, and bit-identical to IEEE c/d for every UNORM8 and UNORM16
// code (checked exhaustively on the device: tests/test_gpu_properties.py::test_unorm_decode_is_exact_division, and in exact
// rational arithmetic: tests/test_host_math.py::test_split_reciprocal_decode_is_exact).
__device__ __forceinline__ float decode_u8f(float c) { return fma_(c, 0x1.010102p-8f, c * -0x1.fdfdfep-33f); }
__device__ __forceinline__ float decode_u16f(float c) { return fma_(c, 0x1.0001p-16f, c * 0x1.0001p-48f); }
__device__ __forceinline__ float decode_u8(uint32_t c) { return decode_u8f((float) c); }
__device__ __forceinline__ float decode_u16(uint32_t c) { return decode_u16f((float) c); }
// Store: the code as a float. v_med3_f32 returns the minimum of the non-NaN operands when an operand is NaN, i.e. 0 —
// the D3D rule — so the clamp needs no separate NaN test; x*255+0.5 is >= 0.5, so floor == the spec's truncation.
__device__ __forceinline__ float quantize_u8(float x)
{
    return __builtin_floorf(__builtin_amdgcn_fmed3f(x, 0.0f, 1.0f) * 255.0f + 0.5f);
}
__device__ __forceinline__ uint32_t encode_u8(float x) { return (uint32_t) quantize_u8(x); }

// Texel split of a normalised coordinate: floor/frac of u*N - 0.5 (clamped to +-2^30 so the conversion is defined).
__device__ __forceinline__ void texel_split(float u, float n, int& i0, float& f)
{
    float x = u * n - 0.5f;
    x = fminf(fmaxf(x, -0x1p30f), 0x1p30f);
    const float fl = floorf(x);
    i0 = (int) fl;
    f = x - fl;
}

// texel_split for a coordinate the caller knows to be finite and within +-2^30 texels (a sample position within a step of the unit
// cube, a saturated coordinate): the clamp is dead there — the same bits without it
__device__ __forceinline__ void texel_split_bounded(float u, float n, int& i0, float& f)
{
    const float x = u * n - 0.5f;
    const float fl = floorf(x);
    i0 = (int) fl;
    f = x - fl;
}

__device__ __forceinline__ int wrap_index(int i, int n)
{
    i %= n;
    return i < 0 ? i + n : i;
}
__device__ __forceinline__ int clamp_index(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

// GetTransferFuncPosition (WindowedSampling.usf:14-17)
__device__ __forceinline__ float tf_position(float value, float center, float width)
{
    return (value - center + (width / 2.0f)) / width;
}

// The same quotient, correctly rounded, in three instructions instead of the ~10 of the IEEE division sequence (two of them
// quarter-rate): q = RN(a y) with y = RN(1 / width) from the host is within an ulp of a / width, the remainder a - q width is exact in
// an fma, and one correction step rounds to the nearest (Markstein's theorem; it needs y correctly rounded — the host's division —,
// a significand of `width` that is not all ones, and no over- / underflow on the way: the host checks all three and the operand range —
// UNORM data, |center|, |width| within 2^+-40 — before it sets WindowDev::fast_div; tbrm_selftest_window_division compares the two
// forms over every float in [0, 1] on the device).
__device__ __forceinline__ float tf_position_fast(float value, float center, float width, float inv_width)
{
    const float a = value - center + (width / 2.0f);
    const float q = a * inv_width;
    const float r = fma_(-q, width, a);
    return fma_(r, inv_width, q);
}

// IsCurPosClipped (RaymarcherCommon.usf:22-25)
__device__ __forceinline__ bool is_clipped(float px, float py, float pz, const float* cc, const float* cd)
{
    return (((px - cc[0]) * cd[0] + (py - cc[1]) * cd[1]) + (pz - cc[2]) * cd[2]) <= 0.0f;
}

} // namespace tbrm
