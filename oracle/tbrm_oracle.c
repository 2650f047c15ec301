/*
 * tbrm_oracle.c — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's raymarch + illumination hot path, written
 * line-by-line after the reference HLSL / C++ it cites. It is the checker for the HIP kernels: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it. Nothing under
 * tbraymarcherplugin_amd/ includes, links or calls it; the product has no CPU path.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, numeric tests or CPU implementation of this path
 * (SURVEY.md §4, §8c), and its arithmetic cannot be compiled here (HLSL .usf against Unreal Engine 5.4
 * includes + D3D11 fixed-function samplers; no HLSL compiler, no engine). This file is therefore pinned
 * only by (i) analytic known-answer tests (tests/test_oracle_kat.py) and (ii) data recovered from the
 * reference's own assets (tests/golden/tf_curves.json). Engine behaviour that lives outside
 * /root/reference (Unreal Engine 5.4.0 per TBRaymarcherPlugin.uplugin:5 — texture filtering, UNORM
 * conversion, border colours, FFloat16, Rand3DPCG16, FTransform) is restated from its published
 * definition; each such place says so.
 *
 * Arithmetic contract (shared with the HIP kernels, DESIGN.md "Arithmetic spec"): fp32 throughout the
 * device part, no contraction (-ffp-contract=off), fused multiply-add only where written as fmaf();
 * IEEE division and sqrt; pow() = exp2(y*log2(x)) through the polynomials below. Host parameter math is
 * double (FVector is double in UE5) narrowed to float where the reference binds shader parameters.
 */
#include "tbrm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ================================================================================================ */
/* scalar helpers                                                                                    */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* HLSL saturate(): clamp to [0,1], NaN -> 0. */
static inline float saturatef(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
static inline float lerpf(float a, float b, float f) { return fmaf(f, b - a, a); }

/* ---- S6: pow(x,y) = exp2(y*log2(x)). HLSL `pow` (WindowedSampling.usf:35) is an engine/driver intrinsic
 * with no bit-level definition; this polynomial form is the build's definition, evaluated identically on
 * CPU and GPU so that the UNORM8 light volume can be compared bit-for-bit. Coefficients:
 * tools/gen_pow_coeffs.py (log2(1+r)/r degree 8 on [sqrt(1/2)-1, sqrt(2)-1]; (2^g-1)/g degree 5 on
 * [-1/2,1/2]). */
static const float ORC_LOG2_Q[9] = {
    0x1.715476p+0f, -0x1.71547p-1f, 0x1.ec73d4p-2f, -0x1.715c9cp-2f, 0x1.26d41p-2f,
    -0x1.e94f12p-3f, 0x1.b9b11ep-3f, -0x1.a8cc5cp-3f, 0x1.025a2p-3f};
static const float ORC_EXP2_R[6] = {
    0x1.62e43p-1f, 0x1.ebfbep-3f, 0x1.c6af6cp-5f, 0x1.3b2a1cp-7f, 0x1.5f0896p-10f, 0x1.444004p-13f};

float orc_log2f(float x) /* x normal and > 0 */
{
    uint32_t ix = f2u(x);
    int e = (int) (ix >> 23) - 127;
    float m = u2f((ix & 0x007fffffu) | 0x3f800000u); /* [1,2) */
    if (m >= 0x1.6a09e6p+0f) { m = m * 0.5f; e += 1; } /* -> [sqrt(1/2), sqrt(2)) */
    const float r = m - 1.0f;
    float q = ORC_LOG2_Q[8];
    for (int i = 7; i >= 0; --i) q = fmaf(q, r, ORC_LOG2_Q[i]);
    return fmaf(r, q, (float) e);
}

float orc_exp2f(float p)
{
    if (!(p >= -150.0f)) return (p != p) ? p : 0.0f; /* NaN passes; underflow -> 0 */
    if (p >= 128.0f) return INFINITY;
    const float n = floorf(p + 0.5f);
    const float g = p - n;
    float r = ORC_EXP2_R[5];
    for (int i = 4; i >= 0; --i) r = fmaf(r, g, ORC_EXP2_R[i]);
    float v = fmaf(g, r, 1.0f);
    int ni = (int) n;
    if (ni > 127) { v = v * 2.0f; ni -= 1; }
    if (ni < -126) { v = v * 0x1p-64f; ni += 64; }
    return v * u2f((uint32_t) (ni + 127) << 23);
}

float orc_powf(float x, float y)
{
    if (!(x >= 0x1p-126f)) { /* zero, denormal, negative or NaN base: treated as 0 (x = 1 - saturate(a) >= 0) */
        if (x != x) return x;
        return (y > 0.0f) ? 0.0f : ((y == 0.0f) ? 1.0f : INFINITY);
    }
    if (x == INFINITY) return (y > 0.0f) ? INFINITY : ((y == 0.0f) ? 1.0f : 0.0f);
    return orc_exp2f(y * orc_log2f(x));
}

/* ---- FFloat16 (RaymarchUtils.cpp:151-161 stores TF samples as FFloat16). Engine type; restated as IEEE
 * binary16 round-to-nearest-even, the definition numpy.float16 uses. */
static uint16_t float_to_half_rne(float f)
{
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t mant = x & 0x007fffffu;
    int exp = (int) ((x >> 23) & 0xff);
    if (exp == 0xff) return (uint16_t) (sign | 0x7c00u | (mant ? 0x200u | (mant >> 13) : 0));
    int e = exp - 127 + 15;
    if (e >= 31) return (uint16_t) (sign | 0x7c00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t) sign;
        mant |= 0x00800000u;
        int shift = 14 - e;
        uint32_t h = mant >> shift;
        uint32_t rem = mant & ((1u << shift) - 1);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1))) h++;
        return (uint16_t) (sign | h);
    }
    uint32_t h = ((uint32_t) e << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;
    return (uint16_t) (sign | h);
}

static float half_to_float(uint16_t h)
{
    uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
    int e = (h >> 10) & 0x1f;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float) m * 0x1p-24f;
        return sign ? -v : v;
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((uint32_t) (e - 15 + 127) << 23) | (m << 13));
}

float orc_round_to_half(float f) { return half_to_float(float_to_half_rne(f)); }

/* ================================================================================================ */
/* TF construction: URaymarchUtils::ColorCurveToTexture (RaymarchUtils.cpp:143-174) and                */
/* MakeDefaultTFTexture (:113-141). The 16 identical rows collapse to one (bilinear at v=0.5 between    */
/* two identical rows returns the row exactly).                                                        */

void orc_bake_tf(const float* rgba_256x4, float* out_rgba_256x4)
{
    for (int i = 0; i < 256 * 4; ++i) out_rgba_256x4[i] = orc_round_to_half(rgba_256x4[i]);
}

/* UCurveLinearColor::GetLinearColorValue for curves whose keys are all RCIM_Linear (every shipped
 * Content/Curves/TF_CT-*.uasset): per channel, linear interpolation between
 * neighbouring keys, constant before the first / after the last key. Engine code; restated. */
static float eval_linear_curve(const float* t, const float* v, int n, float x)
{
    if (n <= 0) return 0.0f;
    if (x <= t[0]) return v[0];
    if (x >= t[n - 1]) return v[n - 1];
    int k = 1;
    while (k < n - 1 && x >= t[k]) ++k;
    const float t0 = t[k - 1], t1 = t[k];
    const float diff = t1 - t0;
    if (!(diff > 0.0f)) return v[k - 1];
    const float alpha = (x - t0) / diff;
    return v[k - 1] + alpha * (v[k] - v[k - 1]); /* FMath::Lerp(A,B,Alpha) = A + Alpha*(B-A) */
}

void orc_color_curve_to_lut(const float* const key_times[4], const float* const key_values[4],
                            const int32_t n_keys[4], float* out_rgba_256x4)
{
    for (unsigned i = 0; i < 256; ++i) {
        float index = ((float) i) / ((float) 256 - 1); /* RaymarchUtils.cpp:155 */
        for (int c = 0; c < 4; ++c)
            out_rgba_256x4[i * 4 + c] = eval_linear_curve(key_times[c], key_values[c], n_keys[c], index);
    }
}

void orc_make_default_tf_lut(float* out_rgba_256x4)
{
    for (unsigned i = 0; i < 256; ++i) {
        float w = (float) i / (float) (256 - 1); /* RaymarchUtils.cpp:123 */
        out_rgba_256x4[i * 4 + 0] = w;
        out_rgba_256x4[i * 4 + 1] = w;
        out_rgba_256x4[i * 4 + 2] = w;
        out_rgba_256x4[i * 4 + 3] = 1.0f;
    }
}

/* ================================================================================================ */
/* texture sampling (D3D11 fixed function; engine/driver behaviour outside the reference, restated from  */
/* the D3D11 functional spec: texel centres at (i+0.5)/N, linear filter weights from frac(u*N-0.5),     */
/* address mode applied per tap, UNORM decode c/(2^n-1), UNORM8 store trunc(clamp(x,0,1)*255+0.5)).     */

enum { ADDR_WRAP = 0, ADDR_CLAMP = 1, ADDR_BORDER = 2 };

static inline float decode_voxel(const void* data, int fmt, size_t idx)
{
    switch (fmt) {
        case TBRM_FMT_G8: return (float) ((const uint8_t*) data)[idx] / 255.0f;
        case TBRM_FMT_G16: return (float) ((const uint16_t*) data)[idx] / 65535.0f;
        default: return ((const float*) data)[idx];
    }
}

static inline uint8_t encode_unorm8(float x)
{
    if (x != x) return 0;
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    return (uint8_t) (x * 255.0f + 0.5f);
}

static inline int addr_index(int i, int n, int mode, int* oob)
{
    if (mode == ADDR_WRAP) {
        i %= n;
        if (i < 0) i += n;
        return i;
    }
    if (mode == ADDR_CLAMP) return i < 0 ? 0 : (i >= n ? n - 1 : i);
    if (i < 0 || i >= n) *oob = 1;
    return i;
}

/* floor + frac of a texel coordinate; coordinates are clamped to +-2^30 so the int conversion is defined. */
static inline void texel_split(float x, int* i0, float* f)
{
    x = fminf(fmaxf(x, -0x1p30f), 0x1p30f);
    const float fl = floorf(x);
    *i0 = (int) fl;
    *f = x - fl;
}

static float sample_volume_trilinear(const orc_volume_view* vol, float u, float v, float w, int mode, float border)
{
    int ix, iy, iz;
    float fx, fy, fz;
    texel_split(u * (float) vol->dim_x - 0.5f, &ix, &fx);
    texel_split(v * (float) vol->dim_y - 0.5f, &iy, &fy);
    texel_split(w * (float) vol->dim_z - 0.5f, &iz, &fz);
    float t[2][2][2];
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx) {
                int oob = 0;
                const int x = addr_index(ix + dx, vol->dim_x, mode, &oob);
                const int y = addr_index(iy + dy, vol->dim_y, mode, &oob);
                const int z = addr_index(iz + dz, vol->dim_z, mode, &oob);
                t[dz][dy][dx] = oob ? border
                                    : decode_voxel(vol->data, vol->format,
                                          ((size_t) z * vol->dim_y + y) * (size_t) vol->dim_x + x);
            }
    const float c00 = lerpf(t[0][0][0], t[0][0][1], fx);
    const float c10 = lerpf(t[0][1][0], t[0][1][1], fx);
    const float c01 = lerpf(t[1][0][0], t[1][0][1], fx);
    const float c11 = lerpf(t[1][1][0], t[1][1][1], fx);
    const float c0 = lerpf(c00, c10, fy);
    const float c1 = lerpf(c01, c11, fy);
    return lerpf(c0, c1, fz);
}

/* Bilinear, border-addressed fetch from a 2D read/write buffer (ReadBuffer.SampleLevel, AddDirLightShader.usf:82). */
static float sample_buffer_bilinear(const void* buf, int fmt, int w, int h, float u, float v, float border)
{
    int ix, iy;
    float fx, fy;
    texel_split(u * (float) w - 0.5f, &ix, &fx);
    texel_split(v * (float) h - 0.5f, &iy, &fy);
    float t[2][2];
    for (int dy = 0; dy < 2; ++dy)
        for (int dx = 0; dx < 2; ++dx) {
            const int x = ix + dx, y = iy + dy;
            t[dy][dx] = (x < 0 || x >= w || y < 0 || y >= h) ? border : decode_voxel(buf, fmt, (size_t) y * w + x);
        }
    const float c0 = lerpf(t[0][0], t[0][1], fx);
    const float c1 = lerpf(t[1][0], t[1][1], fx);
    return lerpf(c0, c1, fy);
}

/* TF.SampleLevel(clamp-bilinear, float2(TFPos, 0.5), 0) on the 256 x 16 RGBA16F texture (WindowedSampling.usf:33). */
static void sample_tf(const float* tf, float pos, float out[4])
{
    int i0;
    float f;
    texel_split(pos * 256.0f - 0.5f, &i0, &f);
    int i1 = i0 + 1;
    i0 = i0 < 0 ? 0 : (i0 > 255 ? 255 : i0);
    i1 = i1 < 0 ? 0 : (i1 > 255 ? 255 : i1);
    for (int c = 0; c < 4; ++c) out[c] = lerpf(tf[i0 * 4 + c], tf[i1 * 4 + c], f);
}

/* ================================================================================================ */
/* A1-A3: windowed sampling (WindowedSampling.usf:14-44)                                              */

static inline float get_transfer_func_position(float value, float center, float width)
{
    return (value - center + (width / 2.0f)) / width; /* WindowedSampling.usf:16 */
}

/* Diagnostics only (tools/exit_flip_rate.py): the corrected opacity of every sample times this factor. 1 (the default) leaves
 * the arithmetic untouched; 1 + 2^-23 is the smallest perturbation an approximate opacity path could introduce. */
static float g_debug_opacity_scale = 1.0f;
void orc_debug_set_opacity_scale(float s) { g_debug_opacity_scale = s; }

static void sample_windowed_transfer_function(float value, float step_size, const float* tf,
                                              const tbrm_windowing_params* wp, float out[4])
{
    const float tfpos = get_transfer_func_position(value, wp->center, wp->width);
    const float wz = wp->low_cutoff ? 1.0f : 0.0f, ww = wp->high_cutoff ? 1.0f : 0.0f; /* VolumeInfo.h:49-52 */
    if ((tfpos < 0.0f && wz > 0.0f) || (tfpos > 1.0f && ww > 0.0f)) { /* WindowedSampling.usf:28 */
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        return;
    }
    sample_tf(tf, tfpos, out);
    out[3] = saturatef(out[3]);
    out[3] = 1.0f - orc_powf(1.0f - out[3], step_size); /* :35 */
    if (g_debug_opacity_scale != 1.0f) out[3] = out[3] * g_debug_opacity_scale;
}

/* ================================================================================================ */
/* host math (double): FTransform / FVector helpers (engine types; restated from their definitions)   */

typedef struct { double x, y, z; } v3;

static v3 v3_make(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static v3 v3_from(tbrm_vec3d a) { return v3_make(a.x, a.y, a.z); }
static v3 v3_cross(v3 a, v3 b) { return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static v3 v3_scale(v3 a, double s) { return v3_make(a.x * s, a.y * s, a.z * s); }
static v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static double v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static v3 v3_neg(v3 a) { return v3_make(-a.x, -a.y, -a.z); }

/* FVector::Normalize(Tolerance = SMALL_NUMBER): scale by 1/sqrt(|v|^2) when |v|^2 > 1e-8, else unchanged. */
static v3 v3_normalize(v3 a)
{
    const double ss = a.x * a.x + a.y * a.y + a.z * a.z;
    if (ss > 1.e-8) return v3_scale(a, 1.0 / sqrt(ss));
    return a;
}

/* FQuat::RotateVector: V + W*T + cross(Q.xyz, T) with T = 2*cross(Q.xyz, V). */
static v3 quat_rotate(tbrm_quatd q, v3 v)
{
    const v3 qv = v3_make(q.x, q.y, q.z);
    const v3 t = v3_scale(v3_cross(qv, v), 2.0);
    return v3_add(v3_add(v, v3_scale(t, q.w)), v3_cross(qv, t));
}
static v3 quat_unrotate(tbrm_quatd q, v3 v)
{
    tbrm_quatd qi = {-q.x, -q.y, -q.z, q.w};
    return quat_rotate(qi, v);
}
/* FTransform::GetSafeScaleReciprocal(Scale, SMALL_NUMBER). */
static v3 safe_scale_reciprocal(tbrm_vec3d s)
{
    return v3_make(fabs(s.x) <= 1.e-8 ? 0.0 : 1.0 / s.x, fabs(s.y) <= 1.e-8 ? 0.0 : 1.0 / s.y,
        fabs(s.z) <= 1.e-8 ? 0.0 : 1.0 / s.z);
}
static v3 inverse_transform_vector(const tbrm_transform* t, v3 v)
{
    return v3_mul(quat_unrotate(t->rotation, v), safe_scale_reciprocal(t->scale3d));
}
static v3 inverse_transform_vector_no_scale(const tbrm_transform* t, v3 v) { return quat_unrotate(t->rotation, v); }
static v3 inverse_transform_position(const tbrm_transform* t, v3 p)
{
    return v3_mul(quat_unrotate(t->rotation, v3_sub(p, v3_from(t->translation))), safe_scale_reciprocal(t->scale3d));
}

/* WorldToLocal of the cube mesh component as the row-vector matrix the material reads through
 * GetPrimitiveData().WorldToLocal (RaymarchMaterialCommon.usf:35,47-48): local = world_row * M. */
void orc_world_to_local(const tbrm_transform* t, float out_m[12])
{
    const v3 rs = safe_scale_reciprocal(t->scale3d);
    const v3 e[3] = {v3_make(1, 0, 0), v3_make(0, 1, 0), v3_make(0, 0, 1)};
    for (int r = 0; r < 3; ++r) {
        const v3 row = v3_mul(quat_unrotate(t->rotation, e[r]), rs);
        out_m[r * 3 + 0] = (float) row.x;
        out_m[r * 3 + 1] = (float) row.y;
        out_m[r * 3 + 2] = (float) row.z;
    }
    const v3 tr = inverse_transform_position(t, v3_make(0, 0, 0));
    out_m[9] = (float) tr.x;
    out_m[10] = (float) tr.y;
    out_m[11] = (float) tr.z;
}

/* GetLocalClippingParameters (LightingShaderUtils.cpp:205-220), then FVector3f() as bound (LightingShaders.h:100-101). */
void orc_local_clipping(const tbrm_world_params* world, float out_center[3], float out_dir[3])
{
    const v3 c = v3_add(inverse_transform_position(&world->volume_transform, v3_from(world->clipping_plane.center)),
        v3_make(0.5, 0.5, 0.5));
    v3 d = inverse_transform_vector_no_scale(&world->volume_transform, v3_from(world->clipping_plane.direction));
    d = v3_mul(d, v3_from(world->volume_transform.scale3d));
    d = v3_normalize(d);
    out_center[0] = (float) c.x; out_center[1] = (float) c.y; out_center[2] = (float) c.z;
    out_dir[0] = (float) d.x; out_dir[1] = (float) d.y; out_dir[2] = (float) d.z;
}

/* FLinearColor -> FColor -> sampler border colour. Engine code (Color.cpp); restated: sRGB transfer
 * functions of IEC 61966-2-1, 8-bit quantisation round-half-up, decoded back to float when the sampler
 * is created. */
static float srgb8_round_trip(float linear)
{
    double v = linear;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    if (v != v) v = 0.0;
    const double enc = v <= 0.0031308 ? 12.92 * v : 1.055 * pow(v, 1.0 / 2.4) - 0.055;
    const double q = floor(enc * 255.0 + 0.5);
    const double s = q / 255.0;
    const double dec = s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4);
    return (float) dec;
}

float orc_data_border(const tbrm_windowing_params* wp, int border_mode)
{
    /* LightingShaders.h:82-85: ZeroTFValue = Center - 0.5 * Width (float * double literal -> double), then
     * FLinearColor(float) and ToFColor(false) (linear 8-bit). */
    const float zero_tf = (float) ((double) wp->center - 0.5 * (double) wp->width);
    if (border_mode == TBRM_BORDER_EXACT_FLOAT) return zero_tf;
    float c = zero_tf;
    if (c != c) c = 0.0f;
    c = fminf(fmaxf(c, 0.0f), 1.0f);
    return floorf(c * 255.0f + 0.5f) / 255.0f;
}

/* FMajorAxes::GetMajorAxes (LightingShaderUtils.cpp:29-46) + GetLocalLightParamsAndAxes (:160-188) and the
 * per-axis parameter block of AddDirLightToSingleLightVolume_RenderThread (LightingShaders.cpp:91-131). */
static const double FACE_NORMALS[6][3] = {/* LightingShaderUtils.h:35-42 */
    {1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};

int orc_light_passes(const tbrm_dir_light_params* light, const tbrm_world_params* world,
                     const int32_t lv_dims[3], int border_mode, tbrm_light_pass out[2], int* n_passes)
{
    memset(out, 0, 2 * sizeof(tbrm_light_pass));
    *n_passes = 0;
    const v3 dir_world = v3_from(light->light_direction);
    if (dir_world.x == 0.0 && dir_world.y == 0.0 && dir_world.z == 0.0) return 1; /* LightingShaders.cpp:41-46 */

    /* LightingShaderUtils.cpp:167-169 */
    v3 dir_local = v3_normalize(inverse_transform_vector(&world->volume_transform, dir_world));
    const v3 light_pos = v3_neg(dir_local); /* :177 */

    int face[6];
    float weight[6];
    for (int i = 0; i < 6; ++i) {
        float w = (float) v3_dot(v3_make(FACE_NORMALS[i][0], FACE_NORMALS[i][1], FACE_NORMALS[i][2]), light_pos);
        w = (w > 0 ? w * w : 0); /* :40 */
        face[i] = i;
        weight[i] = w;
    }
    /* std::sort descending (:44) is unstable; the build fixes ties by ascending face index. */
    for (int i = 1; i < 6; ++i) {
        const int f = face[i];
        const float w = weight[i];
        int j = i - 1;
        while (j >= 0 && weight[j] < w) { face[j + 1] = face[j]; weight[j + 1] = weight[j]; --j; }
        face[j + 1] = f;
        weight[j + 1] = w;
    }
    if (weight[0] > 0.99f) weight[0] = 1.0f; /* :181-184 */
    weight[1] = 1 - weight[0];               /* :187 */

    for (int i = 0; i < 2; ++i) {
        tbrm_light_pass* p = &out[i];
        p->face = face[i];
        p->axis = face[i] / 2;
        p->weight = weight[i];
        p->light_alpha = light->light_intensity * weight[i]; /* GetLightAlpha :222-225 */
        p->border_light = border_mode == TBRM_BORDER_EXACT_FLOAT ? p->light_alpha : srgb8_round_trip(p->light_alpha);
        /* GetTransposedDimensions :48-64 */
        switch (p->axis) {
            case 0: p->td[0] = lv_dims[1]; p->td[1] = lv_dims[2]; p->td[2] = lv_dims[0]; break;
            case 1: p->td[0] = lv_dims[0]; p->td[1] = lv_dims[2]; p->td[2] = lv_dims[1]; break;
            default: p->td[0] = lv_dims[0]; p->td[1] = lv_dims[1]; p->td[2] = lv_dims[2]; break;
        }
        /* GetUVOffset :82-129 (FVector /= scalar multiplies by the reciprocal) */
        double major = p->axis == 0 ? light_pos.x : (p->axis == 1 ? light_pos.y : light_pos.z);
        if (face[i] % 2 == 1) major = -major; /* negative faces divide by -component */
        const v3 nlp = v3_scale(light_pos, 1.0 / major);
        double ou, ov;
        if (p->axis == 0) { ou = nlp.y; ov = nlp.z; }
        else if (p->axis == 1) { ou = nlp.x; ov = nlp.z; }
        else { ou = nlp.x; ov = nlp.y; }
        const double rtdz = 1.0 / (double) p->td[2];
        p->prev_pixel_offset[0] = (float) (ou * rtdz); /* FVector2f(PixelOffset), LightingShaders.h:153-156 */
        p->prev_pixel_offset[1] = (float) (ov * rtdz);
        /* GetStepSizeAndUVWOffset :132-158 */
        const double comp = fabs(p->axis == 0 ? light_pos.x : (p->axis == 1 ? light_pos.y : light_pos.z));
        v3 uvw = v3_scale(light_pos, 1.0 / (comp * (double) p->td[2]));
        p->step_size = (float) sqrt(uvw.x * uvw.x + uvw.y * uvw.y + uvw.z * uvw.z);
        /* LightingShaders.cpp:121-124 */
        int lowest = p->td[0] < p->td[1] ? p->td[0] : p->td[1];
        lowest = lowest < p->td[2] ? lowest : p->td[2];
        const float longest_side = 1.0f / (float) lowest;
        uvw = v3_scale(v3_normalize(uvw), (double) longest_side);
        p->uvw_offset[0] = (float) uvw.x; p->uvw_offset[1] = (float) uvw.y; p->uvw_offset[2] = (float) uvw.z;
        /* GetLoopStartStopIndexes :251-265 */
        p->dir = (face[i] % 2) ? 1 : -1;
        if (p->dir == -1) { p->start = p->td[2] - 1; p->stop = -1; }
        else { p->start = 0; p->stop = p->td[2]; }
    }
    *n_passes = (out[0].weight == 0) ? 0 : ((out[1].weight == 0) ? 1 : 2); /* LightingShaders.cpp:65,94 */
    return 0;
}

/* ================================================================================================ */
/* A12/A13: propagation (AddDirLightShader.usf:68-128, ChangeDirLightShader.usf:74-156)              */

static inline size_t lv_index(const int32_t d[3], int x, int y, int z) { return ((size_t) z * d[1] + y) * (size_t) d[0] + x; }

static inline void permute(int axis, int px, int py, int loop, int pos[3])
{
    /* mul(int3(px,py,Loop), PermutationMatrix) with GetPermutationMatrix (LightingShaderUtils.cpp:227-249) */
    if (axis == 0) { pos[0] = loop; pos[1] = px; pos[2] = py; }
    else if (axis == 1) { pos[0] = px; pos[1] = loop; pos[2] = py; }
    else { pos[0] = px; pos[1] = py; pos[2] = loop; }
}

static inline float buf_load(const void* b, int lv_fmt, size_t i) { return decode_voxel(b, lv_fmt, i); }
static inline void buf_store(void* b, int lv_fmt, size_t i, float v)
{
    if (lv_fmt == TBRM_FMT_G8) ((uint8_t*) b)[i] = encode_unorm8(v);
    else ((float*) b)[i] = v;
}

/* AlphaWeight of AddDirLightShader.usf:87-105. */
static inline float clip_alpha_weight(const float uvw[3], const float cc[3], const float cd[3], const int32_t res[3])
{
    const float dx = uvw[0] - cc[0], dy = uvw[1] - cc[1], dz = uvw[2] - cc[2];
    const float dist = (dx * cd[0] + dy * cd[1]) + dz * cd[2];
    const float ipx = uvw[0] + cd[0] * dist, ipy = uvw[1] + cd[1] * dist, ipz = uvw[2] + cd[2] * dist;
    const float ox = (uvw[0] - ipx) * (float) (uint32_t) res[0];
    const float oy = (uvw[1] - ipy) * (float) (uint32_t) res[1];
    const float oz = (uvw[2] - ipz) * (float) (uint32_t) res[2];
    const float vd = sqrtf((ox * ox + oy * oy) + oz * oz);
    const float sg = dist > 0.0f ? 1.0f : (dist < 0.0f ? -1.0f : 0.0f);
    return fminf(fmaxf(0.5f + ((0.57735026919f * vd) * sg), 0.0f), 1.0f);
}

typedef struct {
    const tbrm_light_pass* pass;
    const void* read;
    void* write;
} prop_stream;

/* one thread of MainComputeShader; `guard` = the all(uvw == saturate(uvw)) test that only Add has. */
static inline float propagate_voxel(const orc_scene* sc, const tbrm_light_pass* p, const void* read_buf,
                                    int px, int py, const int pos[3], const float cc[3], const float cd[3],
                                    float data_border, int guard)
{
    const int lv_fmt = sc->light_format;
    const float tsx = (float) p->td[0], tsy = (float) p->td[1];
    const float pu = (((float) (uint32_t) px + 0.5f) / tsx) + p->prev_pixel_offset[0];
    const float pv = (((float) (uint32_t) py + 0.5f) / tsy) + p->prev_pixel_offset[1];
    const float prev = sample_buffer_bilinear(read_buf, lv_fmt, p->td[0], p->td[1], pu, pv, p->border_light);

    float uvw[3];
    for (int c = 0; c < 3; ++c) /* GetUVW(pos, uResolution) + UVWOffset */
        uvw[c] = (((float) (uint32_t) pos[c] + 0.5f) / (float) (uint32_t) sc->light_dims[c]) + p->uvw_offset[c];

    const float aw = clip_alpha_weight(uvw, cc, cd, sc->light_dims);
    float cur = 0.0f;
    int inside = 1;
    if (guard)
        inside = (uvw[0] == saturatef(uvw[0])) && (uvw[1] == saturatef(uvw[1])) && (uvw[2] == saturatef(uvw[2]));
    if (aw > 0.0f && inside) {
        const float v = sample_volume_trilinear(&sc->data, uvw[0], uvw[1], uvw[2], ADDR_BORDER, data_border);
        float s[4];
        sample_windowed_transfer_function(v, p->step_size * 100.0f, sc->tf, &sc->windowing, s);
        cur = s[3] * aw;
    }
    return prev * (1 - cur);
}

static void clear_buffer(void* b, int fmt, size_t n, float v)
{
    for (size_t i = 0; i < n; ++i) buf_store(b, fmt, i, v); /* ClearTextureShader.usf:12-16 */
}

static size_t lv_elem_size(int fmt) { return fmt == TBRM_FMT_G8 ? 1 : 4; }

/* one axis pass of AddDirLightToSingleLightVolume_RenderThread (LightingShaders.cpp:100-158) */
static void add_light_pass(orc_scene* sc, const tbrm_light_pass* p, int b_added, const float cc[3], const float cd[3], float data_border)
{
    const int fmt = sc->light_format;
    const size_t npx = (size_t) p->td[0] * p->td[1];
    void* bufs[2] = {malloc(npx * lv_elem_size(fmt)), malloc(npx * lv_elem_size(fmt))};
    clear_buffer(bufs[0], fmt, npx, p->light_alpha); /* LightingShaders.cpp:76-79 */
    clear_buffer(bufs[1], fmt, npx, p->light_alpha);
    for (int j = p->start; j != p->stop; j += p->dir) {
        const void* rd = (j % 2 == 0) ? bufs[0] : bufs[1]; /* LightingShaders.cpp:149-156 */
        void* wr = (j % 2 == 0) ? bufs[1] : bufs[0];
#pragma omp parallel for schedule(static) if (npx >= 65536)
        for (int py = 0; py < p->td[1]; ++py)
            for (int px = 0; px < p->td[0]; ++px) {
                int pos[3];
                permute(p->axis, px, py, j, pos);
                const float l = propagate_voxel(sc, p, rd, px, py, pos, cc, cd, data_border, 1);
                buf_store(wr, fmt, (size_t) py * p->td[0] + px, l); /* AddDirLightShader.usf:120 */
                if (fabsf(l) > 1e-3f) { /* :123 */
                    const size_t li = lv_index(sc->light_dims, pos[0], pos[1], pos[2]);
                    buf_store(sc->light, fmt, li, buf_load(sc->light, fmt, li) + (l * (float) b_added)); /* :126 */
                }
            }
    }
    free(bufs[0]);
    free(bufs[1]);
}

int orc_add_dir_light(orc_scene* sc, const tbrm_dir_light_params* light, int added, const tbrm_world_params* world)
{
    tbrm_light_pass passes[2];
    int n = 0;
    if (orc_light_passes(light, world, sc->light_dims, sc->border_mode, passes, &n)) return 0; /* zero direction */
    float cc[3], cd[3];
    orc_local_clipping(world, cc, cd);
    const float data_border = orc_data_border(&sc->windowing, sc->border_mode);
    for (int i = 0; i < n; ++i) add_light_pass(sc, &passes[i], added ? 1 : -1, cc, cd, data_border);
    return n;
}

/* Only axis pass `pass` of the light (0 or 1): lets a checker replay the pass order of a batched multi-light add
 * (tbrm_add_dir_lights reports it). Returns 1 when the pass exists and ran, else 0. */
int orc_add_dir_light_pass(orc_scene* sc, const tbrm_dir_light_params* light, int added, const tbrm_world_params* world, int pass)
{
    tbrm_light_pass passes[2];
    int n = 0;
    if (orc_light_passes(light, world, sc->light_dims, sc->border_mode, passes, &n)) return 0;
    if (pass < 0 || pass >= n) return 0;
    float cc[3], cd[3];
    orc_local_clipping(world, cc, cd);
    add_light_pass(sc, &passes[pass], added ? 1 : -1, cc, cd, orc_data_border(&sc->windowing, sc->border_mode));
    return 1;
}

int orc_change_dir_light(orc_scene* sc, const tbrm_dir_light_params* old_light, const tbrm_dir_light_params* new_light,
                         const tbrm_world_params* world)
{
    const tbrm_vec3d a = new_light->light_direction, r = old_light->light_direction;
    if ((a.x == 0 && a.y == 0 && a.z == 0) || (r.x == 0 && r.y == 0 && r.z == 0)) return 0; /* LightingShaders.cpp:173-179 */
    tbrm_light_pass rp[2], ap[2];
    int rn = 0, an = 0;
    orc_light_passes(old_light, world, sc->light_dims, sc->border_mode, rp, &rn);
    orc_light_passes(new_light, world, sc->light_dims, sc->border_mode, ap, &an);
    if (rp[0].face != ap[0].face || rp[1].face != ap[1].face) { /* :192-198 */
        orc_add_dir_light(sc, old_light, 0, world);
        orc_add_dir_light(sc, new_light, 1, world);
        return -1;
    }
    float cc[3], cd[3];
    orc_local_clipping(world, cc, cd);
    const float data_border = orc_data_border(&sc->windowing, sc->border_mode);
    const int fmt = sc->light_format;

    for (int i = 0; i < 2; ++i) { /* no break on weight 0 (:238) */
        const tbrm_light_pass* pr = &rp[i];
        const tbrm_light_pass* pa = &ap[i];
        const size_t npx = (size_t) pr->td[0] * pr->td[1];
        void* bufs[4];
        for (int k = 0; k < 4; ++k) bufs[k] = malloc(npx * lv_elem_size(fmt));
        clear_buffer(bufs[0], fmt, npx, pr->light_alpha); /* :214-222 */
        clear_buffer(bufs[1], fmt, npx, pr->light_alpha);
        clear_buffer(bufs[2], fmt, npx, pa->light_alpha);
        clear_buffer(bufs[3], fmt, npx, pa->light_alpha);
        for (int j = pr->start; j != pr->stop; j += pr->dir) {
            const int e = (j % 2 == 0);
            const void* rrd = e ? bufs[0] : bufs[1]; /* :303-316 */
            void* rwr = e ? bufs[1] : bufs[0];
            const void* ard = e ? bufs[2] : bufs[3];
            void* awr = e ? bufs[3] : bufs[2];
#pragma omp parallel for schedule(static) if (npx >= 65536)
            for (int py = 0; py < pr->td[1]; ++py)
                for (int px = 0; px < pr->td[0]; ++px) {
                    int pos[3];
                    permute(pr->axis, px, py, j, pos);
                    const float lr = propagate_voxel(sc, pr, rrd, px, py, pos, cc, cd, data_border, 0);
                    const float la = propagate_voxel(sc, pa, ard, px, py, pos, cc, cd, data_border, 0);
                    buf_store(rwr, fmt, (size_t) py * pr->td[0] + px, lr); /* ChangeDirLightShader.usf:147-148 */
                    buf_store(awr, fmt, (size_t) py * pr->td[0] + px, la);
                    if (fabsf(la - lr) > 1e-3f) { /* :152 */
                        const size_t li = lv_index(sc->light_dims, pos[0], pos[1], pos[2]);
                        buf_store(sc->light, fmt, li, buf_load(sc->light, fmt, li) + la - lr); /* :154 */
                    }
                }
        }
        for (int k = 0; k < 4; ++k) free(bufs[k]);
    }
    return 2;
}

void orc_clear_light_volume(orc_scene* sc, float value)
{
    /* ClearVolumeTextureShader.usf:14-20 (the z == ZSize iteration is an out-of-bounds write D3D drops). */
    const size_t n = (size_t) sc->light_dims[0] * sc->light_dims[1] * sc->light_dims[2];
    clear_buffer(sc->light, sc->light_format, n, value);
}

/* ================================================================================================ */
/* A6-A11: raymarch                                                                                  */

/* Rand3DPCG16 (engine Random.ush; restated from its published recurrence). */
static void rand3d_pcg16(int px, int py, int pz, uint32_t out[3])
{
    uint32_t x = (uint32_t) px, y = (uint32_t) py, z = (uint32_t) pz;
    x = x * 1664525u + 1013904223u;
    y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u;
    x += y * z; y += z * x; z += x * y;
    x += y * z; y += z * x; z += x * y;
    out[0] = x >> 16; out[1] = y >> 16; out[2] = z >> 16;
}

typedef struct {
    float cam_pos[3], fwd[3], right[3], up[3];
    float thx, thy;
    float m[12];
    float cc[3], cd[3];
} ray_consts;

static void make_ray_consts(const tbrm_camera* cam, const tbrm_world_params* world, ray_consts* rc)
{
    rc->cam_pos[0] = (float) cam->position.x; rc->cam_pos[1] = (float) cam->position.y; rc->cam_pos[2] = (float) cam->position.z;
    rc->fwd[0] = (float) cam->forward.x; rc->fwd[1] = (float) cam->forward.y; rc->fwd[2] = (float) cam->forward.z;
    rc->right[0] = (float) cam->right.x; rc->right[1] = (float) cam->right.y; rc->right[2] = (float) cam->right.z;
    rc->up[0] = (float) cam->up.x; rc->up[1] = (float) cam->up.y; rc->up[2] = (float) cam->up.z;
    rc->thx = (float) cam->tan_half_fov_x;
    rc->thy = (float) cam->tan_half_fov_y;
    orc_world_to_local(&world->volume_transform, rc->m);
    orc_local_clipping(world, rc->cc, rc->cd);
}

static inline void mul3(const float v[3], const float m[12], float o[3])
{
    for (int c = 0; c < 3; ++c) o[c] = (v[0] * m[0 + c] + v[1] * m[3 + c]) + v[2] * m[6 + c];
}
static inline void normalize3(float v[3])
{
    const float l = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] = v[0] / l; v[1] = v[1] / l; v[2] = v[2] / l;
}

/* PerformRaymarchCubeSetup (RaymarchMaterialCommon.usf:23-69) for framebuffer pixel (px,py).
 * out: entry[3], thickness, local_cam_vec[3] (unit). */
static void cube_setup(const ray_consts* rc, const tbrm_camera* cam, int px, int py, const float* scene_depth,
                       float entry[3], float* thickness, float lcv[3])
{
    const float sx = (((2.0f * ((float) px + 0.5f)) / (float) cam->width) - 1.0f) * rc->thx;
    const float sy = (1.0f - ((2.0f * ((float) py + 0.5f)) / (float) cam->height)) * rc->thy;
    float d[3];
    for (int c = 0; c < 3; ++c) d[c] = (rc->fwd[c] + rc->right[c] * sx) + rc->up[c] * sy;
    normalize3(d);
    const float camvec[3] = {-d[0], -d[1], -d[2]}; /* MaterialParameters.CameraVector: pixel -> camera */

    /* :47-48 */
    float lcp[3];
    for (int c = 0; c < 3; ++c)
        lcp[c] = ((rc->cam_pos[0] * rc->m[0 + c] + rc->cam_pos[1] * rc->m[3 + c]) + rc->cam_pos[2] * rc->m[6 + c]) + rc->m[9 + c];
    mul3(camvec, rc->m, lcv);
    normalize3(lcv);
    lcv[0] = -lcv[0]; lcv[1] = -lcv[1]; lcv[2] = -lcv[2];
    for (int c = 0; c < 3; ++c) lcp[c] = lcp[c] + 0.5f; /* :51 */

    /* RayAABBIntersection(LocalCamPos, LocalCamVec, 0, 1)  (RaymarcherCommon.usf:66-88) */
    float t0 = -INFINITY, t1 = INFINITY;
    for (int c = 0; c < 3; ++c) {
        const float inv = 1.0f / lcv[c];
        const float tmin = (0.0f - lcp[c]) * inv;
        const float tmax = (1.0f - lcp[c]) * inv;
        const float lo = fminf(tmax, tmin), hi = fmaxf(tmax, tmin);
        if (c == 0) { t0 = lo; t1 = hi; }
        else { /* max(x, max(y,z)) / min(x, min(y,z)) are order-insensitive for non-NaN inputs */
            t0 = fmaxf(t0, lo);
            t1 = fminf(t1, hi);
        }
    }
    t0 = fmaxf(0.0f, t0); /* :57 */
    if (scene_depth) {    /* :26-44, :60 */
        float nv[3] = {camvec[0], camvec[1], camvec[2]};
        normalize3(nv);
        const float depth = scene_depth[(size_t) py * cam->width + px];
        float wdv[3] = {nv[0] * depth, nv[1] * depth, nv[2] * depth};
        float ldv[3];
        mul3(wdv, rc->m, ldv);
        float lsd = sqrtf((ldv[0] * ldv[0] + ldv[1] * ldv[1]) + ldv[2] * ldv[2]);
        lsd = lsd / fabsf((rc->fwd[0] * camvec[0] + rc->fwd[1] * camvec[1]) + rc->fwd[2] * camvec[2]);
        t1 = fminf(lsd, t1);
    }
    *thickness = fmaxf(0.0f, t1 - t0); /* :63 */
    for (int c = 0; c < 3; ++c) entry[c] = lcp[c] + (t0 * lcv[c]); /* :66 */
}

static inline int tile_row(const tbrm_tile* t, int j)
{
    const int step = t->row_group_step > 0 ? t->row_group_step : 1;
    return t->y0 + (j / 8) * 8 * step + (j % 8);
}

static inline int is_clipped(const float p[3], const float cc[3], const float cd[3])
{
    return (((p[0] - cc[0]) * cd[0] + (p[1] - cc[1]) * cd[1]) + (p[2] - cc[2]) * cd[2]) <= 0.0f; /* RaymarcherCommon.usf:24 */
}

/* AccumulateWindowedRaymarchStep (WindowedRaymarchMaterials.usf:21-33) */
static inline void accumulate_step(const orc_scene* sc, float le[4], const float pos[3], float step_size)
{
    const int mode = sc->data_address_mode == TBRM_ADDRESS_CLAMP ? ADDR_CLAMP : ADDR_WRAP;
    const float v = sample_volume_trilinear(&sc->data, pos[0], pos[1], pos[2], mode, 0.0f);
    float s[4];
    sample_windowed_transfer_function(v, step_size, sc->tf, &sc->windowing, s);
    orc_volume_view lv = {sc->light, sc->light_dims[0], sc->light_dims[1], sc->light_dims[2], sc->light_format};
    const float l = sample_volume_trilinear(&lv, saturatef(pos[0]), saturatef(pos[1]), saturatef(pos[2]), ADDR_WRAP, 0.0f);
    s[0] = s[0] * l; s[1] = s[1] * l; s[2] = s[2] * l;
    /* AccumulateLightEnergy (RaymarchMaterialCommon.usf:82-88) */
    const float om = 1.0f - le[3];
    le[0] = le[0] + ((s[0] * s[3]) * om);
    le[1] = le[1] + ((s[1] * s[3]) * om);
    le[2] = le[2] + ((s[2] * s[3]) * om);
    le[3] = le[3] + (s[3] * om);
}

void orc_raymarch_lit(const orc_scene* sc, const tbrm_camera* cam, const tbrm_tile* tile,
                      const tbrm_raymarch_params* rp, const tbrm_world_params* world,
                      const float* scene_depth, float* out_rgba, uint64_t* out_nominal_samples)
{
    ray_consts rc;
    make_ray_consts(cam, world, &rc);
    uint64_t total = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : total)
    for (int j = 0; j < tile->h; ++j) {
        const int py = tile_row(tile, j);
        for (int i = 0; i < tile->w; ++i) {
            const int px = tile->x0 + i;
            float pos[3], thickness, lcv[3];
            cube_setup(&rc, cam, px, py, scene_depth, pos, &thickness, lcv);

            /* PerformWindowedLitRaymarch (WindowedRaymarchMaterials.usf:36-96) */
            const float step_count = rp->steps;
            const float step_size = 1 / step_count;             /* :47 */
            const float actual = step_count * thickness;        /* :49 */
            const float fl = floorf(actual);
            const int max_steps = (int) fl;                     /* :51 */
            const float final_step = actual - fl;               /* :53 frac() */
            const float sv[3] = {lcv[0] * step_size, lcv[1] * step_size, lcv[2] * step_size}; /* :56 */
            const float step_world = 100.0f * step_size;        /* :58 */
            float le[4] = {0, 0, 0, 0};
            if (rp->jitter_frame >= 0) { /* JitterEntryPos (RaymarchMaterialCommon.usf:73-78) */
                uint32_t r[3];
                rand3d_pcg16(px, py, rp->jitter_frame & 7, r);
                const float rnd = (float) r[0] / 65535.0f;
                for (int c = 0; c < 3; ++c) pos[c] = pos[c] - (sv[c] * rnd);
            }
            total += (uint64_t) max_steps + (final_step > 0.0f ? 1u : 0u);
            if (!out_rgba) continue;
            int k = 0;
            for (k = 0; k < max_steps; k++) {
                for (int c = 0; c < 3; ++c) pos[c] = pos[c] + sv[c]; /* :67 */
                if (!is_clipped(pos, rc.cc, rc.cd)) {
                    accumulate_step(sc, le, pos, step_world);
                    if (le[3] > 0.95f) { le[3] = 1.0f; break; } /* :75-79 */
                }
            }
            if (k == max_steps && final_step > 0.0f) { /* :84-93 */
                for (int c = 0; c < 3; ++c) pos[c] = pos[c] + (sv[c] * final_step);
                if (!is_clipped(pos, rc.cc, rc.cd)) accumulate_step(sc, le, pos, 100.0f * final_step);
            }
            float* o = out_rgba + ((size_t) j * tile->w + i) * 4;
            o[0] = le[0]; o[1] = le[1]; o[2] = le[2]; o[3] = le[3];
        }
    }
    if (out_nominal_samples) *out_nominal_samples = total;
}

/* PerformWindowedIntensityRaymarch (WindowedRaymarchMaterials.usf:187-242): the windowed intensity of the first sample that
 * the clipping plane does not remove — the slice view of the cut surface. The data volume is read with the material's CLAMP
 * sampler at saturate(CurPos) in the full steps (:213-216) and at the unsaturated CurPos in the fractional step (:229-232). */
void orc_raymarch_intensity(const orc_scene* sc, const tbrm_camera* cam, const tbrm_tile* tile,
                            const tbrm_raymarch_params* rp, const tbrm_world_params* world,
                            const float* scene_depth, float* out_rgba)
{
    ray_consts rc;
    make_ray_consts(cam, world, &rc);
#pragma omp parallel for schedule(dynamic, 4)
    for (int j = 0; j < tile->h; ++j) {
        const int py = tile_row(tile, j);
        for (int i = 0; i < tile->w; ++i) {
            const int px = tile->x0 + i;
            float pos[3], thickness, lcv[3];
            cube_setup(&rc, cam, px, py, scene_depth, pos, &thickness, lcv);
            const float step_count = rp->steps;
            const float step_size = 1 / step_count;             /* :197 */
            const float actual = step_count * thickness;        /* :199 */
            const float fl = floorf(actual);
            const int max_steps = (int) fl;                     /* :201 */
            const float final_step = actual - fl;               /* :203 */
            const float sv[3] = {lcv[0] * step_size, lcv[1] * step_size, lcv[2] * step_size}; /* :206 */
            if (rp->jitter_frame >= 0) { /* JitterEntryPos (:208) */
                uint32_t r[3];
                rand3d_pcg16(px, py, rp->jitter_frame & 7, r);
                const float rnd = (float) r[0] / 65535.0f;
                for (int c = 0; c < 3; ++c) pos[c] = pos[c] - (sv[c] * rnd);
            }
            float* o = out_rgba + ((size_t) j * tile->w + i) * 4;
            o[0] = o[1] = o[2] = o[3] = 0.0f;                   /* :241 didn't hit anything */
            int hit = 0;
            for (int k = 0; k < max_steps && !hit; k++) {
                for (int c = 0; c < 3; ++c) pos[c] = pos[c] + sv[c]; /* :213 */
                const float sp[3] = {saturatef(pos[0]), saturatef(pos[1]), saturatef(pos[2])};
                if (!is_clipped(sp, rc.cc, rc.cd)) {            /* :215 */
                    const float v = sample_volume_trilinear(&sc->data, sp[0], sp[1], sp[2], ADDR_CLAMP, 0.0f); /* :217 */
                    const float t = saturatef(get_transfer_func_position(v, sc->windowing.center, sc->windowing.width)); /* :220 */
                    o[0] = o[1] = o[2] = t; o[3] = 1.0f;        /* :222 */
                    hit = 1;
                }
            }
            if (!hit && final_step > 0.0f) {                    /* :227-239 */
                for (int c = 0; c < 3; ++c) pos[c] = pos[c] + (sv[c] * final_step);
                if (!is_clipped(pos, rc.cc, rc.cd)) {
                    const float v = sample_volume_trilinear(&sc->data, pos[0], pos[1], pos[2], ADDR_CLAMP, 0.0f);
                    const float t = saturatef(get_transfer_func_position(v, sc->windowing.center, sc->windowing.width));
                    o[0] = o[1] = o[2] = t; o[3] = 1.0f;
                }
            }
        }
    }
}

/* ---- Octree render mode (ERaymarchMaterial::Octree) ------------------------------------------------------------------
 * GenerateOctreeShader.usf:28-107 driven by OctreeShaders.cpp:28-54: a 4-mip UNORM16 render target whose base level has the
 * volume's dimensions rounded up to powers of two (RaymarchVolume.cpp:873-877); mip 0 = Volume.Load(pos) * MinMaxValues.y
 * with MinMaxValues = (0, 1) (OctreeShaders.h:49) — loads outside the volume return 0 —, mip m = the maximum of the 2x2x2
 * texels of mip m-1 (:66-101). UNORM16 store: trunc(clamp(x, 0, 1) * 65535 + 0.5). */
static inline int pow2_at_least(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline uint16_t encode_unorm16(float x)
{
    if (x != x) return 0;
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    return (uint16_t) (x * 65535.0f + 0.5f);
}
void orc_octree_dims(const orc_volume_view* vol, int mip, int32_t out_dims[3])
{
    const int d[3] = {pow2_at_least(vol->dim_x), pow2_at_least(vol->dim_y), pow2_at_least(vol->dim_z)};
    for (int c = 0; c < 3; ++c) { int v = d[c] >> mip; out_dims[c] = v < 1 ? 1 : v; }
}
void orc_generate_octree(const orc_volume_view* vol, uint16_t* const mips[4])
{
    int32_t d0[3];
    orc_octree_dims(vol, 0, d0);
#pragma omp parallel for
    for (int z = 0; z < d0[2]; ++z)
        for (int y = 0; y < d0[1]; ++y)
            for (int x = 0; x < d0[0]; ++x) {
                const int in = x < vol->dim_x && y < vol->dim_y && z < vol->dim_z;
                const float v = in ? decode_voxel(vol->data, vol->format, ((size_t) z * vol->dim_y + y) * vol->dim_x + x) * 1.0f : 0.0f;
                mips[0][((size_t) z * d0[1] + y) * d0[0] + x] = encode_unorm16(v);
            }
    for (int m = 1; m < 4; ++m) {
        int32_t dl[3], dm[3];
        orc_octree_dims(vol, m - 1, dl);
        orc_octree_dims(vol, m, dm);
#pragma omp parallel for
        for (int z = 0; z < dm[2]; ++z)
            for (int y = 0; y < dm[1]; ++y)
                for (int x = 0; x < dm[0]; ++x) {
                    uint16_t mx = 0; /* "float Max = 0" (:79) over UNORM values: the same order as over their codes */
                    for (int c = 0; c < 2; ++c)
                        for (int b = 0; b < 2; ++b)
                            for (int a = 0; a < 2; ++a) {
                                const int sx = 2 * x + a, sy = 2 * y + b, sz = 2 * z + c;
                                if (sx < dl[0] && sy < dl[1] && sz < dl[2]) {
                                    const uint16_t t = mips[m - 1][((size_t) sz * dl[1] + sy) * dl[0] + sx];
                                    if (mx < t) mx = t;
                                }
                            }
                    mips[m][((size_t) z * dm[1] + y) * dm[0] + x] = mx;
                }
    }
}

/* PerformWindowedRaymarchOctree (WindowedRaymarchMaterials.usf:99-183): the unlit march over one mip of the octree, point
 * sampled with Load (SampleWindowedVolumeOctreeStep, WindowedSampling.usf:47-52; texels outside the mip read 0). */
void orc_raymarch_octree(const orc_scene* sc, const uint16_t* mip, int octree_mip, const tbrm_camera* cam, const tbrm_tile* tile,
                         const tbrm_raymarch_params* rp, const tbrm_world_params* world, const float* scene_depth, float* out_rgba)
{
    ray_consts rc;
    make_ray_consts(cam, world, &rc);
    int32_t d0[3], dm[3];
    orc_octree_dims(&sc->data, 0, d0);
    orc_octree_dims(&sc->data, octree_mip, dm);
    const float ow = (float) dm[0], oh = (float) dm[1], od = (float) dm[2];
    const float data_depth = (float) sc->data.dim_z, od0 = (float) d0[2];
#pragma omp parallel for schedule(dynamic, 4)
    for (int j = 0; j < tile->h; ++j) {
        const int py = tile_row(tile, j);
        for (int i = 0; i < tile->w; ++i) {
            const int px = tile->x0 + i;
            float pos[3], thickness, lcv[3];
            cube_setup(&rc, cam, px, py, scene_depth, pos, &thickness, lcv);
            const float step_count = rp->steps;
            const float step_size = 1 / step_count;             /* :113 */
            const float actual = step_count * thickness;
            const float fl = floorf(actual);
            const int max_steps = (int) fl;
            const float final_step = actual - fl;
            const float sv[3] = {lcv[0] * step_size, lcv[1] * step_size, lcv[2] * step_size};
            const float step_world = 100.0f * step_size;        /* :124 */
            float le[4] = {0, 0, 0, 0};
            if (rp->jitter_frame >= 0) {
                uint32_t r[3];
                rand3d_pcg16(px, py, rp->jitter_frame & 7, r);
                const float rnd = (float) r[0] / 65535.0f;
                for (int c = 0; c < 3; ++c) pos[c] = pos[c] - (sv[c] * rnd);
            }
            int k = 0;
            for (k = 0; k <= max_steps; k++) {
                float ss = step_world;
                if (k < max_steps) {
                    for (int c = 0; c < 3; ++c) pos[c] = pos[c] + sv[c];       /* :143 */
                } else {
                    if (!(final_step > 0.0f)) break;                            /* :169 */
                    for (int c = 0; c < 3; ++c) pos[c] = pos[c] + (sv[c] * final_step);
                    /* the fractional step still uses StepSizeWorld (:176), unlike the lit march */
                }
                if (is_clipped(pos, rc.cc, rc.cd)) continue;
                /* int3 VoxelPos = float3(x * W, y * H, (z * DataDepth / OctreeDepth0) * OctreeDepth) (:150,:174): truncation */
                const float fxp = pos[0] * ow, fyp = pos[1] * oh, fzp = ((pos[2] * data_depth) / od0) * od;
                const int vx = (int) fxp, vy = (int) fyp, vz = (int) fzp;
                float v = 0.0f;
                if (fxp == fxp && fyp == fyp && fzp == fzp && vx >= 0 && vy >= 0 && vz >= 0 && vx < dm[0] && vy < dm[1] && vz < dm[2])
                    v = (float) mip[((size_t) vz * dm[1] + vy) * dm[0] + vx] / 65535.0f;
                float s[4];
                sample_windowed_transfer_function(v, ss, sc->tf, &sc->windowing, s);
                const float om = 1.0f - le[3];                                  /* AccumulateLightEnergy */
                le[0] = le[0] + ((s[0] * s[3]) * om);
                le[1] = le[1] + ((s[1] * s[3]) * om);
                le[2] = le[2] + ((s[2] * s[3]) * om);
                le[3] = le[3] + (s[3] * om);
                if (k < max_steps && le[3] > 0.95f) { le[3] = 1.0f; break; }   /* :158-162 */
            }
            float* o = out_rgba + ((size_t) j * tile->w + i) * 4;
            o[0] = le[0]; o[1] = le[1]; o[2] = le[2]; o[3] = le[3];
        }
    }
}

/* Single-sample probes used by the known-answer tests. */
float orc_probe_sample_volume(const orc_volume_view* vol, float u, float v, float w, int mode, float border)
{
    return sample_volume_trilinear(vol, u, v, w, mode, border);
}
void orc_probe_windowed_tf(float value, float step_size, const float* tf, const tbrm_windowing_params* wp, float out[4])
{
    sample_windowed_transfer_function(value, step_size, tf, wp, out);
}
void orc_probe_ray_aabb(const float origin[3], const float dir[3], float out_t[2])
{
    float t0 = 0, t1 = 0;
    for (int c = 0; c < 3; ++c) {
        const float inv = 1.0f / dir[c];
        const float tmin = (0.0f - origin[c]) * inv, tmax = (1.0f - origin[c]) * inv;
        const float lo = fminf(tmax, tmin), hi = fmaxf(tmax, tmin);
        if (c == 0) { t0 = lo; t1 = hi; } else { t0 = fmaxf(t0, lo); t1 = fminf(t1, hi); }
    }
    out_t[0] = t0; out_t[1] = t1;
}
uint8_t orc_probe_encode_unorm8(float x) { return encode_unorm8(x); }
float orc_probe_srgb8_round_trip(float x) { return srgb8_round_trip(x); }
int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void) n;
#endif
}
