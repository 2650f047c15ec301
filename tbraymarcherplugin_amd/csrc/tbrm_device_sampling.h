// tbrm_device_sampling.h — voxel access in the bricked HBM layout + the restated D3D samplers.
//
// HBM layout (DESIGN.md "Data layout"): both the data volume and the light volume live as dense 8x8x8 bricks,
// brick-major, x fastest inside a brick. One brick = 512 voxels = 1 KiB (UNORM16) / 512 B (UNORM8) / 2 KiB (f32):
// whole cache lines per brick, so an 8x8 patch of rays or of slice pixels touches at most 4 lines per
// wave-instruction on ANY of the three propagation axes (a linear x-fastest layout gives 1, 1 and 64).
// D3D hides the same problem behind swizzled textures (SURVEY.md §7 "Axis-dependent memory order").
#pragma once
#include "tbrm_device_math.h"
#include "tbrm_internal.h"

namespace tbrm {

// offset of voxel (x,y,z) in voxels: separable in x, y, z
__device__ __forceinline__ uint32_t brick_off_x(int x) { return ((uint32_t) (x >> 3) << 9) | (uint32_t) (x & 7); }
__device__ __forceinline__ uint32_t brick_off_y(int y, int bnx) { return ((uint32_t) ((y >> 3) * bnx) << 9) | ((uint32_t) (y & 7) << 3); }
__device__ __forceinline__ uint32_t brick_off_z(int z, int bnxy) { return ((uint32_t) ((z >> 3) * bnxy) << 9) | ((uint32_t) (z & 7) << 6); }
__device__ __forceinline__ uint32_t brick_off(int x, int y, int z, int bnx, int bnxy)
{
    return brick_off_x(x) + brick_off_y(y, bnx) + brick_off_z(z, bnxy);
}

template <int FMT>
__device__ __forceinline__ float load_voxel(const void* p, size_t i)
{
    if constexpr (FMT == FMT_U8) return decode_u8(((const uint8_t*) p)[i]);
    else if constexpr (FMT == FMT_U16) return decode_u16(((const uint16_t*) p)[i]);
    else return ((const float*) p)[i];
}

template <int FMT>
__device__ __forceinline__ void store_voxel(void* p, size_t i, float v)
{
    if constexpr (FMT == FMT_U8) ((uint8_t*) p)[i] = (uint8_t) encode_u8(v);
    else ((float*) p)[i] = v;
}

// what a value becomes after a round trip through a buffer/volume of format FMT
template <int FMT>
__device__ __forceinline__ float through_format(float v)
{
    if constexpr (FMT == FMT_U8) return decode_u8f(quantize_u8(v));
    else return v;
}

__device__ __forceinline__ int wrap_fast(int i, int n)
{
    if ((unsigned) i >= (unsigned) n) {
        i = i < 0 ? i + n : i - n;
        if ((unsigned) i >= (unsigned) n) i = wrap_index(i, n);
    }
    return i;
}

template <int MODE>
__device__ __forceinline__ int address(int i, int n)
{
    if constexpr (MODE == ADDR_WRAP) return wrap_fast(i, n);
    else return clamp_index(i, n);
}

// Trilinear fetch with wrap or clamp addressing (the material samplers), bricked layout.
template <int FMT, int MODE>
__device__ __forceinline__ float sample_trilinear(const VolumeDev& v, int ix, int iy, int iz, float fx, float fy, float fz)
{
    const uint32_t x0 = brick_off_x(address<MODE>(ix, v.nx)), x1 = brick_off_x(address<MODE>(ix + 1, v.nx));
    const uint32_t y0 = brick_off_y(address<MODE>(iy, v.ny), v.bnx), y1 = brick_off_y(address<MODE>(iy + 1, v.ny), v.bnx);
    const uint32_t z0 = brick_off_z(address<MODE>(iz, v.nz), v.bnxy), z1 = brick_off_z(address<MODE>(iz + 1, v.nz), v.bnxy);
    const float t000 = load_voxel<FMT>(v.data, z0 + y0 + x0), t001 = load_voxel<FMT>(v.data, z0 + y0 + x1);
    const float t010 = load_voxel<FMT>(v.data, z0 + y1 + x0), t011 = load_voxel<FMT>(v.data, z0 + y1 + x1);
    const float t100 = load_voxel<FMT>(v.data, z1 + y0 + x0), t101 = load_voxel<FMT>(v.data, z1 + y0 + x1);
    const float t110 = load_voxel<FMT>(v.data, z1 + y1 + x0), t111 = load_voxel<FMT>(v.data, z1 + y1 + x1);
    const float c00 = lerp_(t000, t001, fx), c10 = lerp_(t010, t011, fx);
    const float c01 = lerp_(t100, t101, fx), c11 = lerp_(t110, t111, fx);
    const float c0 = lerp_(c00, c10, fy), c1 = lerp_(c01, c11, fy);
    return lerp_(c0, c1, fz);
}

// The eight brick offsets of a trilinear footprint (wrap or clamp addressing): computed once per sample and shared by
// every volume with the same dimensions.
struct TapOffsets {
    uint32_t x0, x1, y0, y1, z0, z1;
};

template <int MODE, bool RELOCATE = false> // RELOCATE: slab-resident volume (VolumeDev::wrap_layer)
__device__ __forceinline__ TapOffsets tap_offsets(const VolumeDev& v, int ix, int iy, int iz)
{
    // the +1 tap of an in-range base tap needs no general wrap: it is either base+1 or the first/last texel
    const int x0 = address<MODE>(ix, v.nx), y0 = address<MODE>(iy, v.ny);
    int z0 = address<MODE>(iz, v.nz);
    int x1 = x0 + 1, y1 = y0 + 1, z1 = z0 + 1;
    if constexpr (MODE == ADDR_WRAP) { x1 = x1 == v.nx ? 0 : x1; y1 = y1 == v.ny ? 0 : y1; z1 = z1 == v.nz ? 0 : z1; }
    else { // clamp: the +1 tap clamps on its own (a base tap of -1 and its +1 tap are BOTH texel 0)
        x1 = clamp_index(ix + 1, v.nx);
        y1 = clamp_index(iy + 1, v.ny);
        z1 = clamp_index(iz + 1, v.nz);
    }
    if constexpr (RELOCATE) {
        if ((z0 >> 3) == v.wrap_layer) z0 += v.wrap_shift;
        if ((z1 >> 3) == v.wrap_layer) z1 += v.wrap_shift;
    }
    TapOffsets t;
    t.x0 = brick_off_x(x0); t.x1 = brick_off_x(x1);
    t.y0 = brick_off_y(y0, v.bnx); t.y1 = brick_off_y(y1, v.bnx);
    t.z0 = brick_off_z(z0, v.bnxy); t.z1 = brick_off_z(z1, v.bnxy);
    return t;
}

template <int FMT>
__device__ __forceinline__ float sample_trilinear_at(const void* data, const TapOffsets& t, float fx, float fy, float fz)
{
    const float t000 = load_voxel<FMT>(data, t.z0 + t.y0 + t.x0), t001 = load_voxel<FMT>(data, t.z0 + t.y0 + t.x1);
    const float t010 = load_voxel<FMT>(data, t.z0 + t.y1 + t.x0), t011 = load_voxel<FMT>(data, t.z0 + t.y1 + t.x1);
    const float t100 = load_voxel<FMT>(data, t.z1 + t.y0 + t.x0), t101 = load_voxel<FMT>(data, t.z1 + t.y0 + t.x1);
    const float t110 = load_voxel<FMT>(data, t.z1 + t.y1 + t.x0), t111 = load_voxel<FMT>(data, t.z1 + t.y1 + t.x1);
    const float c00 = lerp_(t000, t001, fx), c10 = lerp_(t010, t011, fx);
    const float c01 = lerp_(t100, t101, fx), c11 = lerp_(t110, t111, fx);
    const float c0 = lerp_(c00, c10, fy), c1 = lerp_(c01, c11, fy);
    return lerp_(c0, c1, fz);
}

// Raw (undecoded) taps of a trilinear footprint: issuing the 8 loads and using them are separate steps, so a sample's
// loads can be in flight while the previous sample is being shaded.
template <int FMT> struct RawVoxel { using type = uint32_t; };
template <> struct RawVoxel<FMT_F32> { using type = float; };

template <int FMT>
__device__ __forceinline__ typename RawVoxel<FMT>::type load_raw(const void* p, uint32_t i)
{
    if constexpr (FMT == FMT_U8) return ((const uint8_t*) p)[i];
    else if constexpr (FMT == FMT_U16) return ((const uint16_t*) p)[i];
    else return ((const float*) p)[i];
}
template <int FMT>
__device__ __forceinline__ float decode_raw(typename RawVoxel<FMT>::type r)
{
    if constexpr (FMT == FMT_U8) return decode_u8(r);
    else if constexpr (FMT == FMT_U16) return decode_u16(r);
    else return r;
}

template <int FMT>
struct RawTaps {
    typename RawVoxel<FMT>::type t[8];
    __device__ __forceinline__ void issue(const void* data, const TapOffsets& o)
    {
        t[0] = load_raw<FMT>(data, o.z0 + o.y0 + o.x0); t[1] = load_raw<FMT>(data, o.z0 + o.y0 + o.x1);
        t[2] = load_raw<FMT>(data, o.z0 + o.y1 + o.x0); t[3] = load_raw<FMT>(data, o.z0 + o.y1 + o.x1);
        t[4] = load_raw<FMT>(data, o.z1 + o.y0 + o.x0); t[5] = load_raw<FMT>(data, o.z1 + o.y0 + o.x1);
        t[6] = load_raw<FMT>(data, o.z1 + o.y1 + o.x0); t[7] = load_raw<FMT>(data, o.z1 + o.y1 + o.x1);
    }
    __device__ __forceinline__ float filter(float fx, float fy, float fz) const
    {
        const float c00 = lerp_(decode_raw<FMT>(t[0]), decode_raw<FMT>(t[1]), fx), c10 = lerp_(decode_raw<FMT>(t[2]), decode_raw<FMT>(t[3]), fx);
        const float c01 = lerp_(decode_raw<FMT>(t[4]), decode_raw<FMT>(t[5]), fx), c11 = lerp_(decode_raw<FMT>(t[6]), decode_raw<FMT>(t[7]), fx);
        return lerp_(lerp_(c00, c10, fy), lerp_(c01, c11, fy), fz);
    }
};

// Trilinear fetch with border addressing from pre-split texel coordinates (the propagation shaders'
// VolumeSampler, LightingShaders.h:82-89).
template <int FMT>
__device__ __forceinline__ float sample_trilinear_border(const VolumeDev& v, int ix, int iy, int iz, float fx, float fy,
                                                         float fz, float border)
{
    float t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int x = ix + (k & 1), y = iy + ((k >> 1) & 1), z = iz + (k >> 2);
        const bool in = (unsigned) x < (unsigned) v.nx && (unsigned) y < (unsigned) v.ny && (unsigned) z < (unsigned) v.nz;
        t[k] = in ? load_voxel<FMT>(v.data, brick_off(x, y, z, v.bnx, v.bnxy)) : border;
    }
    const float c00 = lerp_(t[0], t[1], fx), c10 = lerp_(t[2], t[3], fx);
    const float c01 = lerp_(t[4], t[5], fx), c11 = lerp_(t[6], t[7], fx);
    const float c0 = lerp_(c00, c10, fy), c1 = lerp_(c01, c11, fy);
    return lerp_(c0, c1, fz);
}

template <int FMT>
__device__ __forceinline__ float sample_trilinear_border_uvw(const VolumeDev& v, float u, float vv, float w, float border)
{
    int ix, iy, iz;
    float fx, fy, fz;
    texel_split(u, (float) v.nx, ix, fx);
    texel_split(vv, (float) v.ny, iy, fy);
    texel_split(w, (float) v.nz, iz, fz);
    return sample_trilinear_border<FMT>(v, ix, iy, iz, fx, fy, fz, border);
}

// TF.SampleLevel(clamp-bilinear, (pos, 0.5)) — 1D linear between neighbouring texels of the 256-wide row.
__device__ __forceinline__ float4 sample_tf(const float4* tf, float pos)
{
    int i0;
    float f;
    texel_split(pos, 256.0f, i0, f);
    const int i1 = min(max(i0 + 1, 0), 255);
    i0 = min(max(i0, 0), 255);
    const float4 a = tf[i0], b = tf[i1];
    return make_float4(lerp_(a.x, b.x, f), lerp_(a.y, b.y, f), lerp_(a.z, b.z, f), lerp_(a.w, b.w, f));
}

// alpha channel only, from a 256-float alpha table
__device__ __forceinline__ float sample_tf_alpha(const float* tf_alpha, float pos)
{
    int i0;
    float f;
    texel_split(pos, 256.0f, i0, f);
    const int i1 = min(max(i0 + 1, 0), 255);
    i0 = min(max(i0, 0), 255);
    return lerp_(tf_alpha[i0], tf_alpha[i1], f);
}

// the same from a table of pairs: entry i + 1 = (alpha[clamp(i)], alpha[clamp(i + 1)]) for i = -1 .. 255 (257 entries, built by the
// workgroup) — one clamp and one 8-byte LDS read instead of two clamps and two reads; the same two operands, the same lerp
__device__ __forceinline__ float sample_tf_alpha(const float2* tf_alpha_pairs, float pos)
{
    int i0;
    float f;
    texel_split(pos, 256.0f, i0, f);
    const float2 ab = tf_alpha_pairs[min(max(i0, -1), 255) + 1];
    return lerp_(ab.x, ab.y, f);
}

// GetTransferFuncPosition for a value filtered out of UNORM data (in [0, 1]: the range the host's fast_div vouches for); float data
// always divides
template <bool UNORM_DATA>
__device__ __forceinline__ float window_position(float value, const WindowDev& w)
{
    if constexpr (UNORM_DATA) {
        if (w.fast_div) return tf_position_fast(value, w.center, w.width, w.inv_width); // (wave-uniform)
    }
    return tf_position(value, w.center, w.width);
}

// SampleWindowedTransferFunction(...).a  (WindowedSampling.usf:20-37)
// (NONNEG: the caller has checked step >= 0 — wave-uniform, so outside its loop — and the general pow_ drops out of the instantiation)
template <bool UNORM_DATA = false, bool NONNEG = false, class TF = float>
__device__ __forceinline__ float windowed_alpha(float value, float step, const TF* tf_alpha, const WindowDev& w)
{
    const float pos = window_position<UNORM_DATA>(value, w);
    if ((pos < 0.0f && w.low_cutoff > 0.0f) || (pos > 1.0f && w.high_cutoff > 0.0f)) return 0.0f;
    const float a = saturate_(sample_tf_alpha(tf_alpha, pos));
    if (a == 0.0f) return 0.0f; // 1 - pow(1, s) == 0 exactly
    if constexpr (NONNEG) return one_minus_pow01_(1.0f - a, step);
    else return step >= 0.0f ? one_minus_pow01_(1.0f - a, step) : 1.0f - pow_(1.0f - a, step); // (a in (0, 1]; step >= 0: wave-uniform)
}

// windowed_alpha(value, step0, ...) and windowed_alpha(value, step1, ...), bit for bit, sharing everything up to the opacity
// correction's exponent (one sample position, two axis passes)
template <bool UNORM_DATA = false, bool NONNEG = false, class TF = float>
__device__ __forceinline__ void windowed_alpha2(float value, float step0, float step1, const TF* tf_alpha, const WindowDev& w, float& a0, float& a1)
{
    a0 = 0.0f; a1 = 0.0f;
    const float pos = window_position<UNORM_DATA>(value, w);
    if ((pos < 0.0f && w.low_cutoff > 0.0f) || (pos > 1.0f && w.high_cutoff > 0.0f)) return;
    const float a = saturate_(sample_tf_alpha(tf_alpha, pos));
    if (a == 0.0f) return;
    if (NONNEG || (step0 >= 0.0f && step1 >= 0.0f)) { one_minus_pow01_2_(1.0f - a, step0, step1, a0, a1); return; } // (wave-uniform)
    float p0, p1;
    pow2_(1.0f - a, step0, step1, p0, p1);
    a0 = 1.0f - p0;
    a1 = 1.0f - p1;
}

// AlphaWeight (AddDirLightShader.usf:87-105)
__device__ __forceinline__ float clip_alpha_weight(float u, float v, float w, const float* cc, const float* cd, const int* res)
{
    const float dist = ((u - cc[0]) * cd[0] + (v - cc[1]) * cd[1]) + (w - cc[2]) * cd[2];
    const float ipx = u + cd[0] * dist, ipy = v + cd[1] * dist, ipz = w + cd[2] * dist;
    const float ox = (u - ipx) * (float) (uint32_t) res[0];
    const float oy = (v - ipy) * (float) (uint32_t) res[1];
    const float oz = (w - ipz) * (float) (uint32_t) res[2];
    const float vd = sqrtf((ox * ox + oy * oy) + oz * oz);
    const float sg = dist > 0.0f ? 1.0f : (dist < 0.0f ? -1.0f : 0.0f);
    return fminf(fmaxf(0.5f + ((0.57735026919f * vd) * sg), 0.0f), 1.0f);
}

} // namespace tbrm
