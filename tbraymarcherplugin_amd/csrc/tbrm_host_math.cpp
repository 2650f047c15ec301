// tbrm_host_math.cpp — host-side parameter math of the illumination + raymarch path, in double like the
// reference (FVector/FTransform are double in UE5), narrowed to float exactly where the reference binds shader
// parameters (LightingShaders.h:100-101,:153-160).
//
// Restates (reference paths relative to Source/Raymarcher/):
//   FMajorAxes::GetMajorAxes                 Private/Rendering/LightingShaderUtils.cpp:29-46
//   GetTransposedDimensions / GetAxisDirection / GetLoopStartStopIndexes   :48-70, :251-265
//   GetUVOffset / GetStepSizeAndUVWOffset     :82-158
//   GetLocalLightParamsAndAxes                :160-188
//   GetBorderColorIntSingle / GetLightAlpha   :197-203, :222-225
//   GetLocalClippingParameters                :205-220
//   per-axis block of AddDirLightToSingleLightVolume_RenderThread   Private/Rendering/LightingShaders.cpp:100-131
//   FAddDirLightShader::SetRaymarchResources (data border colour)   Public/Rendering/LightingShaders.h:76-94
//   URaymarchUtils::ColorCurveToTexture / MakeDefaultTFTexture      Private/Util/RaymarchUtils.cpp:113-174
#include "tbrm_host_math.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

namespace tbrm {

namespace {

struct Vec3 {
    double x = 0, y = 0, z = 0;
    Vec3() = default;
    Vec3(double a, double b, double c) : x(a), y(b), z(c) {}
    explicit Vec3(const tbrm_vec3d& v) : x(v.x), y(v.y), z(v.z) {}
    Vec3 operator+(const Vec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
    Vec3 operator-(const Vec3& o) const { return {x - o.x, y - o.y, z - o.z}; }
    Vec3 operator*(const Vec3& o) const { return {x * o.x, y * o.y, z * o.z}; }
    Vec3 operator*(double s) const { return {x * s, y * s, z * s}; }
    Vec3 operator-() const { return {-x, -y, -z}; }
    double operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    double dot(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
    Vec3 cross(const Vec3& o) const { return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x}; }
    double size() const { return std::sqrt(x * x + y * y + z * z); }
    // FVector /= scalar multiplies by the reciprocal
    Vec3 divided_by(double s) const { const double r = 1.0 / s; return {x * r, y * r, z * r}; }
    // FVector::Normalize(SMALL_NUMBER)
    void normalize()
    {
        const double ss = x * x + y * y + z * z;
        if (ss > 1.e-8) { const double s = 1.0 / std::sqrt(ss); x *= s; y *= s; z *= s; }
    }
};

// FQuat::RotateVector / UnrotateVector
Vec3 rotate(const tbrm_quatd& q, const Vec3& v)
{
    const Vec3 qv(q.x, q.y, q.z);
    const Vec3 t = qv.cross(v) * 2.0;
    return v + t * q.w + qv.cross(t);
}
Vec3 unrotate(const tbrm_quatd& q, const Vec3& v) { return rotate(tbrm_quatd{-q.x, -q.y, -q.z, q.w}, v); }

// FTransform::GetSafeScaleReciprocal
Vec3 safe_reciprocal(const tbrm_vec3d& s)
{
    auto r = [](double v) { return std::fabs(v) <= 1.e-8 ? 0.0 : 1.0 / v; };
    return {r(s.x), r(s.y), r(s.z)};
}
Vec3 inverse_transform_vector(const tbrm_transform& t, const Vec3& v) { return unrotate(t.rotation, v) * safe_reciprocal(t.scale3d); }
Vec3 inverse_transform_vector_no_scale(const tbrm_transform& t, const Vec3& v) { return unrotate(t.rotation, v); }
Vec3 inverse_transform_position(const tbrm_transform& t, const Vec3& p)
{
    return unrotate(t.rotation, p - Vec3(t.translation)) * safe_reciprocal(t.scale3d);
}

// FLinearColor(v,0,0,0).ToFColor(true) packed into the sampler, decoded again by the RHI (engine code outside
// the reference; the build's definition: IEC 61966-2-1 encode, round-half-up to 8 bit, decode).
float srgb8_round_trip(float linear)
{
    double v = std::isnan(linear) ? 0.0 : std::clamp((double) linear, 0.0, 1.0);
    const double enc = v <= 0.0031308 ? v * 12.92 : 1.055 * std::pow(v, 1.0 / 2.4) - 0.055;
    const double s = std::floor(enc * 255.0 + 0.5) / 255.0;
    return (float) (s <= 0.04045 ? s / 12.92 : std::pow((s + 0.055) / 1.055, 2.4));
}

} // namespace

bool host_light_passes(const tbrm_dir_light_params& light, const tbrm_world_params& world, const int32_t lv[3],
                       int border_mode, tbrm_light_pass out[2], int* n_passes)
{
    std::memset(out, 0, 2 * sizeof(tbrm_light_pass));
    *n_passes = 0;
    const Vec3 world_dir(light.light_direction);
    if (world_dir.x == 0.0 && world_dir.y == 0.0 && world_dir.z == 0.0) return false;

    Vec3 local_dir = inverse_transform_vector(world.volume_transform, world_dir);
    local_dir.normalize();
    const Vec3 light_pos = -local_dir;

    static const Vec3 normals[6] = {{1, 0, 0}, {-1, 0, 0}, {0, 1, 0}, {0, -1, 0}, {0, 0, 1}, {0, 0, -1}};
    std::array<std::pair<int, float>, 6> faces;
    for (int i = 0; i < 6; ++i) {
        float w = (float) normals[i].dot(light_pos);
        w = (w > 0 ? w * w : 0);
        faces[i] = {i, w};
    }
    // the reference's std::sort is unstable; ties are fixed by ascending face index
    std::stable_sort(faces.begin(), faces.end(), [](const auto& a, const auto& b) { return a.second > b.second; });
    if (faces[0].second > 0.99f) faces[0].second = 1.0f;
    faces[1].second = 1 - faces[0].second;

    for (int i = 0; i < 2; ++i) {
        tbrm_light_pass& p = out[i];
        p.face = faces[i].first;
        p.axis = p.face / 2;
        p.weight = faces[i].second;
        p.light_alpha = light.light_intensity * p.weight;
        p.border_light = border_mode == TBRM_BORDER_EXACT_FLOAT ? p.light_alpha : srgb8_round_trip(p.light_alpha);
        const int a = p.axis;
        const int u_axis = a == 0 ? 1 : 0, v_axis = a == 2 ? 1 : 2;
        p.td[0] = lv[u_axis];
        p.td[1] = lv[v_axis];
        p.td[2] = lv[a];
        // GetUVOffset: divide by the (sign-corrected) major component, keep the two minor ones, then / TD.Z
        const double major = (p.face % 2 == 0) ? light_pos[a] : -light_pos[a];
        const Vec3 nlp = light_pos.divided_by(major);
        const double rz = 1.0 / (double) p.td[2];
        p.prev_pixel_offset[0] = (float) (nlp[u_axis] * rz);
        p.prev_pixel_offset[1] = (float) (nlp[v_axis] * rz);
        // GetStepSizeAndUVWOffset, then renormalisation to the longest voxel side
        Vec3 uvw = light_pos.divided_by(std::fabs(light_pos[a]) * (double) p.td[2]);
        p.step_size = (float) uvw.size();
        const int lowest = std::min({p.td[0], p.td[1], p.td[2]});
        const float longest_side = 1.0f / (float) lowest;
        uvw.normalize();
        uvw = uvw * (double) longest_side;
        p.uvw_offset[0] = (float) uvw.x;
        p.uvw_offset[1] = (float) uvw.y;
        p.uvw_offset[2] = (float) uvw.z;
        p.dir = (p.face % 2) ? 1 : -1;
        p.start = p.dir == -1 ? p.td[2] - 1 : 0;
        p.stop = p.dir == -1 ? -1 : p.td[2];
    }
    *n_passes = out[0].weight == 0 ? 0 : (out[1].weight == 0 ? 1 : 2);
    return true;
}

void host_local_clipping(const tbrm_world_params& world, float center[3], float dir[3])
{
    const Vec3 c = inverse_transform_position(world.volume_transform, Vec3(world.clipping_plane.center)) + Vec3(0.5, 0.5, 0.5);
    Vec3 d = inverse_transform_vector_no_scale(world.volume_transform, Vec3(world.clipping_plane.direction));
    d = d * Vec3(world.volume_transform.scale3d);
    d.normalize();
    center[0] = (float) c.x; center[1] = (float) c.y; center[2] = (float) c.z;
    dir[0] = (float) d.x; dir[1] = (float) d.y; dir[2] = (float) d.z;
}

float host_data_border(const tbrm_windowing_params& w, int border_mode)
{
    const float zero_tf = (float) ((double) w.center - 0.5 * (double) w.width);
    if (border_mode == TBRM_BORDER_EXACT_FLOAT) return zero_tf;
    float c = std::isnan(zero_tf) ? 0.0f : std::min(std::max(zero_tf, 0.0f), 1.0f);
    return std::floor(c * 255.0f + 0.5f) / 255.0f;
}

void host_world_to_local(const tbrm_transform& t, float m[12])
{
    const Vec3 rs = safe_reciprocal(t.scale3d);
    const Vec3 basis[3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r) {
        const Vec3 row = unrotate(t.rotation, basis[r]) * rs;
        m[r * 3 + 0] = (float) row.x; m[r * 3 + 1] = (float) row.y; m[r * 3 + 2] = (float) row.z;
    }
    const Vec3 tr = inverse_transform_position(t, Vec3(0, 0, 0));
    m[9] = (float) tr.x; m[10] = (float) tr.y; m[11] = (float) tr.z;
}

// min over the box [lo,hi]^3 (UVW space) of dot(p - centre, dir): the smallest signed distance any sample
// position in that box can have from the clip plane, measured along the kept direction. The callers use it to
// prove a clip plane inert (IsCurPosClipped never true; AlphaWeight exactly 1) and drop the per-sample test.
double host_min_plane_distance(const float cc[3], const float cd[3], double lo, double hi)
{
    double dmin = 0.0;
    for (int c = 0; c < 3; ++c) {
        const double n = cd[c];
        dmin += ((n >= 0.0 ? lo : hi) - (double) cc[c]) * n;
    }
    return dmin;
}

// ---- transfer function -------------------------------------------------------------------------------------

uint16_t float_to_half(float f) // FFloat16: IEEE binary16, round to nearest even
{
    uint32_t x;
    std::memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const int32_t exp = (int32_t) ((x >> 23) & 0xffu) - 127;
    uint32_t man = x & 0x7fffffu;
    if (exp == 128) return (uint16_t) (sign | 0x7c00u | (man ? (0x200u | (man >> 13)) : 0u));
    if (exp > 15) return (uint16_t) (sign | 0x7c00u);
    if (exp >= -14) {
        uint32_t h = ((uint32_t) (exp + 15) << 10) | (man >> 13);
        const uint32_t rest = man & 0x1fffu;
        if (rest > 0x1000u || (rest == 0x1000u && (h & 1u))) ++h; // may carry into the exponent: still correct
        return (uint16_t) (sign | h);
    }
    if (exp < -25) return (uint16_t) sign;
    man |= 0x800000u;
    const int shift = -exp - 1; // 14..24
    uint32_t h = man >> shift;
    const uint32_t rest = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rest > halfway || (rest == halfway && (h & 1u))) ++h;
    return (uint16_t) (sign | h);
}

float half_to_float(uint16_t h)
{
    const uint32_t sign = ((uint32_t) h & 0x8000u) << 16;
    const uint32_t exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    uint32_t out;
    if (exp == 0) {
        if (man == 0) out = sign;
        else {
            const float v = std::ldexp((float) man, -24);
            std::memcpy(&out, &v, 4);
            out |= sign;
        }
    } else if (exp == 31) out = sign | 0x7f800000u | (man << 13);
    else out = sign | ((exp + 112u) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &out, 4);
    return f;
}

void host_bake_tf(const float* rgba_256x4, float* out)
{
    for (int i = 0; i < 1024; ++i) out[i] = half_to_float(float_to_half(rgba_256x4[i]));
}

static float eval_curve(const float* times, const float* values, int n, float x)
{
    if (n <= 0) return 0.0f;
    if (x <= times[0]) return values[0];
    if (x >= times[n - 1]) return values[n - 1];
    const float* hi = std::upper_bound(times, times + n, x); // first key with time > x
    const int k = (int) (hi - times);
    const float t0 = times[k - 1], t1 = times[k];
    const float span = t1 - t0;
    if (!(span > 0.0f)) return values[k - 1];
    const float alpha = (x - t0) / span;
    return values[k - 1] + alpha * (values[k] - values[k - 1]);
}

void host_color_curve_to_lut(const float* const times[4], const float* const values[4], const int32_t n[4], float* out)
{
    for (unsigned i = 0; i < 256; ++i) {
        const float index = ((float) i) / ((float) 256 - 1);
        for (int c = 0; c < 4; ++c) out[i * 4 + c] = eval_curve(times[c], values[c], n[c], index);
    }
}

void host_default_tf_lut(float* out)
{
    for (unsigned i = 0; i < 256; ++i) {
        const float w = (float) i / (float) 255;
        out[i * 4 + 0] = out[i * 4 + 1] = out[i * 4 + 2] = w;
        out[i * 4 + 3] = 1.0f;
    }
}

} // namespace tbrm
