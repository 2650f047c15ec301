/*
 * tbrm_oracle.h — interface of the CPU oracle (TEST INFRASTRUCTURE ONLY; see tbrm_oracle.c).
 * Parameter structs are the public ABI PODs of include/tbrm.h so tests feed both sides identical bytes.
 */
#ifndef TBRM_ORACLE_H
#define TBRM_ORACLE_H

#include "../include/tbrm.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_volume_view {
    const void* data; /* dense, x fastest */
    int32_t dim_x, dim_y, dim_z;
    int32_t format; /* TBRM_FMT_* */
} orc_volume_view;

/* Everything FBasicRaymarchRenderingResources holds, as host pointers. */
typedef struct orc_scene {
    orc_volume_view data;
    const float* tf;             /* 256 x RGBA, already FFloat16-rounded (orc_bake_tf) */
    tbrm_windowing_params windowing;
    void* light;                 /* light volume, u8 (TBRM_FMT_G8) or float (TBRM_FMT_R32_FLOAT) */
    int32_t light_dims[3];
    int32_t light_format;
    int32_t data_address_mode;   /* TBRM_ADDRESS_* (raymarch material sampler) */
    int32_t border_mode;         /* TBRM_BORDER_* */
} orc_scene;

float orc_log2f(float x);
float orc_exp2f(float p);
float orc_powf(float x, float y);
float orc_round_to_half(float f);

void orc_bake_tf(const float* rgba_256x4, float* out_rgba_256x4);
void orc_color_curve_to_lut(const float* const key_times[4], const float* const key_values[4],
                            const int32_t n_keys[4], float* out_rgba_256x4);
void orc_make_default_tf_lut(float* out_rgba_256x4);

void orc_world_to_local(const tbrm_transform* t, float out_m[12]);
void orc_local_clipping(const tbrm_world_params* world, float out_center[3], float out_dir[3]);
float orc_data_border(const tbrm_windowing_params* wp, int border_mode);
int orc_light_passes(const tbrm_dir_light_params* light, const tbrm_world_params* world,
                     const int32_t lv_dims[3], int border_mode, tbrm_light_pass out[2], int* n_passes);

/* returns the number of axis passes run (Add), 2 for a fused Change, -1 for the remove+add fallback */
int orc_add_dir_light(orc_scene* sc, const tbrm_dir_light_params* light, int added, const tbrm_world_params* world);
int orc_add_dir_light_pass(orc_scene* sc, const tbrm_dir_light_params* light, int added, const tbrm_world_params* world, int pass);
int orc_change_dir_light(orc_scene* sc, const tbrm_dir_light_params* old_light,
                         const tbrm_dir_light_params* new_light, const tbrm_world_params* world);
void orc_clear_light_volume(orc_scene* sc, float value);

/* out_rgba may be NULL (count nominal samples only); scene_depth may be NULL. */
void orc_raymarch_lit(const orc_scene* sc, const tbrm_camera* cam, const tbrm_tile* tile,
                      const tbrm_raymarch_params* rp, const tbrm_world_params* world,
                      const float* scene_depth, float* out_rgba, uint64_t* out_nominal_samples);

/* the Intensity render mode (ERaymarchMaterial::Intensity) */
void orc_raymarch_intensity(const orc_scene* sc, const tbrm_camera* cam, const tbrm_tile* tile,
                            const tbrm_raymarch_params* rp, const tbrm_world_params* world,
                            const float* scene_depth, float* out_rgba);

/* the Octree render mode (ERaymarchMaterial::Octree): 4-mip UNORM16 max pyramid + its unlit, point-sampled march */
void orc_octree_dims(const orc_volume_view* vol, int mip, int32_t out_dims[3]);
void orc_generate_octree(const orc_volume_view* vol, uint16_t* const mips[4]);
void orc_raymarch_octree(const orc_scene* sc, const uint16_t* mip, int octree_mip, const tbrm_camera* cam, const tbrm_tile* tile,
                         const tbrm_raymarch_params* rp, const tbrm_world_params* world, const float* scene_depth, float* out_rgba);

float orc_probe_sample_volume(const orc_volume_view* vol, float u, float v, float w, int mode, float border);
void orc_probe_windowed_tf(float value, float step_size, const float* tf, const tbrm_windowing_params* wp, float out[4]);
void orc_probe_ray_aabb(const float origin[3], const float dir[3], float out_t[2]);
uint8_t orc_probe_encode_unorm8(float x);
float orc_probe_srgb8_round_trip(float x);
int orc_num_threads(void);
void orc_set_num_threads(int n);
void orc_debug_set_opacity_scale(float s); /* diagnostics: see tbrm_oracle.c */

#ifdef __cplusplus
}
#endif
#endif
