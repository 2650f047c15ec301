// tbrm_host_math.h — host parameter math (see tbrm_host_math.cpp for the reference lines each function restates).
#pragma once
#include "../../include/tbrm.h"

namespace tbrm {

bool host_light_passes(const tbrm_dir_light_params& light, const tbrm_world_params& world, const int32_t lv[3],
                       int border_mode, tbrm_light_pass out[2], int* n_passes); // false: zero direction
void host_local_clipping(const tbrm_world_params& world, float center[3], float dir[3]);
float host_data_border(const tbrm_windowing_params& w, int border_mode);
void host_world_to_local(const tbrm_transform& t, float m[12]);
double host_min_plane_distance(const float cc[3], const float cd[3], double lo, double hi);

uint16_t float_to_half(float f);
float half_to_float(uint16_t h);
void host_bake_tf(const float* rgba_256x4, float* out_rgba_256x4);
void host_color_curve_to_lut(const float* const times[4], const float* const values[4], const int32_t n[4], float* out);
void host_default_tf_lut(float* out);

} // namespace tbrm
