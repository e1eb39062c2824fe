// Kernel-argument block of the fused field kernel (field.hip), shared with the frame renderer.
#pragma once
#include <atomic>
#include <cstdint>

#include "../../include/cednerf_hip.h"

namespace ced {

struct FieldArgs {
    int64_t n;
    const int64_t *n_dev;                             // optional device-side sample count (<= n)
    unsigned long long *stamp;                        // optional device {min start, max end} of the launch on the constant
                                                      // wall clock (wall_clock64): first workgroup in, last workgroup out
    const int64_t *base_dev;                          // optional device-side first sample: the per-sample arrays of a
                                                      // rays-mode call (ray_idx32, t0, t1, rgb, sigma) start there
    const float *pos, *t, *dir;                       // explicit mode
    const float *rays_o, *rays_d;                     // rays mode
    const int64_t *ray_idx;
    const int32_t *ray_idx32;                         // alternative 32-bit ray indices (frame renderer)
    const float *t0, *t1, *timestamps;
    int rays_mode, t_per_ray, want_rgb;
    float *rgb, *sigma, *geo;
    float aabb[6];
    float moving_step;
    int use_div, time_mode;
    const void *weights;                              // packed blob of the descriptor's mlp_precision
    int table_dtype, temporal;
    int stagger;                                      // start-up phase offset between SIMD-mates (s_sleep(127) units)
    int spread_tiles;                                 // tile -> wave mapping (field.hip); ced_set_option("field_spread_tiles")
    int level_mode;                                   // 2 bits per gather slot: 0 mixed, 1 all dense, 2 all hashed, 3 all dense and cannot wrap
    int max_blocks;                                   // workgroups of the launch (ced_field_desc.max_workgroups; <= 0: one per CU)
    const void *table;
    float scale[CED_MAX_LEVELS];
    uint32_t res[CED_MAX_LEVELS], offset[CED_MAX_LEVELS], size[CED_MAX_LEVELS], hashed[CED_MAX_LEVELS];
};

extern std::atomic<int> g_march_early_out;
extern std::atomic<int> g_march_two_pass;
extern std::atomic<int> g_field_spread_tiles;
constexpr int kFieldBlocksDefault = 256;      // one persistent workgroup per CU

// Fills the field/hash parts of A from the descriptor, validates, and launches on `stream`.
int launch_field(const ced_field_desc *d, FieldArgs &A, void *stream);
// field_half.hip: the f16x2 / f16 MLP variants (A already filled by launch_field)
int launch_field_half(FieldArgs &A, int time_mode, int precision, void *stream);
void set_half_variant(int v);
// field_mixed.hip: exact sigma chain + split-fp16 colour head (CED_MLP_F32_HEAD16X2)
int launch_field_mixed(FieldArgs &A, int time_mode, void *stream);
void set_mixed_variant(int v);

}  // namespace ced
