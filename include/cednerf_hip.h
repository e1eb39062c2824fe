/*
 * cednerf_hip.h -- C ABI of libcednerf_hip.so, the MI355X (gfx950) rendering hot path behind
 * Ced-NeRF's Python API.
 *
 * The reference has no FFI of its own for this path: its native entry points are the Python
 * bindings of two un-vendored CUDA packages (nerfacc's `nerfacc.cuda._C` ops and tiny-cuda-nn's
 * torch modules) plus Taichi kernels called with torch tensors.  Each function below names the
 * reference interface it replaces (file:line relative to /root/reference).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is documented "host";
 *   - tensors are dense, row-major, float32 unless stated; index tensors are int64 as in nerfacc;
 *     boolean tensors are one byte per element (torch.bool storage);
 *   - the library never allocates or frees user-visible memory and never synchronises the device:
 *     outputs and scratch are caller-allocated, work is enqueued on `stream` (a hipStream_t passed
 *     as void*, NULL = default stream);
 *   - return value 0 = success, negative = error (CED_E_*); ced_last_error_string() describes the
 *     last failure of the calling thread.  Nothing throws or aborts.
 *   - arithmetic contract: IEEE binary32, no FMA contraction except where the algorithm states
 *     one; dot products are ascending-k fused-multiply-add chains (the fp32 MFMA's native
 *     behaviour); per-ray sums run in sample order.  See DESIGN.md, "Arithmetic contract".
 */
#ifndef CEDNERF_HIP_H
#define CEDNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CED_MAX_LEVELS 16

#define CED_OK 0
#define CED_E_INVALID (-1)      /* bad argument (null pointer, negative size, unsupported option) */
#define CED_E_LAUNCH (-2)       /* HIP launch / runtime error */
#define CED_E_UNSUPPORTED (-3)  /* configuration outside what the kernels implement */

/* Multi-resolution hash grid description (host struct, copied at launch).
 * Replaces the tcnn `HashGrid` encoding built at cednerf/model.py:242-252; arithmetic spec
 * cednerf/taichi_kernel/hash_encoder_half.py:67-161 (and hash_encoder_inter.py:121-199 when
 * `temporal` != 0).  Level tables are computed by the host in float64. */
typedef struct ced_hash_desc {
    int32_t n_levels;            /* 1..16; the fused field kernel requires 16 */
    int32_t table_dtype;         /* 0 = float32 entries, 1 = float16 entries */
    int32_t temporal;            /* 0: entry = 2 features; 1: entry = 4 key-frames x 2 features */
    int32_t reserved;
    float scale[CED_MAX_LEVELS];
    uint32_t res[CED_MAX_LEVELS];
    uint32_t offset[CED_MAX_LEVELS];   /* first entry of the level */
    uint32_t size[CED_MAX_LEVELS];     /* entries in the level */
    uint32_t hashed[CED_MAX_LEVELS];   /* 1: xor-prime hash, 0: dense x + y*res + z*res^2 */
    const void *table;           /* device: [total_entries][2 or 8] */
    uint64_t total_entries;
} ced_hash_desc;

/* DNGPradianceField description (host struct).  Replaces the module built at
 * cednerf/model.py:100-344 (xyz_wrap :200-222, direction_encoding :225-239, hash_encoder :242-252,
 * mlp_base :280-290, mlp_head :291-309). */
typedef struct ced_field_desc {
    float aabb[6];
    float moving_step;           /* model.py:151, train_real.py:104,136,169 */
    int32_t use_div_offsets;     /* model.py:356-358 */
    int32_t time_mode;           /* 0 none, 1 SinusoidalEncoder, 2 SinusoidalEncoderWithExp (model.py:386-396) */
    int32_t mlp_precision;       /* CED_MLP_*: arithmetic of the three MLPs (below) */
    int32_t max_workgroups;      /* workgroups of a field launch: 0 = one per CU (256); callers that keep several
                                    frames in flight pass 128 so that two frames' field kernels run side by side.
                                    A per-call launch property (copy the descriptor to vary it), not process state. */
    const void *packed_weights;  /* device: blob written by ced_pack_field_weights[_half] for that precision */
    uint64_t packed_floats;      /* its size in 32-bit words: ced_packed_weight_words() */
    ced_hash_desc hash;
} ced_field_desc;

int ced_version(void);
const char *ced_last_error_string(void);

/* Diagnostic knobs (process-wide; results are identical for every setting -- launch properties that callers vary
 * per call, such as the workgroup count of a field launch, are descriptor fields instead).  "field_variant": launch geometry of the fused field kernel,
 * 0 = 4 column tiles x 512 threads, 1 = 2 x 512, 2 = 2 x 768, 3 = 2 x 1024 (default; the temporal-table kernels, which need
 * more registers, take 2 x 512), 4 = 1 x 1024;
 * "half_variant": the same for the half-precision kernels, 0 = automatic (default: f16x2 2 x 1024, f16 and the
 * time-embedding kernels 2 x 768, temporal tables 2 x 512), 1 = 2 x 512, 2 = 2 x 1024, 3 = 2 x 768;
 * "field_spread_tiles": 1 deals the sample tiles of a launch across all CUs in groups of four before any CU takes
 * more (shorter last round, lower frame latency); 2 (default) does the same inside each XCD, the eight XCDs taking eight
 * contiguous parts of the sample stream (one L2 per table line instead of up to eight); 0 = contiguous tiles per workgroup;
 * "field_stagger": start-up phase offset between the waves of a SIMD (0 = none, default);
 * "march_early_out": 1 (default) lets the frame renderer's marching cross empty space through the brick distance
 * field, 0 walks every cell;
 * "march_two_pass": the frame renderer's first iteration as a culling pass (the sphere trace alone, every ray) and a
 * marching pass over the rays it could not rule out: 1 / 0, -1 (default) = when there are several grid levels.
 * (Round 3's "march_sm" -- that pass on persistent waves with lane-level ray fetch -- was bit-exact and 3x slower; removed
 * in round 4, HISTORY 4.2 keeps the measurements.) */
int ced_set_option(const char *key, int value);

/* Arithmetic of xyz_wrap / mlp_base / mlp_head (everything else is fp32 in every mode):
 *   CED_MLP_F32    v_mfma_f32_16x16x4_f32, an ascending-k fp32 FMA chain: bit-identical to the CPU oracle.
 *   CED_MLP_F16X2  every operand split into two fp16 numbers (22 significant bits), three fp16 MFMA
 *                  blocks per product block, fp32 accumulation: fp32-grade results.
 *   CED_MLP_F16    operands rounded to fp16, fp32 accumulation: the precision class of the reference's
 *                  tiny-cuda-nn FullyFusedMLP (cednerf/model.py:200-222,280-309).
 * The half modes assume |weights|, |activations| <= 65504 (activations saturate there), like tcnn. */
#define CED_MLP_F32 0
#define CED_MLP_F16X2 1
#define CED_MLP_F16 2
/*   CED_MLP_F32_HEAD16X2  the chain that decides sample counts, opacity and depth -- xyz_wrap, the hash features,
 *                  mlp_base, trunc_exp (cednerf/model.py:354-445) -- exactly as CED_MLP_F32, bit-identical to the CPU
 *                  oracle; only mlp_head (model.py:447-466), which feeds rgb alone, on split-fp16 operands as
 *                  CED_MLP_F16X2.  sigma, counts, opacity and depth equal the exact mode's bits; rgb within 1e-4. */
#define CED_MLP_F32_HEAD16X2 3

/* Number of floats in the packed (MFMA-fragment-order) fp32 weight blob. */
int64_t ced_packed_weight_floats(int use_div_offsets, int time_mode);

/* HOST function: reorders natural-layout weights W[out][in] (row-major, host pointers) into the
 * fragment-order blob the fused kernel stages into LDS.  Layer dims: xyz_wrap 32-64-64-64-(3|6),
 * mlp_base (32|41)-64-16, mlp_head 19-64-64-3 (model.py:200-222,280-309).  `out` is host memory
 * of ced_packed_weight_floats() floats; upload it and put the device address in
 * ced_field_desc.packed_weights.  Replaces tcnn's flat `params` tensor layout. */
int ced_pack_field_weights(int use_div_offsets, int time_mode,
                           const float *m_w0, const float *m_w1, const float *m_w2, const float *m_w3,
                           const float *b_w0, const float *b_w1,
                           const float *h_w0, const float *h_w1, const float *h_w2,
                           float *out);

/* The same for the half-precision MLP modes: size of the blob in 32-bit words for any mlp_precision, and
 * the HOST packer for CED_MLP_F16X2 / CED_MLP_F16 (weights are rounded to fp16 here; F16X2 also stores the
 * fp16 remainders). */
int64_t ced_packed_weight_words(int use_div_offsets, int time_mode, int mlp_precision);
int ced_pack_field_weights_half(int use_div_offsets, int time_mode, int mlp_precision,
                                const float *m_w0, const float *m_w1, const float *m_w2, const float *m_w3,
                                const float *b_w0, const float *b_w1,
                                const float *h_w0, const float *h_w1, const float *h_w2,
                                void *out);
/* HOST packer for CED_MLP_F32_HEAD16X2: `out` holds ced_packed_weight_floats() floats (the head's region carries fp16
 * fragments). */
int ced_pack_field_weights_mixed(int use_div_offsets, int time_mode,
                                 const float *m_w0, const float *m_w1, const float *m_w2, const float *m_w3,
                                 const float *b_w0, const float *b_w1,
                                 const float *h_w0, const float *h_w1, const float *h_w2,
                                 float *out);

/* nerfacc.ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane, far_plane, miss_value)
 * -- call site cednerf/utils.py:215.  Outputs [n_rays, n_aabbs]. */
int ced_ray_aabb_intersect(int64_t n_rays, const float *rays_o, const float *rays_d,
                           int32_t n_aabbs, const float *aabbs,
                           float near_plane, float far_plane, float miss_value,
                           float *t_mins, float *t_maxs, uint8_t *hits, void *stream);

/* The sorted entry / exit events of cednerf/utils.py:219-225: per ray, torch.sort(cat([t_mins, t_maxs], -1), stable=True).
 * t_mins / t_maxs [n_rays, n_grids] (ced_ray_aabb_intersect's) -> t_sorted [n_rays, 2 n_grids], t_indices [n_rays,
 * 2 n_grids] int64 (event id: level for an entry, n_grids + level for an exit), n_grids <= 8.  Equal keys keep their
 * order, NaN keys sort last (torch's rules). */
int ced_sort_intersections(int64_t n_rays, int32_t n_grids, const float *t_mins, const float *t_maxs, float *t_sorted,
                           int64_t *t_indices, void *stream);

/* nerfacc.traverse_grids(...) -- call sites cednerf/utils.py:241-264 and, via
 * OccGridEstimator.sampling, cednerf/utils.py:115-125.  The caller allocates, so the op is split:
 *   mode 0  count: writes counts[n_rays] and termination_planes (t_starts/t_ends/ray_indices unused);
 *   mode 1  fill : writes samples of ray r at base[r] + i (base = exclusive scan of counts);
 *   mode 2  over-allocate (nerfacc `over_allocate=True`): ray r owns slots [r*limit, (r+1)*limit),
 *           counts[r] tells how many are valid; `base` unused; requires limit > 0;
 *   mode 3  fill, with `base` = INCLUSIVE scan of the counts of a preceding mode-0 call and
 *           `counts` still holding those counts (start of ray r = base[r] - counts[r]).
 * packed_info_out (may be NULL): [n_rays,2] = (start, count) written by the same launch.
 * near_planes and termination_planes may alias (each ray reads its near plane before writing).
 * rays_mask may be NULL (all rays).  ray_indices may be NULL.  binaries: [n_grids,res,res,res]
 * bytes; t_sorted/t_indices: [n_rays, 2*n_grids]; hits: [n_rays, n_grids]. */
int ced_traverse_grids(int64_t n_rays, const float *rays_o, const float *rays_d,
                       const uint8_t *binaries, int32_t n_grids, int32_t res, const float *aabbs,
                       const float *near_planes, const float *far_planes,
                       float step_size, float cone_angle, int32_t limit, const uint8_t *rays_mask,
                       const float *t_sorted, const int64_t *t_indices, const uint8_t *hits,
                       int32_t mode, const int64_t *base,
                       int64_t *counts, float *t_starts, float *t_ends, int64_t *ray_indices,
                       float *termination_planes, int64_t *packed_info_out, void *stream);

/* ---- occupancy acceleration structure of the frame renderer ----
 * Chebyshev distance fields over `binaries` (per 8^3-cell brick and per cell): the frame renderer's marching
 * sphere-traces them through empty space and re-enters the exact cell walk in closed form (csrc/march_accel.hpp), so
 * the emitted samples are those of nerfacc.traverse_grids (cednerf/utils.py:241-264) bit for bit.  Build it once per
 * occupancy-grid update (train_real.py:332-336) and hand it to ced_render_image_test / ced_render_frames_test; with
 * NULL there only the brick-level field is built, inside every call.
 * accel: device memory of ced_occupancy_accel_bytes(n_grids, res) bytes. */
int64_t ced_occupancy_accel_bytes(int32_t n_grids, int32_t res);
int ced_build_occupancy_accel(const uint8_t *binaries, int32_t n_grids, int32_t res, void *accel, int64_t accel_bytes,
                              void *stream);

/* HOST functions (validation aids, no GPU needed): the acceleration structure built on the host (accel_host:
 * ced_occupancy_accel_bytes() bytes of host memory, same layout as the device one); the closed-form DDA re-entry  k = 0; while (k < kcap && *x < tau) { *prev = *x; *x += d; ++k; }
 * and the frame renderer's marching of n_rays rays with the SAME code the device runs (march_accel.hpp is host +
 * device): counts[r] samples (<= limit) at t_starts/t_ends[r * limit + i]; t_term[r] is the termination plane when
 * counts[r] == limit (unspecified otherwise: such a ray is dead, cednerf/utils.py:303-306).  accel_mode: 0 = plain
 * cell-by-cell walk, 1 = brick field only (what a render call builds for itself), 2 = brick + cell fields (what
 * ced_build_occupancy_accel provides).  use_lattice: far skips start from the table of first lattice points per
 * binade, as in a frame with cone_angle == 0 (every near plane must then be a point of the lattice t_0 = near_planes[0],
 * t_{k+1} = t_k + step: the frame's near plane or an earlier termination plane).  start_coarse: 0 = the walk of a
 * frame's later iterations (no trace at a segment's start, emission inline), 1 = a frame's first iteration (trace
 * first, looking loop + emission phase), 2 = the one-shot march (trace first, emission inline).  All pointers are
 * host pointers. */
int ced_host_build_occupancy_accel(const uint8_t *binaries_host, int32_t n_grids, int32_t res, uint8_t *accel_host);
int32_t ced_host_count_steps(float *x, float d, float tau, int32_t kcap, float *prev);
int ced_host_march_frame(int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *binaries,
                         int32_t n_grids, int32_t res, const float *aabbs, const float *near_planes, float far_plane,
                         float step_size, float cone_angle, int32_t limit, const float *t_sorted,
                         const int64_t *t_indices, const uint8_t *hits, const uint8_t *accel_host, int32_t accel_mode,
                         int32_t use_lattice, int32_t start_coarse, int32_t *counts, float *t_starts, float *t_ends,
                         float *t_term);

/* HOST function (validation aid): the kernels' empty-space skip -- advance t_last by whole steps
 * dt = clamp(t*cone_angle, step_size, 1e10) until t_last + dt/2 >= target -- evaluated on the host
 * with the same code the device runs (closed form when cone_angle == 0). */
float ced_host_skip_march(float t_last, float target, float step_size, float cone_angle);

/* hash_encoder(x): the tcnn HashGrid forward at cednerf/model.py:384 (spec
 * hash_encoder_half.py:112-161).  x [n,3] in [0,1] (clamped), t [n] or NULL (temporal only),
 * out [n, 2*n_levels] level-major (hash_encoder_half.py:339-345,385). */
int ced_hash_encode(const ced_hash_desc *desc, int64_t n, const float *x, const float *t,
                    float *out, void *stream);

/* Backward of ced_hash_encode (non-temporal tables): the training-path row of SURVEY 8f, restating
 * hash_encoder_backward_kernel, taichi_kernel/hash_encoder_half.py:164-226.  dy [n, 2*n_levels] is the gradient
 * w.r.t. the encoder output; grad_table [total_entries, 2] fp32 is ACCUMULATED into (one hardware atomic add per
 * corner and feature; zero it first for a fresh gradient, as HashEncoder.backward does at :362-364); dx [n, 3]
 * (optional) receives the position gradient: dx_scaled = 0 as the reference computes it (per level w.r.t. the
 * scaled position, i.e. without the `scale` factor), dx_scaled = 1 the gradient w.r.t. x itself.  grad_table may be
 * NULL when dx is given (position gradient only): the two halves are independent launches and a caller may run them on
 * different streams. */
int ced_hash_encode_backward(const ced_hash_desc *desc, int64_t n, const float *x, const float *dy,
                             float *grad_table, float *dx, int32_t dx_scaled, void *stream);

/* Backward of ced_hash_encode for TEMPORAL tables (desc->temporal: entry = 4 key-frames x 2 features), restating
 * hash_encoder_backward_kernel of taichi_kernel/hash_encoder_inter.py:202-275: t [n] the sample times; grad_table
 * [total_entries, 8] fp32 is ACCUMULATED into -- per corner the key-frames k and k + 1 of the sample's time receive
 * w * dy * (1 - t_frac) and w * dy * t_frac (k, t_frac as in the forward, :228-240).  Like the reference's autograd
 * function (:403-420) it yields the table gradient only: positions and times get none through this encoder. */
int ced_hash_encode_backward_temporal(const ced_hash_desc *desc, int64_t n, const float *x, const float *t,
                                      const float *dy, float *grad_table, void *stream);

/* DNGPradianceField.forward(positions, t, directions) -- cednerf/model.py:468-488 (query_move
 * :354-365, query_density :367-445, _query_rgb :447-466), fused into one kernel.
 * dir/rgb may both be NULL (density only = query_density, model.py:367); geo [n,15] may be NULL
 * (= results['base_mlp_out']). */
int ced_field_forward(const ced_field_desc *desc, int64_t n, const float *positions, const float *t,
                      const float *directions, float *rgb, float *sigma, float *geo, void *stream);

/* The sigma_fn / rgb_sigma_fn closures of the render drivers (cednerf/utils.py:74-104,181-195)
 * fused with the field: positions = o[ray] + d[ray]*(t0+t1)/2, t = timestamps[t_per_ray ? ray : 0].
 * want_rgb == 0 evaluates the density only (sigma_fn).
 * n_dev (may be NULL): device scalar; the kernel evaluates min(n, *n_dev) samples, so a caller
 * that only knows an upper bound of the sample count on the host needs no device->host sync. */
int ced_field_forward_rays(const ced_field_desc *desc, int64_t n, const int64_t *n_dev,
                           const float *rays_o, const float *rays_d,
                           const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                           const float *timestamps, int32_t t_per_ray, int32_t want_rgb,
                           float *rgb, float *sigma, void *stream);

/* nerfacc.render_weight_from_density / render_transmittance_from_density with packed_info
 * [n_rays,2] = (start,count) -- call sites cednerf/render.py:52-54,81-87, cednerf/utils.py:274-281.
 * prefix_trans is per sample (NULL = 1); weights/trans/alphas may each be NULL. */
int ced_render_weights(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                       const float *t_ends, const float *sigmas, const float *prefix_trans,
                       float *weights, float *trans, float *alphas, void *stream);

/* nerfacc.accumulate_along_rays(_): out[ray, :] += sum_i w_i * values[i, :] (values NULL: C must
 * be 1 and the weights are summed) -- cednerf/render.py:158-169, cednerf/utils.py:282-299. */
int ced_accumulate_along_rays(int64_t n_rays, const int64_t *packed_info, const float *weights,
                              const float *values, int32_t n_channels, float *out, void *stream);

/* reduce_along_rays(ray_indices, values, n_rays, weights, reduce) -- cednerf/render.py:8-39 (the training extras of
 * rendering(), render.py:101-124): out [n_rays, C] = scatter_reduce_ of weights * values into zeros, reduce "sum"
 * (mean == 0) or "mean" (mean == 1, with the initial zero counted as an element, torch's include_self default).
 * values [S, C]; weights NULL or [S, weight_channels] with weight_channels 1 (broadcast) or C; ray indices outside
 * [0, n_rays) are ignored; counts_workspace: n_rays int32 of device scratch (mean only).  Float atomics, as in torch:
 * sums of unsorted indices depend on arrival order in the last bits. */
int ced_reduce_along_rays(int64_t n_samples, const int64_t *ray_indices, const float *values, int32_t n_channels,
                          const float *weights, int32_t weight_channels, int64_t n_rays, int32_t mean, float *out,
                          int32_t *counts_workspace, void *stream);

/* nerfacc.render_visibility_from_density inside OccGridEstimator.sampling (cednerf/utils.py:115-125):
 * mask[i] = trans_i >= early_stop_eps && (alpha_thre <= 0 || alpha_i >= alpha_thre). */
int ced_visibility_mask(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                        const float *t_ends, const float *sigmas, float early_stop_eps, float alpha_thre,
                        uint8_t *mask, void *stream);

/* One pass of cednerf/utils.py:274-299 (render_weight_from_density with
 * prefix_trans = 1 - opacity[ray], then the three accumulate_along_rays_) fused per ray, in place.
 * packed_info [n_rays,2] indexes the sample arrays; rgbs [S,3]. */
int ced_composite_prefix(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                         const float *t_ends, const float *sigmas, const float *rgbs,
                         float *rgb, float *opacity, float *depth, void *stream);

/* Backward of the training-time compositing: the derivative of the un-normalised
 * (colors, opacities, depths) = accumulate_along_rays(render_weight_from_density(...), {rgbs, 1, t_mid}) of
 * cednerf/render.py:158-169 w.r.t. the per-sample sigmas [S] and rgbs [S,3], given d_color [n_rays,3],
 * d_opacity [n_rays] and d_depth [n_rays] (the last two may be NULL = zero).  Replaces the backward of nerfacc's
 * render_weight_from_density / accumulate_along_rays autograd functions. */
int ced_composite_backward(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                           const float *t_ends, const float *sigmas, const float *rgbs,
                           const float *d_color, const float *d_opacity, const float *d_depth,
                           float *d_sigmas, float *d_rgbs, void *stream);

/* Weight gradient of a bias-free dense layer over the sample stream:
 *   dw[o][i] = sum_s dy[s][o] * x[s][i],   x [n, n_in], dy [n, n_out], dw [n_out, n_in], widths 1..64, all fp32,
 * x and dy contiguous and 16-byte aligned.  Replaces the weight-gradient GEMM of tiny-cuda-nn's Network backward
 * (modules of cednerf/model.py:200-222,280-309 under loss.backward(), train_real.py:414-419).  fp32 MFMA
 * accumulation, two deterministic stages (per-workgroup partial tiles in `workspace`, then a fixed-order sum):
 * reproducible run to run.  ced_weight_grad_workspace_bytes gives the scratch size for n samples. */
/* The 8-bit frames of the reference's video step, on the device:
 * ced_frame_to_rgb8: out[y][x'][c] = uint8(rgb[y][x][c] * 255), x' = width-1-x when flip_w
 *   (np.flip(rgb * 255, axis=1).astype(np.uint8), train_real.py:556); rgb [H,W,3] f32, out [H,W,3] u8.
 * ced_depth_to_u8: out = uint8((depth - min) / (max - min) * 255) over the image (depth2img before the colour-map
 *   lookup, train_real.py:38-41), same flip; depth [H,W] f32, out [H,W] u8, workspace = 8 bytes of device memory. */
int ced_frame_to_rgb8(int32_t height, int32_t width, const float *rgb, int32_t flip_w, uint8_t *out, void *stream);
int ced_depth_to_u8(int32_t height, int32_t width, const float *depth, int32_t flip_w, uint8_t *out,
                    void *workspace, void *stream);

/* The exchange's consumer ("a local un-permute fused into the consumer"): rendered pixels arrive in marching
 * order (8x8-tile order; with several GPUs every rank's shard of it, all-gathered as [rows, 5] = rgb, opacity, depth);
 * row i goes to raster pixel dest[i] (rows with dest outside [0, n_pixels) are padding).  One pass writes the raster
 * rgb [n_pixels,3] / opacity [n_pixels] / depth [n_pixels] (each may be NULL) and, when rgb8 != NULL, the 8-bit
 * colour frame of ced_frame_to_rgb8 directly ([n_pixels,3] uint8, flipped inside rows of `width` pixels when flip_w).
 * Sources are given with their row strides in floats (5, 5, 5 for the gathered payload; 3, 1, 1 for separate arrays). */
int ced_scatter_pixels(int64_t n_rows, const float *src_rgb, int32_t stride_rgb, const float *src_opacity,
                       int32_t stride_opacity, const float *src_depth, int32_t stride_depth, const int64_t *dest,
                       int64_t n_pixels, float *rgb, float *opacity, float *depth, uint8_t *rgb8, int32_t width,
                       int32_t flip_w, void *stream);

/* Bias-free dense layer over the sample stream, hand-written (csrc/linear.hip) in place of a library GEMM -- the
 * forward and the input gradient of the tiny-cuda-nn Networks the reference trains through (cednerf/model.py:200-222,
 * 280-344; loss.backward(), train_real.py:412-420):
 *   transpose_w == 0:  y[s][o] = sum_i x[s][i] * w[o][i]     w [n_out, n_in]   (forward; relu != 0: y = max(y, 0))
 *   transpose_w != 0:  y[s][o] = sum_i x[s][i] * w[i][o]     w [n_in, n_out]   (input gradient dz W of a layer W)
 * then, when mask != NULL ([n, n_out]):  y[s][o] = mask[s][o] > 0 ? y[s][o] : 0  (the ReLU derivative of the layer
 * below, fused).  x [n, n_in], y [n, n_out], widths 1..64, fp32 MFMA with fp32 accumulation.  w_rows / w_cols are
 * checked against n_in / n_out.  The weight gradient of the same layer is ced_weight_grad. */
int ced_linear(int64_t n, const float *x, int32_t n_in, const float *w, int32_t w_rows, int32_t w_cols,
               int32_t transpose_w, int32_t n_out, int32_t relu, const float *mask, float *y, void *stream);

int64_t ced_weight_grad_workspace_bytes(int64_t n, int32_t n_out, int32_t n_in);
int ced_weight_grad(int64_t n, const float *x, int32_t n_in, const float *dy, int32_t n_out, float *dw,
                    void *workspace, int64_t workspace_bytes, void *stream);

/* ced_composite_prefix plus the per-iteration bookkeeping of cednerf/utils.py:301-307 in the same
 * launch: ray_mask[r] = opacity[r] <= opc_thres && count[r] == n_samples_iter, and
 * stats[0] += number of rays still alive, stats[1] += samples composited (device int64[2], the
 * caller zeroes it).  Lets the host loop read both numbers with a single 16-byte copy. */
int ced_composite_step(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                       const float *t_ends, const float *sigmas, const float *rgbs,
                       float *rgb, float *opacity, float *depth,
                       float opc_thres, int32_t n_samples_iter, uint8_t *ray_mask, int64_t *stats,
                       void *stream);

/* composite_test -- cednerf/taichi_kernel/volume_render_test.py:4-59 (same arguments, same
 * in-place semantics; alive_indices entries are set to -1 when a ray finishes). */
int ced_composite_test(int64_t n_alive, const float *sigmas, const float *rgbs, const float *t_start,
                       const float *t_end, const int64_t *pack_info, int64_t *alive_indices,
                       float T_threshold, float alpha_threshold,
                       float *opacity, float *depth, float *rgb, void *stream);

/* Tail of render_image_test / rendering (cednerf/utils.py:310-311, cednerf/render.py:170-174):
 * rgb += bkgd * (1 - opacity); depth /= max(opacity, FLT_EPSILON).  bkgd: device [3] or NULL. */
int ced_finalize_pixels(int64_t n_rays, const float *bkgd, float *rgb, const float *opacity,
                        float *depth, void *stream);

/* ---- occupancy-grid maintenance (nerfacc OccGridEstimator._update as driven at
 * train_real.py:324-336).  The density query in between is ced_field_forward with directions == NULL. ---- */

/* World positions of jittered cell samples: cell index c = (ix*res + iy)*res + iz of one grid level,
 * x = aabb_min + ((ix,iy,iz) + noise) / res * (aabb_max - aabb_min).  aabb_host: HOST float[6] of that
 * level; noise [n,3] in [0,1); positions [n,3] out. */
int ced_occ_cell_points(int64_t n, const int64_t *cell_indices, const float *noise, int32_t res,
                        const float *aabb_host, float *positions, void *stream);

/* occs[cell_ids[i]] = max(occs[cell_ids[i]] * ema_decay, density[i] * step_size)  (the EMA write-back of
 * _update; duplicates in cell_ids race exactly like the index assignment they replace). */
int ced_occ_ema_update(int64_t n, const int64_t *cell_ids, const float *density, float step_size,
                       float ema_decay, float *occs, void *stream);

/* ---- on-device ray generation: full-frame rays from camera parameters, so no
 * [H,W,3] x 2 upload per frame.  Outputs [height*width, 3] row-major (pixel x fastest), i.e. Rays of
 * shape [H,W,3]. ---- */

/* Pinhole camera of datasets/dnerf_synthetic.py:191-221 / gui.py:43-86: pixel (x,y) ->
 * [(x-cx+.5)/fx, (y-cy+.5)/fy * s, s] (s = -1 OpenGL, +1 OpenCV) rotated by c2w[:3,:3]; origins = c2w[:3,3];
 * viewdirs = normalised directions.  c2w_host: HOST float[12] ([3][4] row-major); directions may be NULL. */
int ced_generate_rays_pinhole(int32_t width, int32_t height, float fx, float fy, float cx, float cy,
                              const float *c2w_host, int32_t opengl, float *origins, float *viewdirs,
                              float *directions, void *stream);

/* HyperNeRF camera, datasets/hyper_cam.py:210-252 on get_pixel_centers() (:299-303), as used at
 * datasets/hypernerf.py:172-176: skew / aspect / principal point, radial (k1,k2,k3) + tangential (p1,p2)
 * undistortion by 10 Newton steps (:22-91), rotation by orientation^T.  orientation_host float[9]
 * (row-major), position_host float[3], radial3_host / tangential2_host HOST arrays (NULL = none). */
int ced_generate_rays_hypercam(int32_t width, int32_t height, const float *orientation_host,
                               const float *position_host, float focal_length, float principal_x, float principal_y,
                               float skew, float pixel_aspect_ratio, const float *radial3_host,
                               const float *tangential2_host, float *origins, float *viewdirs, void *stream);

/* Optional per-iteration trace of ced_render_image_test (host struct, host arrays of `capacity`
 * entries, each may be NULL).  field_begin/field_end are caller-created hipEvent_t handles that the
 * renderer records on `stream` around the field-kernel launch of iteration i, so a benchmark can
 * time that kernel live without perturbing the launch sequence.  Events of iterations >= n_iters may have been
 * recorded around empty launches (the host enqueues ahead of the device-side schedule). */
typedef struct ced_frame_trace {
    int32_t capacity;        /* in  */
    int32_t n_iters;         /* out: iterations executed */
    void **field_begin;      /* in  */
    void **field_end;        /* in  */
    int64_t *iter_alive;     /* out: rays alive entering iteration i (N_alive, cednerf/utils.py:231) */
    int64_t *iter_n_samples; /* out: samples per ray requested in iteration i (N_samples, utils.py:235) */
    int64_t *iter_samples;   /* out: samples marched and composited in iteration i */
    uint64_t *field_stamps;  /* in: DEVICE memory [2 * capacity] or NULL.  Out (device): for iteration i the ticks
                                {first workgroup of the field launch started, its last workgroup finished} on the
                                device's constant wall clock (ced_wall_clock_khz ticks per ms): the interval the kernel
                                was EXECUTING, where the event pair also counts its wait for free CUs */
} ced_frame_trace;

/* Rate of the device wall clock the field_stamps are taken on (hipDeviceAttributeWallClockRate), in kHz; < 0 on error. */
int64_t ced_wall_clock_khz(void);

/* Device bytes ced_render_image_test needs in `workspace` (negative on bad arguments). */
int64_t ced_render_image_test_workspace_bytes(int64_t n_rays, int32_t n_grids, int32_t res, float cone_angle,
                                              int32_t max_samples);

/* render_image_test(max_samples, radiance_field, estimator, rays, near_plane, far_plane,
 * render_step_size, render_bkgd, cone_angle, alpha_thre, early_stop_eps, timestamps)
 * -- cednerf/utils.py:153-318, the eval / GUI frame renderer (callers train_real.py:481-493,
 * :541-553, gui.py:215-228), as ONE call: ray/AABB setup, then per iteration one marching launch,
 * one fused field launch, one compositing launch and a one-wave scheduling launch that computes the next
 * iteration's N_samples = clamp(N_rays // N_alive, min, 64) ON THE DEVICE, and the background / depth
 * normalisation at the end.  Same schedule and per-ray sample sets as the reference loop; unlike it
 * (utils.py:231: one device->host sync per iteration) the host never waits for an iteration: it enqueues up to
 * CED_FRAME_RUN_AHEAD (default 1) iterations beyond the last one whose plan it has seen published in `host_stats`.
 * `alpha_thre` is not a parameter because the reference ignores it in this function.
 *   rays_o, rays_d [n_rays,3]; binaries [n_grids,res,res,res] bytes; aabbs [n_grids,6];
 *   accel: device, from ced_build_occupancy_accel for these binaries, or NULL (built inside the call);
 *   timestamps: device [1] (t_per_ray = 0, eval) or [n_rays] (t_per_ray = 1); bkgd: device [3] or NULL;
 *   outputs rgb [n_rays,3], opacity [n_rays], depth [n_rays] (overwritten);
 *   workspace: device scratch of ced_render_image_test_workspace_bytes() bytes;
 *   host_stats: PINNED host memory, >= 2048 bytes, private to this call while it runs (the scheduling launches
 *   publish {rays alive, done, sequence number} there; the call's table of lattice points travels through it);
 *   total_samples_out: host, receives the number of field evaluations (utils.py:307,317).
 *   field_stream: NULL, or a second stream on which the field kernel is launched (event-ordered with `stream`).
 * The call returns when the frame is complete (it ends with one stream synchronise to read the sample count and
 * the per-iteration record back). */
int ced_render_image_test(const ced_field_desc *field, int64_t n_rays, const float *rays_o, const float *rays_d,
                          const uint8_t *binaries, int32_t n_grids, int32_t res, const float *aabbs, const void *accel,
                          float near_plane, float far_plane, float render_step_size, float cone_angle,
                          float early_stop_eps, int32_t max_samples,
                          const float *timestamps, int32_t t_per_ray, const float *bkgd,
                          float *rgb, float *opacity, float *depth,
                          void *workspace, int64_t workspace_bytes, int64_t *host_stats,
                          int64_t *total_samples_out, ced_frame_trace *trace, void *field_stream, void *stream);

/* Several frames of a video in ONE call: the reference renders them one after another with
 * render_image_test (train_real.py:531-558); here `n_frames` (1..64) frames of `rays_per_frame` rays each share the
 * launches of an iteration -- one marching, one field and one compositing launch cover the alive rays of all the
 * frames -- while EVERY FRAME KEEPS ITS OWN reference loop: N_samples = clamp(N_rays // N_alive, min, 64) on its own
 * counts, its own `iteration < max_samples` end.  Each frame's pixels and sample count are therefore exactly those of
 * ced_render_image_test on that frame alone; what changes is that the launches are n_frames times larger.
 *   rays_o, rays_d [n_frames * rays_per_frame, 3] (frame-major); frame_times: device [n_frames], one time per frame;
 *   outputs rgb [n_frames * rays_per_frame, 3], opacity, depth; workspace of
 *   ced_render_frames_test_workspace_bytes() bytes; total_samples_out: host [n_frames].
 *   Everything else as ced_render_image_test. */
int64_t ced_render_frames_test_workspace_bytes(int32_t n_frames, int64_t rays_per_frame, int32_t n_grids, int32_t res,
                                               float cone_angle, int32_t max_samples);
int ced_render_frames_test(const ced_field_desc *field, int32_t n_frames, int64_t rays_per_frame,
                           const float *rays_o, const float *rays_d, const uint8_t *binaries, int32_t n_grids,
                           int32_t res, const float *aabbs, const void *accel, float near_plane, float far_plane,
                           float render_step_size, float cone_angle, float early_stop_eps, int32_t max_samples,
                           const float *frame_times, const float *bkgd, float *rgb, float *opacity, float *depth,
                           void *workspace, int64_t workspace_bytes, int64_t *host_stats,
                           int64_t *total_samples_out, ced_frame_trace *trace, void *field_stream, void *stream);

/* ---- frames whose rays are sharded over several processes (one process per GPU) ----
 * The loop of render_image_test is ONE loop per image: N_samples = clamp(N_rays // N_alive, min, 64)
 * (cednerf/utils.py:231-235) counts the rays of the whole image, and a ray stays alive on `packed_info[:, 1] ==
 * N_samples` (utils.py:301-306).  When frame f's rays are dealt over several processes, every process must therefore
 * run the IMAGE's schedule, not one of its own shard: per iteration the processes exchange, per frame, the number of
 * their rays that survived (one int64 per frame), and the scheduling launch computes N_samples, the last-iteration
 * flag and the end of the loop from the sums.  Each ray then receives exactly the samples, termination planes and
 * pixel values it receives when a single process renders the whole image (bit for bit), whatever the sharding.
 *
 * The exchange is the caller's (the library links no communication library): `reduce` is called by the rendering
 * thread once per enqueued iteration, between that iteration's compositing launch and its scheduling launch, and must
 * ENQUEUE on `stream` an in-place sum over the processes of counts[0 .. n_counts) (device memory, int64; e.g.
 * ncclAllReduce / torch.distributed.all_reduce on that stream) and return 0, or non-zero to abort the call.  It must
 * not block on the device.  Every process makes the same number of calls in the same order: the host loop ends on the
 * plan of a fixed earlier iteration (CED_FRAME_RUN_AHEAD behind), which is identical on all processes, never on
 * timing.  No kernel of this library waits for another process.
 *   global_rays_per_frame: N_rays of the whole image; local_rays: DEVICE int32 [n_frames] or NULL -- how many of the
 *   frame's `rays_per_frame` local slots hold real rays (shards are padded to a common size; padding is never alive);
 *   counts: DEVICE int64 [(iterations + 1) * n_frames] scratch, iterations = ced_render_frames_test_iterations();
 *   host_stats must then hold ced_render_frames_test_host_bytes() bytes of pinned memory. */
typedef int (*ced_exchange_fn)(void *user, int64_t *counts, int32_t n_counts, int32_t iteration, void *stream);
typedef struct ced_shard_exchange {
    int64_t global_rays_per_frame;
    const int32_t *local_rays;
    int64_t *counts;
    ced_exchange_fn reduce;
    void *user;
} ced_shard_exchange;
/* device bytes of `workspace` for a sharded call: a share's rays may hold any part of the image's alive rays, so its
 * sample arrays are sized for min(rays_per_frame * 64, global_rays_per_frame) samples per frame and iteration */
int64_t ced_render_frames_test_sharded_workspace_bytes(int32_t n_frames, int64_t rays_per_frame,
                                                       int64_t global_rays_per_frame, int32_t n_grids, int32_t res,
                                                       float cone_angle, int32_t max_samples);
/* most iterations the loop can take (each uses at least min_samples of the max_samples budget) */
int32_t ced_render_frames_test_iterations(float cone_angle, int32_t max_samples);
/* pinned bytes `host_stats` needs for a sharded call (per-iteration publication records) */
int64_t ced_render_frames_test_host_bytes(float cone_angle, int32_t max_samples);
/* ced_render_frames_test with `exchange` (NULL: identical to ced_render_frames_test). */
int ced_render_frames_test_sharded(const ced_field_desc *field, int32_t n_frames, int64_t rays_per_frame,
                                   const float *rays_o, const float *rays_d, const uint8_t *binaries, int32_t n_grids,
                                   int32_t res, const float *aabbs, const void *accel, float near_plane, float far_plane,
                                   float render_step_size, float cone_angle, float early_stop_eps, int32_t max_samples,
                                   const float *frame_times, const float *bkgd, float *rgb, float *opacity, float *depth,
                                   void *workspace, int64_t workspace_bytes, int64_t *host_stats,
                                   int64_t *total_samples_out, ced_frame_trace *trace, void *field_stream, void *stream,
                                   const ced_shard_exchange *exchange);

/* ---- render_image in eval mode: cednerf/utils.py:46-150 (`estimator.sampling` with sigma_fn, then `rendering`) ----
 * The reference evaluates the density of EVERY marched sample (nerfacc OccGridEstimator.sampling ->
 * render_visibility_from_density, call site cednerf/utils.py:115-125) and then the whole field again on the survivors
 * (cednerf/render.py:81-87).  This entry returns the same arrays, bit for bit, from ONE field evaluation per sample:
 * each ray's samples are walked front to back in chunks and a ray stops at the first sample whose transmittance so
 * far is below early_stop_eps (the kept samples of a ray are a prefix of its march, so the rest can never be kept).
 * Inputs: the one-shot march of all rays -- packed_info [n_rays, 2] (first sample, count), t_starts / t_ends [n_all],
 * as ced_traverse_grids produces them (samples sorted by ray).  timestamps as in ced_field_forward_rays.
 * Outputs per ray: rgb [n_rays, 3], opacity, depth (finalised as cednerf/render.py:158-176: background blend, depth /
 * max(opacity, eps)), kept [n_rays] = kept samples of the ray.  stats_out (host int64 [3]): [0] samples evaluated
 * (entries of the workspace's sample arrays), [1] iterations, [2] kept samples (= sum of `kept`).  host_stats: PINNED host memory, >= 64 bytes (the
 * iteration schedule is computed on the device and published there, as in ced_render_image_test).  The call returns
 * after one stream synchronisation (the sample total is needed to size the outputs of the gather).
 * ced_render_image_gather then writes the per-sample `extras` of the kept samples in the reference's order (by ray,
 * then along the ray): ray_offsets [n_rays] = exclusive prefix sum of `kept` (int64), outputs of sum(kept) entries;
 * chunk_rays > 0 makes ray_indices relative to the ray's chunk of that many rays (the reference's chunked eval loop,
 * cednerf/utils.py:108-133), 0 keeps them absolute.
 * Sampling only (rgb == opacity == depth == NULL; the training step's `estimator.sampling(sigma_fn=...)`,
 * train_real.py:339-350, per-ray timestamps): the field kernel evaluates the density alone and no pixel sums are formed;
 * the gather then takes sigmas optional and rgbs == weights == trans == alphas == NULL and returns the surviving
 * (ray_indices, t_starts, t_ends) of nerfacc's sampling, bit for bit. */
int64_t ced_render_image_workspace_bytes(int64_t n_rays, int64_t n_all);
int ced_render_image(const ced_field_desc *field, int64_t n_rays, const float *rays_o, const float *rays_d,
                     int64_t n_all, const int64_t *packed_info, const float *t_starts, const float *t_ends,
                     float early_stop_eps, float alpha_thre, const float *timestamps, int32_t t_per_ray,
                     const float *bkgd, float *rgb, float *opacity, float *depth, int32_t *kept, void *workspace,
                     int64_t workspace_bytes, int64_t *host_stats, int64_t *stats_out, void *field_stream, void *stream);
int ced_render_image_gather(int64_t n_rays, int64_t n_all, int64_t processed, const void *workspace,
                            int64_t workspace_bytes, const int64_t *ray_offsets, int64_t chunk_rays,
                            int64_t *ray_indices, float *t_starts, float *t_ends, float *sigmas, float *rgbs,
                            float *weights, float *trans, float *alphas, void *stream);

/* One-shot march of every ray to the far plane on the accelerated walk of the frame renderer: the samples of
 * nerfacc.traverse_grids without a step limit (the marching half of OccGridEstimator.sampling, call sites
 * cednerf/utils.py:115-125, train_real.py:339-350), bit for bit, faster than ced_traverse_grids through empty space.
 * accel: ced_build_occupancy_accel's structure.  t_sorted [n_rays, 2 n_grids], t_indices (int64), hits [n_rays, n_grids]:
 * the sorted ray / box events of cednerf/utils.py:215-225 (ced_ray_aabb_intersect + stable sort); NULL for one level.
 * fill = 0: packed_info[r][1] = samples of ray r.  The caller scans the counts into packed_info[r][0].
 * fill = 1: writes t_starts / t_ends (and ray_indices, optional) of ray r from packed_info[r][0] on.
 * fill = 2: ONE pass for callers that can bound the total: `total` (device int64, zero on entry) counts the samples,
 *   every ray gets a range [packed_info[r][0], + packed_info[r][1]) of t_starts / t_ends [capacity] in workgroup
 *   arrival order (NOT sorted by ray; ray_indices unused) and stores its samples there if the range ends within
 *   `capacity`; *total > capacity afterwards means some rays stored nothing (redo with the two-pass form).
 *   ced_render_image accepts such a march (it treats a range beyond n_all as empty and never reads past the arrays). */
int ced_march_all(int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *binaries, int32_t n_grids,
                  int32_t res, const float *aabbs, const void *accel, const float *near_planes, float far_plane,
                  float render_step_size, float cone_angle, const float *t_sorted, const int64_t *t_indices,
                  const uint8_t *hits, int32_t fill, int64_t *packed_info, float *t_starts, float *t_ends,
                  int64_t *ray_indices, int64_t capacity, int64_t *total, void *stream);

/* A whole bias-free ReLU MLP (widths <= 64, <= 6 layers) in one launch per direction: the fused forward / backward of
 * the tiny-cuda-nn FullyFusedMLP networks the reference trains through (cednerf/model.py:200-222,280-344 under
 * train_real.py:339-420).  Same per-output MFMA order as ced_linear: the results are those of the layer-by-layer calls.
 * widths (host) [n_layers + 1]: input width, then each layer's output width; weights (host array of device pointers):
 * W_l [widths[l+1], widths[l]] row-major.
 * backward = 0: x [n, widths[0]]; outs[l] (device, [n, widths[l+1]]) receives layer l's output -- ReLU applied on all but
 *   the last layer (on the last too with relu_last); masks unused.
 * backward = 1: x = dy [n, widths[n_layers]]; going down from the last layer, outs[l] ([n, widths[l]], may be NULL)
 *   receives the gradient with respect to layer l's INPUT, multiplied by [masks[l] > 0] where masks[l] (that input, i.e.
 *   the forward's outs[l-1]) is given -- masks[0] is normally NULL (the network input has no ReLU). */
int ced_mlp_chain(int64_t n, int32_t n_layers, int32_t backward, const float *x, const int32_t *widths,
                  const float *const *weights, float *const *outs, const float *const *masks, int32_t relu_last,
                  void *stream);

/* Backward of a whole MLP WITH its weight gradients, one launch (+ a fixed-order reduction): the walk of ced_mlp_chain's
 * backward direction, with dW_l = dz_l^T a_l accumulated at every layer from the registers the walk holds (instead of
 * ced_weight_grad re-reading every dz_l and a_l).  Shapes: input width <= 48, 1..3 hidden layers of width 64, output
 * width <= 32 (n_layers = hidden layers + 1); CED_E_INVALID otherwise (callers fall back to ced_mlp_chain +
 * ced_weight_grad).  widths as ced_mlp_chain; acts (host array of device pointers) [n_layers]: acts[0] = the network
 * input x [n, widths[0]], acts[l] = the forward's (post-ReLU) output of layer l-1 [n, 64]; dy [n, widths[n_layers]].
 * Outputs: dws (device, sum of widths[l] * widths[l+1] floats): dW_0, dW_1, ... back to back, each row-major
 * [widths[l+1], widths[l]]; g0 [n, widths[0]] (gradient w.r.t. x) or NULL.  Deterministic (no float atomics). */
int64_t ced_mlp_backward_dw_workspace_bytes(int64_t n, int32_t n_layers, const int32_t *widths);
int ced_mlp_backward_dw(int64_t n, int32_t n_layers, const float *dy, const int32_t *widths,
                        const float *const *weights, const float *const *acts, float *g0, float *dws, void *workspace,
                        int64_t workspace_bytes, void *stream);

/* Element-wise pieces of the DIFFERENTIABLE field between its MLPs and its hash grid (training path; the statements of
 * cednerf/model.py:354-383,414-417,447-455 as train_real.py:339-420 differentiates them), one launch each.  fp32, device
 * pointers, row-major.
 * ced_train_inputs: per sample the position (rays mode, ray_indices != NULL: rays_o[r] + rays_d[r] * ((t_start + t_end) / 2),
 *   time timestamps[r]; explicit mode: positions / directions / timestamps per sample), enc_out [n, 32] = tcnn
 *   Frequency(4) of (x, y, z, t) ([dim][freq][sin, cos] of pi 2^k v), sh_out [n, 4] = SH degree 2 of the normalised
 *   direction, t_out [n].  No backward: nothing upstream has parameters. */
int ced_train_inputs(int64_t n, const float *rays_o, const float *rays_d, const int64_t *ray_indices,
                     const float *t_starts, const float *t_ends, const float *timestamps, const float *positions,
                     const float *directions, float *pos_out, float *enc_out, float *sh_out, float *t_out, void *stream);
/* ced_train_warp: move = mo[:, :3] * moving_step (+ tanh(mo[:, 3:6]) * moving_step with use_div_offsets),
 *   xn = clamp((pos + move - aabb_lo) / (aabb_hi - aabb_lo), 0, 1), selector = 1 where the unclamped xn lies strictly
 *   inside (0, 1) on all axes, else 0.  aabb_host: six floats on the HOST.
 * ced_train_warp_backward: d_mo [n, mo_width] from d_xn (passes where 0 <= unclamped xn <= 1, torch.clamp's rule) and
 *   d_move (may be NULL). */
int ced_train_warp(int64_t n, const float *pos, const float *mo, int32_t mo_width, int32_t use_div_offsets,
                   float moving_step, const float *aabb_host, float *xn, float *move, float *selector, void *stream);
int ced_train_warp_backward(int64_t n, const float *pos, const float *mo, int32_t mo_width, int32_t use_div_offsets,
                            float moving_step, const float *aabb_host, const float *d_xn, const float *d_move,
                            float *d_mo, void *stream);
/* ced_train_head_in: sigma = exp(bout[:, 0] - 1) * selector (trunc_exp, cednerf/utils.py:27-43), head_in [n, 19] =
 *   [sh (4), bout[:, 1:16]].  ced_train_head_in_backward: d_bout [n, 16] from d_head_in [n, 19] and d_sigma [n] (either may
 *   be NULL = zero); the density's derivative is exp(min(raw - 1, 15)) as in the reference's _TruncExp.backward. */
int ced_train_head_in(int64_t n, const float *bout, const float *sh, const float *selector, float *head_in,
                      float *sigma, void *stream);
int ced_train_head_in_backward(int64_t n, const float *bout, const float *selector, const float *d_head_in,
                               const float *d_sigma, float *d_bout, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CEDNERF_HIP_H */
