/*
 * cednerf_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C, scalar, deterministic restatement of the Ced-NeRF rendering hot path
 * (SURVEY.md section 8).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (ced_nerf_amd/) never does.
 *
 * PARITY PINNING: the reference holds no tests/fixtures for this path and its arithmetic
 * lives in un-vendored CUDA packages (nerfacc >= 0.5.3, tiny-cuda-nn master, both unpinned
 * and absent from /root/reference), so at those boundaries this oracle is "parity unpinned".
 * What pins it: golden vectors captured from the importable cednerf/encoder.py
 * (tests/golden/), analytic known-answer tests and an independent
 * PyTorch fp32 restatement (oracle/torch_oracle.py).
 *
 * Arithmetic contract (shared with the HIP kernels, which restate it independently):
 *   - IEEE binary32 everywhere, round-to-nearest-even, subnormals kept;
 *   - compiled with -ffp-contract=off: an FMA happens only where fmaf() is written;
 *   - +,-,*,/,sqrtf are the correctly rounded IEEE operations;
 *   - exp / sin / cos are the explicit polynomial kernels below (no libm calls);
 *   - every dot product is an ascending-k fmaf chain starting from +0.0f;
 *   - per-ray sums run sequentially in sample order.
 *
 * Each function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mfma_f16_model.h"     /* what the fp16-operand matrix instruction of gfx950 computes (measured) */

#define CED_MAX_LEVELS 16

/* ------------------------------------------------------------------------------------------
 * scalar math kernels
 * ---------------------------------------------------------------------------------------- */
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* 2^n for integer n in [-126, 127] built from the exponent field. */
static inline float pow2i(int n) { return u2f((uint32_t)(n + 127) << 23); }

/* exp(x): n = rint(x*log2e); r = x - n*ln2 (two-step Cody-Waite); degree-6 Taylor in r;
 * scaled by 2^n in two exact steps so subnormal results round once.
 * Stands in for torch.exp in trunc_exp (cednerf/utils.py:27-43) and in
 * render_weight_from_density (cednerf/render.py:81-87). */
float ced_o_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -103.97283935546875f) return 0.0f;
    float n = rintf(x * 1.44269502162933349609375f);
    float r = fmaf(-n, 0.693145751953125f, x);
    r = fmaf(-n, 1.428606765330187045e-06f, r);
    float p = 1.38888892e-3f;           /* 1/720 */
    p = fmaf(p, r, 8.33333377e-3f);     /* 1/120 */
    p = fmaf(p, r, 4.16666679e-2f);     /* 1/24  */
    p = fmaf(p, r, 1.66666672e-1f);     /* 1/6   */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    return (p * pow2i(n1)) * pow2i(n2);
}

/* Taylor kernels on |x| <= pi/4. */
static inline float sin_kernel(float x)
{
    float x2 = x * x;
    float p = 2.75573192e-6f;           /* 1/362880 */
    p = fmaf(p, x2, -1.98412701e-4f);   /* -1/5040  */
    p = fmaf(p, x2, 8.33333377e-3f);    /* 1/120    */
    p = fmaf(p, x2, -1.66666672e-1f);   /* -1/6     */
    return fmaf(x * x2, p, x);
}
static inline float cos_kernel(float x)
{
    float x2 = x * x;
    float p = -2.75573188e-7f;          /* -1/3628800 */
    p = fmaf(p, x2, 2.48015876e-5f);    /* 1/40320    */
    p = fmaf(p, x2, -1.38888892e-3f);   /* -1/720     */
    p = fmaf(p, x2, 4.16666679e-2f);    /* 1/24       */
    p = fmaf(p, x2, -0.5f);
    return fmaf(p, x2, 1.0f);
}
static inline float quadrant_select(int q, float x)
{
    switch (q & 3) {
    case 0: return sin_kernel(x);
    case 1: return cos_kernel(x);
    case 2: return -sin_kernel(x);
    default: return -cos_kernel(x);
    }
}

/* sin(pi*y + phase*pi/2), exact argument reduction: n = rint(2y), r = y - n/2 (exact for
 * |y| < 2^22), x = pi*r.  tcnn Frequency encoding: sin(2^k*pi*x), phase 1 is
 * its "+pi/2" (cosine) twin. */
float ced_o_sinpi_phase(float y, int phase)
{
    float n = rintf(y + y);
    float r = y - 0.5f * n;
    float x = 3.14159274101257324f * r;
    int q = (int)((long long)n & 3LL) + phase;
    return quadrant_select(q, x);
}

/* sin(x) for |x| < ~1e4: n = rint(x*2/pi); r = x - n*pi/2 (two-step Cody-Waite).
 * Stands in for torch.sin in cednerf/encoder.py:41,83. */
float ced_o_sinf(float x)
{
    float n = rintf(x * 0.636619746685028076f);
    float r = fmaf(-n, 1.57079625129699707f, x);
    r = fmaf(-n, 7.54978941586159635e-08f, r);
    return quadrant_select((int)((long long)n & 3LL), r);
}

/* ------------------------------------------------------------------------------------------
 * a14: ray_aabb_intersect  [nerfacc, un-vendored; call site cednerf/utils.py:215]
 * slab test with inv_dir, per (ray, aabb); misses -> miss_value.
 * ---------------------------------------------------------------------------------------- */
static int ray_aabb_one(const float *o, const float *inv_d, const float *aabb,
                        float near, float far, float *tmin_out, float *tmax_out)
{
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    if (inv_d[0] >= 0) { tmin = (aabb[0] - o[0]) * inv_d[0]; tmax = (aabb[3] - o[0]) * inv_d[0]; }
    else               { tmin = (aabb[3] - o[0]) * inv_d[0]; tmax = (aabb[0] - o[0]) * inv_d[0]; }
    if (inv_d[1] >= 0) { tymin = (aabb[1] - o[1]) * inv_d[1]; tymax = (aabb[4] - o[1]) * inv_d[1]; }
    else               { tymin = (aabb[4] - o[1]) * inv_d[1]; tymax = (aabb[1] - o[1]) * inv_d[1]; }
    if (tmin > tymax || tymin > tmax) return 0;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    if (inv_d[2] >= 0) { tzmin = (aabb[2] - o[2]) * inv_d[2]; tzmax = (aabb[5] - o[2]) * inv_d[2]; }
    else               { tzmin = (aabb[5] - o[2]) * inv_d[2]; tzmax = (aabb[2] - o[2]) * inv_d[2]; }
    if (tmin > tzmax || tzmin > tmax) return 0;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    if (tmax <= 0) return 0;
    *tmin_out = fmaxf(tmin, near);
    *tmax_out = fminf(tmax, far);
    return 1;
}

void ced_o_ray_aabb_intersect(int64_t n_rays, const float *rays_o, const float *rays_d,
                              int n_aabbs, const float *aabbs, float near, float far,
                              float miss_value, float *t_mins, float *t_maxs, uint8_t *hits)
{
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n_rays; ++r) {
        const float *o = rays_o + 3 * r;
        float inv_d[3] = { 1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2] };
        for (int a = 0; a < n_aabbs; ++a) {
            float t0 = miss_value, t1 = miss_value;
            int hit = ray_aabb_one(o, inv_d, aabbs + 6 * a, near, far, &t0, &t1);
            if (!hit) { t0 = miss_value; t1 = miss_value; }
            t_mins[r * n_aabbs + a] = t0;
            t_maxs[r * n_aabbs + a] = t1;
            hits[r * n_aabbs + a] = (uint8_t)hit;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a14: traverse_grids  [nerfacc, un-vendored; call sites cednerf/utils.py:241-264 and, through
 * OccGridEstimator.sampling, cednerf/utils.py:115-125].  One logical thread per ray.
 * mode 0: count only (out arrays may be NULL); mode 1: fill at packed offset base[r];
 * over-allocation is mode 1 with base[r] = r * limit.
 * ---------------------------------------------------------------------------------------- */
static inline float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    float v = t * cone_angle;
    return fminf(fmaxf(v, dt_min), dt_max);
}

/* the empty-space skip of traverse_grids on its own (sequential recurrence), for tests */
float ced_o_skip_march(float t_last, float target, float step_size, float cone_angle)
{
    if (step_size <= 0.0f) return target;
    for (;;) {
        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
        if (t_last + dt * 0.5f >= target) break;
        t_last += dt;
    }
    return t_last;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

static int64_t traverse_one_ray(
    const float *o, const float *d, const uint8_t *binaries, int n_grids, int res,
    const float *aabbs, float near, float far, float step_size, float cone_angle,
    int limit, const float *t_sorted, const int64_t *t_indices, const uint8_t *hits,
    float *t_starts, float *t_ends, float *t_term)
{
    const float eps = 1e-6f;
    float inv_d[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    float t_last = near;
    int continuous = 0;
    int64_t n = 0;
    const float resf = (float)res;
    for (int i = 0; i < 2 * n_grids - 1; ++i) {
        int entering = t_indices[i] < n_grids;
        int lvl = (int)(t_indices[i] % n_grids);
        if (!hits[lvl]) continue;
        if (!entering) {
            int next_entering = t_indices[i + 1] < n_grids;
            if (next_entering) continue;
            lvl = (int)(t_indices[i + 1] % n_grids);
            if (!hits[lvl]) continue;
        }
        float this_tmin = fmaxf(t_sorted[i], near);
        float this_tmax = fminf(t_sorted[i + 1], far);
        if (this_tmin >= this_tmax) continue;
        if (!continuous) {
            if (step_size <= 0.0f) {
                t_last = this_tmin;
            } else {
                for (;;) {
                    float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                    if (t_last + dt * 0.5f >= this_tmin) break;
                    t_last += dt;
                }
            }
        }
        /* DDA setup on grid level lvl */
        const float *ab = aabbs + 6 * lvl;
        float vox[3], tdist[3], delta[3];
        int cur[3], fin[3], stp[3], ovf[3];
        float ts = this_tmin + eps, te = this_tmax - eps;
        for (int a = 0; a < 3; ++a) {
            float ext = ab[3 + a] - ab[a];
            vox[a] = ext / resf;
            float ps = o[a] + d[a] * ts;
            float pe = o[a] + d[a] * te;
            cur[a] = clampi((int)(((ps - ab[a]) / ext) * resf), 0, res - 1);
            fin[a] = clampi((int)(((pe - ab[a]) / ext) * resf), 0, res - 1);
            int idelta = d[a] > 0.0f ? 1 : 0;
            float tm = ((ab[a] + (((float)(cur[a] + idelta) * vox[a]) - ps)) * inv_d[a]) + this_tmin;
            float stepf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
            stp[a] = (int)stepf;
            tdist[a] = (d[a] == 0.0f) ? this_tmax : tm;
            delta[a] = (d[a] == 0.0f) ? this_tmax : (vox[a] * inv_d[a]) * stepf;
            ovf[a] = fin[a] + stp[a];
        }
        const uint8_t *grid = binaries + (int64_t)lvl * res * res * res;
        while (limit <= 0 || n < limit) {
            float t_trav = fminf(fminf(tdist[0], fminf(tdist[1], tdist[2])), this_tmax);
            int64_t cell = ((int64_t)cur[0] * res + cur[1]) * res + cur[2];
            if (!grid[cell]) {
                if (step_size <= 0.0f) {
                    t_last = t_trav;
                } else {
                    for (;;) {
                        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_trav) break;
                        t_last += dt;
                    }
                }
                continuous = 0;
            } else {
                while (limit <= 0 || n < limit) {
                    float t_next;
                    if (step_size <= 0.0f) {
                        t_next = t_trav;
                    } else {
                        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_trav) break;
                        t_next = t_last + dt;
                    }
                    if (t_starts) { t_starts[n] = t_last; t_ends[n] = t_next; }
                    n += 1;
                    continuous = 1;
                    t_last = t_next;
                    if (t_next >= t_trav) break;
                }
            }
            /* advance to the next voxel */
            int ax;
            if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) ax = 0;
            else if (tdist[1] < tdist[2]) ax = 1;
            else ax = 2;
            cur[ax] += stp[ax];
            tdist[ax] += delta[ax];
            if (cur[ax] == ovf[ax]) break;
        }
    }
    *t_term = t_last;
    return n;
}

void ced_o_traverse_grids(
    int64_t n_rays, const float *rays_o, const float *rays_d,
    const uint8_t *binaries, int n_grids, int res, const float *aabbs,
    const float *near_planes, const float *far_planes, float step_size, float cone_angle,
    int limit, const uint8_t *rays_mask /* may be NULL */,
    const float *t_sorted, const int64_t *t_indices, const uint8_t *hits,
    int mode, const int64_t *base /* mode 1 */,
    int64_t *counts, float *t_starts, float *t_ends, float *termination_planes)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rays; ++r) {
        if (rays_mask && !rays_mask[r]) {
            counts[r] = 0;
            /* masked rays keep their near plane as the termination plane */
            if (termination_planes) termination_planes[r] = near_planes[r];
            continue;
        }
        float tt;
        float *ts = NULL, *te = NULL;
        if (mode == 1) { ts = t_starts + base[r]; te = t_ends + base[r]; }
        counts[r] = traverse_one_ray(rays_o + 3 * r, rays_d + 3 * r, binaries, n_grids, res, aabbs,
                                     near_planes[r], far_planes[r], step_size, cone_angle, limit,
                                     t_sorted + r * 2 * n_grids, t_indices + r * 2 * n_grids,
                                     hits + r * n_grids, ts, te, &tt);
        if (termination_planes) termination_planes[r] = tt;
    }
}

/* ------------------------------------------------------------------------------------------
 * a7/a8: multi-resolution hash grid.  Spec: cednerf/taichi_kernel/hash_encoder_half.py:67-107
 * (index), :112-161 (trilinear gather), hash_encoder_inter.py:148-197 (temporal variant).
 * Level tables (scale/res/offset/size/hashed) come from the host (float64).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_levels;
    int32_t table_dtype;           /* 0 = fp32 entries, 1 = fp16 entries */
    int32_t temporal;              /* 0: entry = 2 feats; 1: entry = 4 key-frames x 2 feats */
    int32_t pad_;
    float scale[CED_MAX_LEVELS];
    uint32_t res[CED_MAX_LEVELS];
    uint32_t offset[CED_MAX_LEVELS];
    uint32_t size[CED_MAX_LEVELS];
    uint32_t hashed[CED_MAX_LEVELS];
    const void *table;
} ced_o_hash_t;

static inline float half_to_float(uint16_t h)
{
    uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(s);
        /* subnormal half: value = m * 2^-24 */
        float v = (float)m * 5.9604644775390625e-08f;
        return (s ? -v : v);
    }
    if (e == 31) return u2f(s | 0x7f800000u | (m << 13));
    return u2f(s | ((e + 112u) << 23) | (m << 13));
}

static inline float table_read(const ced_o_hash_t *h, uint64_t entry, int width, int comp)
{
    uint64_t i = entry * (uint64_t)width + (uint64_t)comp;
    if (h->table_dtype == 0) return ((const float *)h->table)[i];
    return half_to_float(((const uint16_t *)h->table)[i]);
}

static inline uint32_t grid_index(const ced_o_hash_t *h, int l, uint32_t x, uint32_t y, uint32_t z)
{
    uint32_t idx;
    if (h->hashed[l]) {
        idx = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);   /* hash_encoder_half.py:67-74 */
    } else {
        uint32_t r = h->res[l];
        idx = x + y * r + z * r * r;                              /* hash_encoder_half.py:77-83 */
    }
    return idx % h->size[l];                                      /* hash_encoder_half.py:94 */
}

/* x in [0,1]^3 (clamped here), tq = time in [0,1] (temporal only); out[2*n_levels]. */
static void hash_encode_one(const ced_o_hash_t *h, const float *x_in, float tq, float *out)
{
    float x[3];
    for (int a = 0; a < 3; ++a) x[a] = fminf(fmaxf(x_in[a], 0.0f), 1.0f);
    int width = h->temporal ? 8 : 2;
    int k_lo = 0;
    float t_frac = 0.0f;
    if (h->temporal) {                                            /* hash_encoder_inter.py:148-160 */
        float ts = tq * 3.0f;
        float fl = floorf(ts);
        t_frac = ts - fl;
        fl = fminf(fl, 2.0f);
        k_lo = (int)fl;
    }
    for (int l = 0; l < h->n_levels; ++l) {
        float sc = h->scale[l];
        uint32_t g[3];
        float fr[3], om[3];
        for (int a = 0; a < 3; ++a) {
            float pos = x[a] * sc + 0.5f;                         /* hash_encoder_half.py:131 */
            float fl = floorf(pos);
            g[a] = (uint32_t)fl;
            fr[a] = pos - fl;
            om[a] = 1.0f - fr[a];
        }
        float acc0 = 0.0f, acc1 = 0.0f;
        for (int c = 0; c < 8; ++c) {                             /* hash_encoder_half.py:137-159 */
            float w = 1.0f;
            uint32_t p[3];
            for (int a = 0; a < 3; ++a) {
                if ((c & (1 << a)) == 0) { p[a] = g[a]; w = w * om[a]; }
                else { p[a] = g[a] + 1u; w = w * fr[a]; }
            }
            uint64_t e = (uint64_t)h->offset[l] + grid_index(h, l, p[0], p[1], p[2]);
            float f0, f1;
            if (!h->temporal) {
                f0 = table_read(h, e, width, 0);
                f1 = table_read(h, e, width, 1);
            } else {                                              /* hash_encoder_inter.py:185-193 */
                float a0 = table_read(h, e, width, 2 * k_lo), a1 = table_read(h, e, width, 2 * k_lo + 1);
                float b0 = table_read(h, e, width, 2 * k_lo + 2), b1 = table_read(h, e, width, 2 * k_lo + 3);
                float omt = 1.0f - t_frac;
                f0 = a0 * omt + b0 * t_frac;
                f1 = a1 * omt + b1 * t_frac;
            }
            acc0 = fmaf(w, f0, acc0);
            acc1 = fmaf(w, f1, acc1);
        }
        out[2 * l] = acc0;
        out[2 * l + 1] = acc1;
    }
}

void ced_o_hash_encode(const ced_o_hash_t *h, int64_t n, const float *x, const float *t /* may be NULL */,
                       float *out)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        hash_encode_one(h, x + 3 * i, t ? t[i] : 0.0f, out + (int64_t)2 * h->n_levels * i);
}

/* Backward of the (non-temporal) hash encode, hash_encoder_half.py:164-226 (next row f2: training path).
 *   dy [n][L][2]          gradient w.r.t. the encoder output
 *   grad_table [E][2]     += w * dy per corner (accumulated here in DOUBLE, in sample/level/corner order: the
 *                         reference adds with atomics in no particular order, so only the sum is specified)
 *   dx [n][3] (optional)  = sum over levels and corners of (table_feat . dy) * dw/dpos, with the reference's
 *                         form dw/dpos_d = w / (+-other factor) (:212-213) and, like the reference, WITHOUT
 *                         the d pos / d x = scale factor unless dx_scaled (the true gradient w.r.t. x). */
void ced_o_hash_encode_backward(const ced_o_hash_t *h, int64_t n, const float *x_in, const float *dy,
                                double *grad_table, float *dx, int dx_scaled /* 1: times d pos / d x = scale */)
{
    for (int64_t i = 0; i < n; ++i) {
        float x[3];
        for (int a = 0; a < 3; ++a) x[a] = fminf(fmaxf(x_in[3 * i + a], 0.0f), 1.0f);
        float gx[3] = { 0.0f, 0.0f, 0.0f };
        for (int l = 0; l < h->n_levels; ++l) {
            const float sc = h->scale[l];
            const float g0 = dy[(i * h->n_levels + l) * 2], g1 = dy[(i * h->n_levels + l) * 2 + 1];
            uint32_t g[3];
            float fr[3], om[3];
            for (int a = 0; a < 3; ++a) {
                float pos = x[a] * sc + 0.5f;
                float fl = floorf(pos);
                g[a] = (uint32_t)fl;
                fr[a] = pos - fl;
                om[a] = 1.0f - fr[a];
            }
            if (g0 == 0.0f && g1 == 0.0f) continue;                       /* grad_dy_temp.any(), :209 */
            for (int c = 0; c < 8; ++c) {
                float w = 1.0f, dwd[3];
                uint32_t p[3];
                for (int a = 0; a < 3; ++a) {
                    if ((c & (1 << a)) == 0) { p[a] = g[a]; w = w * om[a]; dwd[a] = -om[a]; }
                    else { p[a] = g[a] + 1u; w = w * fr[a]; dwd[a] = fr[a]; }
                }
                const uint32_t idx = h->offset[l] + grid_index(h, l, p[0], p[1], p[2]);
                float f0, f1;
                if (h->table_dtype == 0) {
                    const float *tb = (const float *)h->table + (size_t)idx * 2;
                    f0 = tb[0]; f1 = tb[1];
                } else {
                    const uint16_t *tb = (const uint16_t *)h->table + (size_t)idx * 2;
                    f0 = half_to_float(tb[0]); f1 = half_to_float(tb[1]);
                }
                const float dot = f0 * g0 + f1 * g1;
                if (dx) for (int a = 0; a < 3; ++a) gx[a] = gx[a] + (dx_scaled ? sc : 1.0f) * (dot * (w / dwd[a]));
                grad_table[(size_t)idx * 2] += (double)(w * g0);
                grad_table[(size_t)idx * 2 + 1] += (double)(w * g1);
            }
        }
        if (dx) for (int a = 0; a < 3; ++a) dx[3 * i + a] = gx[a];
    }
}

/* Backward of the TEMPORAL hash encode, hash_encoder_inter.py:202-275: table gradient only (the reference's autograd
 * function gives positions none).  grad_table [E][8], accumulated in DOUBLE: per corner the key-frames k and k + 1 of
 * the sample's time receive (w * dy) * (1 - t_frac) and (w * dy) * t_frac, k / t_frac as in the forward (:228-240). */
void ced_o_hash_encode_backward_temporal(const ced_o_hash_t *h, int64_t n, const float *x_in, const float *t,
                                         const float *dy, double *grad_table)
{
    for (int64_t i = 0; i < n; ++i) {
        float x[3];
        for (int a = 0; a < 3; ++a) x[a] = fminf(fmaxf(x_in[3 * i + a], 0.0f), 1.0f);
        float ts = t[i] * 3.0f;
        float fl = floorf(ts);
        const float t_frac = ts - fl;
        fl = fminf(fl, 2.0f);
        const int k_lo = (int)fl;
        for (int l = 0; l < h->n_levels; ++l) {
            const float sc = h->scale[l];
            const float g0 = dy[(i * h->n_levels + l) * 2], g1 = dy[(i * h->n_levels + l) * 2 + 1];
            if (g0 == 0.0f && g1 == 0.0f) continue;                       /* grad_dy_temp.any(), :259 */
            uint32_t g[3];
            float fr[3], om[3];
            for (int a = 0; a < 3; ++a) {
                float pos = x[a] * sc + 0.5f;
                float f = floorf(pos);
                g[a] = (uint32_t)f;
                fr[a] = pos - f;
                om[a] = 1.0f - fr[a];
            }
            for (int c = 0; c < 8; ++c) {
                float w = 1.0f;
                uint32_t p[3];
                for (int a = 0; a < 3; ++a) {
                    if ((c & (1 << a)) == 0) { p[a] = g[a]; w = w * om[a]; }
                    else { p[a] = g[a] + 1u; w = w * fr[a]; }
                }
                const size_t e = (size_t)h->offset[l] + grid_index(h, l, p[0], p[1], p[2]);
                const float wg[2] = { w * g0, w * g1 };
                for (int f = 0; f < 2; ++f) {
                    grad_table[e * 8 + 2 * k_lo + f] += (double)(wg[f] * (1.0f - t_frac));
                    grad_table[e * 8 + 2 * k_lo + 2 + f] += (double)(wg[f] * t_frac);
                }
            }
        }
    }
}

/* corner entry indices (into the table, offset included) of level l for a point: exposed so
 * tests can check the integer part of the lookup bit-exactly. */
void ced_o_hash_indices(const ced_o_hash_t *h, int64_t n, const float *x_in, uint32_t *idx /* [n][L][8] */)
{
    for (int64_t i = 0; i < n; ++i) {
        float x[3];
        for (int a = 0; a < 3; ++a) x[a] = fminf(fmaxf(x_in[3 * i + a], 0.0f), 1.0f);
        for (int l = 0; l < h->n_levels; ++l) {
            uint32_t g[3];
            for (int a = 0; a < 3; ++a) g[a] = (uint32_t)floorf(x[a] * h->scale[l] + 0.5f);
            for (int c = 0; c < 8; ++c)
                idx[(i * h->n_levels + l) * 8 + c] =
                    h->offset[l] + grid_index(h, l, g[0] + (c & 1), g[1] + ((c >> 1) & 1), g[2] + ((c >> 2) & 1));
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a9: time encoders, cednerf/encoder.py:6-44 (SinusoidalEncoder(1,0,4,True)) and :46-90
 * (SinusoidalEncoderWithExp(1,0,4,True)); out[9].
 * ---------------------------------------------------------------------------------------- */
void ced_o_time_encode(float t, float move_norm, int with_exp, float *out)
{
    const float HALF_PI = 1.57079637050628662f;   /* float32(0.5*math.pi), encoder.py:41,83 */
    out[0] = t;
    if (!with_exp) {
        for (int k = 0; k < 4; ++k) {
            float xb = t * (float)(1 << k);
            out[1 + k] = ced_o_sinf(xb);
            out[5 + k] = ced_o_sinf(xb + HALF_PI);
        }
    } else {
        for (int k = 0; k < 4; ++k) {
            float xb = t * (float)(1 << k);
            float att = ced_o_expf(-1.0f * (move_norm * (float)(k * (1 << k))));
            out[1 + 2 * k] = ced_o_sinf(xb) * att;
            out[2 + 2 * k] = ced_o_sinf(xb + HALF_PI) * att;
        }
    }
}

void ced_o_time_encode_batch(int64_t n, const float *t, const float *move_norm, int with_exp, float *out)
{
    for (int64_t i = 0; i < n; ++i) ced_o_time_encode(t[i], move_norm ? move_norm[i] : 0.0f, with_exp, out + 9 * i);
}

/* ------------------------------------------------------------------------------------------
 * a6/a10/a11/a12: DNGPradianceField.forward, cednerf/model.py:354-488 (query_move :354-365,
 * query_density :367-445, _query_rgb :447-466); MLPs are bias-free ReLU, weights
 * W[out][in] row-major fp32.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float aabb[6];
    float moving_step;
    int32_t use_div_offsets;       /* motion head has 6 outputs, fine = tanh (model.py:356-358) */
    int32_t time_mode;             /* 0: none; 1: SinusoidalEncoder; 2: WithExp (model.py:386-396) */
    int32_t base_in;               /* 32 or 41 */
    const float *m_w0, *m_w1, *m_w2, *m_w3;   /* 64x32, 64x64, 64x64, (3|6)x64 */
    const float *b_w0, *b_w1;                 /* 64xbase_in, 16x64 */
    const float *h_w0, *h_w1, *h_w2;          /* 64x19, 64x64, 3x64 */
    ced_o_hash_t hash;
    int32_t mlp_half;              /* MLP arithmetic (the library's ced_field_desc.mlp_precision):
                                      0: fp32, ascending-k fmaf chains (the contract in this file's header);
                                      1: "f16"   -- the class of tcnn's FullyFusedMLP (cednerf/model.py:200-222,280-309;
                                         SURVEY A.8): weights and every layer input rounded to fp16, products summed in
                                         fp32 BY THE MATRIX INSTRUCTION, whose blocking / truncation / rounding is
                                         restated in mfma_f16_model.h;
                                      2: "f16x2" -- every operand split x = hi + lo into two fp16 numbers, three
                                         product groups lo*hi, hi*lo, hi*hi per 32 inputs, same instruction model;
                                      3: "f32+h16x2" -- motion MLP and mlp_base as 0, mlp_head as 2.
                                      The caller passes fp32 weights; they are rounded / split here. */
    int32_t reserved;
    void *half_cache;              /* ced_o_field_prepare(): the weights of the fp16-operand modes, rounded / split and
                                      decomposed once (NULL: dense_half() does it per product -- same results, ~30x slower) */
} ced_o_field_t;

/* x rounded to the nearest fp16 value (ties to even, subnormals kept), saturating at +-65504 */
static float round_f16(float x)
{
    if (x != x) return x;
    float a = fabsf(x), r;
    if (a >= 65504.0f) r = 65504.0f;
    else if (a < 6.103515625e-05f) r = rintf(a * 16777216.0f) / 16777216.0f;
    else {
        int e;
        (void)frexpf(a, &e);                       /* a = m * 2^e, m in [0.5, 1): 11 significant bits -> ulp 2^(e-11) */
        float ulp = ldexpf(1.0f, e - 11);
        r = rintf(a / ulp) * ulp;
    }
    return copysignf(r, x);
}

/* exported for the oracle's own tests (checked against numpy's float16 conversion) */
void ced_o_round_f16(int64_t n, const float *x, float *y)
{
    for (int64_t i = 0; i < n; ++i) y[i] = round_f16(x[i]);
}

static inline float dotf(const float *w, const float *x, int n)
{
    float acc = 0.0f;
    for (int k = 0; k < n; ++k) acc = fmaf(w[k], x[k], acc);
    return acc;
}

/* Which input of a layer sits at operand position k (k = 32*ks + 8*g + e: k-step, lane group, element) of the
 * half-precision kernels, or -1 for a zero pad.  The position decides which block of eight products of the matrix
 * instruction an input falls into (mfma_f16_model.h), hence the rounding.  Restates the library's packing
 * (csrc/field_half.hip: pack_half_layer) -- independently written, same table. */
enum { COL_NATURAL = 0, COL_HASH = 1, COL_HEAD = 2 };
static int half_input_at(int col_map, int k, int n_in)
{
    const int g = (k % 32) / 8, e = k % 8;
    int in = k;
    if (col_map == COL_HASH) {
        if (k < 32) in = 2 * (4 * (e >> 1) + g) + (e & 1);          /* level 4*(e/2) + g, feature e&1 */
        else in = (e < 3 && 4 * e + g <= 8) ? 32 + 4 * e + g : -1;   /* time feature 4e + g */
    } else if (col_map == COL_HEAD) {
        if (e == 0) in = g;                                          /* SH coefficient g */
        else if (e <= 4 && 4 * g + e - 1 < 15) in = 4 + 4 * g + e - 1;   /* geometry feature 4g + e - 1 */
        else in = -1;
    }
    return (in >= 0 && in < n_in) ? in : -1;
}

/* The matrix-instruction model on explicit operands, for the tests that pin it to hardware records
 * (tests/golden/mfma_f16_records.npz): n dot products of n_blocks x 8 fp16-representable operands each, consumed block
 * by block on top of acc[i] (a, b: [n][n_blocks * 8]). */
void ced_o_mfma_f16_dot(int64_t n, int n_blocks, const float *a, const float *b, const float *acc, float *out)
{
    for (int64_t i = 0; i < n; ++i) {
        float v = acc[i];
        for (int q = 0; q < n_blocks; ++q)
            v = mfma_f16_block(v, 8, a + (i * n_blocks + q) * 8, b + (i * n_blocks + q) * 8);
        out[i] = v;
    }
}

/* Debug recorder (tools/probes/mfma_replay.py): every block the SLOW path evaluates, for a replay on the hardware.
 * Not thread-safe: record one sample at a time. */
typedef struct { float acc_in; int32_t n; float a[8], b[8]; float out; } ced_o_block_rec;
static ced_o_block_rec *g_rec = NULL;
static int64_t g_rec_cap = 0, g_rec_n = 0;
void ced_o_record_blocks(ced_o_block_rec *buf, int64_t cap) { g_rec = buf; g_rec_cap = cap; g_rec_n = 0; }
int64_t ced_o_recorded_blocks(void) { return g_rec_n; }

/* One layer on the matrix instruction.  The library issues, per output and per k-step of 32 positions,
 * [lo*hi, hi*lo,] hi*hi -- each as two v_mfma_f32_16x16x16_f16: positions with e < 4, then e >= 4; each instruction
 * consumes lane groups {0,1} then {2,3} as one block of eight products (csrc/field_half_device.hpp: mfma_k32,
 * mlp_layer_h).  clamp_in: the operand is saturated to the fp16 range first (activations and hash features are). */
static void dense_half(const float *w, int n_out, int n_in, const float *x_in, float *y, int relu, int split,
                       int col_map, int ks_n, int clamp_in)
{
    float xh[64], xl[64];
    for (int i = 0; i < n_in; ++i) {
        float v = x_in[i];
        if (clamp_in) v = v > 65504.0f ? 65504.0f : (v < -65504.0f ? -65504.0f : v);
        xh[i] = round_f16(v);
        xl[i] = split ? round_f16(v - xh[i]) : 0.0f;
    }
    for (int o = 0; o < n_out; ++o) {
        const float *wr = w + (int64_t)o * n_in;
        float acc = 0.0f;
        for (int ks = 0; ks < ks_n; ++ks)
            for (int term = split ? 0 : 2; term < 3; ++term)          /* 0: w_lo*x_hi  1: w_hi*x_lo  2: w_hi*x_hi */
                for (int half = 0; half < 2; ++half)
                    for (int gp = 0; gp < 2; ++gp) {
                        float a[8], b[8];
                        int cnt = 0;
                        for (int g = 2 * gp; g < 2 * gp + 2; ++g)
                            for (int e = 4 * half; e < 4 * half + 4; ++e) {
                                const int in = half_input_at(col_map, 32 * ks + 8 * g + e, n_in);
                                if (in < 0) continue;
                                const float wh = round_f16(wr[in]);
                                a[cnt] = term == 0 ? round_f16(wr[in] - wh) : wh;
                                b[cnt] = term == 1 ? xl[in] : xh[in];
                                ++cnt;
                            }
                        const float acc_in = acc;
                        acc = mfma_f16_block(acc, cnt, a, b);
                        if (g_rec && g_rec_n < g_rec_cap) {
                            ced_o_block_rec *R = &g_rec[g_rec_n++];
                            memset(R, 0, sizeof *R);
                            R->acc_in = acc_in; R->n = cnt; R->out = acc;
                            memcpy(R->a, a, sizeof(float) * (size_t)cnt);
                            memcpy(R->b, b, sizeof(float) * (size_t)cnt);
                        }
                    }
        y[o] = relu ? (acc > 0.0f ? acc : 0.0f) : acc;
    }
}

/* ---- the same layer with the weights decomposed once (ced_o_field_prepare) ---- */
typedef struct { int8_t term, cnt; int16_t in[8]; } half_block_t;
typedef struct {
    int n_out, n_in, n_blocks, split;
    half_block_t *blocks;                    /* [n_blocks], the same for every output */
    int16_t *we[2], *wm[2];                  /* [hi, lo][n_out * n_in]: exponent, signed significand (0: zero) */
} half_layer_t;
typedef struct { half_layer_t layer[9]; } half_cache_t;

static void half_layer_build(half_layer_t *L, const float *w, int n_out, int n_in, int split, int col_map)
{
    const int ks_n = (n_in + 31) / 32;
    L->n_out = n_out; L->n_in = n_in; L->split = split;
    L->blocks = (half_block_t *)calloc((size_t)ks_n * 3 * 4, sizeof(half_block_t));
    L->n_blocks = 0;
    for (int ks = 0; ks < ks_n; ++ks)
        for (int term = split ? 0 : 2; term < 3; ++term)
            for (int half = 0; half < 2; ++half)
                for (int gp = 0; gp < 2; ++gp) {
                    half_block_t *B = &L->blocks[L->n_blocks];
                    B->term = (int8_t)term; B->cnt = 0;
                    for (int g = 2 * gp; g < 2 * gp + 2; ++g)
                        for (int e = 4 * half; e < 4 * half + 4; ++e) {
                            const int in = half_input_at(col_map, 32 * ks + 8 * g + e, n_in);
                            if (in >= 0) B->in[B->cnt++] = (int16_t)in;
                        }
                    if (B->cnt) ++L->n_blocks;
                }
    for (int h = 0; h < 2; ++h) {
        L->we[h] = (int16_t *)calloc((size_t)n_out * n_in, sizeof(int16_t));
        L->wm[h] = (int16_t *)calloc((size_t)n_out * n_in, sizeof(int16_t));
    }
    for (int i = 0; i < n_out * n_in; ++i) {
        const float hi = round_f16(w[i]), lo = round_f16(w[i] - hi);
        const float v[2] = { hi, lo };
        for (int h = 0; h < 2; ++h) {
            int e; int32_t m;
            if (mfma_f16_decompose(v[h], &e, &m)) { L->we[h][i] = (int16_t)e; L->wm[h][i] = (int16_t)m; }
        }
    }
}

static void dense_half_cached(const half_layer_t *L, const float *x_in, float *y, int relu, int clamp_in)
{
    int xe[2][64];
    int32_t xm[2][64];
    for (int i = 0; i < L->n_in; ++i) {
        float v = x_in[i];
        if (clamp_in) v = v > 65504.0f ? 65504.0f : (v < -65504.0f ? -65504.0f : v);
        const float hi = round_f16(v), lo = L->split ? round_f16(v - hi) : 0.0f;
        const float p[2] = { hi, lo };
        for (int h = 0; h < 2; ++h) {
            xe[h][i] = 0; xm[h][i] = 0;
            (void)mfma_f16_decompose(p[h], &xe[h][i], &xm[h][i]);
        }
    }
    /* per block, the inputs that are not zero (half of a ReLU layer's activations are): shared by every output row */
    int8_t cnt_c[48];
    int16_t in_c[48][8];
    int16_t xe_c[48][8];
    int32_t xm_c[48][8];
    for (int b = 0; b < L->n_blocks; ++b) {
        const half_block_t *B = &L->blocks[b];
        const int xh = B->term == 1 ? 1 : 0;
        int c = 0;
        for (int q = 0; q < B->cnt; ++q) {
            const int in = B->in[q];
            if (xm[xh][in] == 0) continue;
            in_c[b][c] = (int16_t)in; xe_c[b][c] = (int16_t)xe[xh][in]; xm_c[b][c] = xm[xh][in];
            ++c;
        }
        cnt_c[b] = (int8_t)c;
    }
    for (int o = 0; o < L->n_out; ++o) {
        const int64_t row = (int64_t)o * L->n_in;
        float acc = 0.0f;
        for (int b = 0; b < L->n_blocks; ++b) {
            const int c = cnt_c[b];
            if (c == 0) continue;                                         /* a block of zero products leaves acc as it is */
            const int wh = L->blocks[b].term == 0 ? 1 : 0;                /* 0: w_lo*x_hi  1: w_hi*x_lo  2: w_hi*x_hi */
            const int16_t *we = L->we[wh] + row, *wm = L->wm[wh] + row;
            int e[8];
            int32_t m[8];
            for (int q = 0; q < c; ++q) {
                const int in = in_c[b][q];
                e[q] = we[in] + xe_c[b][q];
                m[q] = (int32_t)wm[in] * xm_c[b][q];
            }
            acc = mfma_f16_block_em(acc, c, e, m);
        }
        y[o] = relu ? (acc > 0.0f ? acc : 0.0f) : acc;
    }
}

/* mode: the field's mlp_half for THIS layer (0, 1 or 2); L: the layer's cache entry or NULL */
static void dense(const float *w, int n_out, int n_in, const float *x_in, float *y, int relu, int mode, int col_map,
                  int clamp_in, const half_layer_t *L)
{
    if (mode && L) {
        dense_half_cached(L, x_in, y, relu, clamp_in);
        return;
    }
    if (mode) {
        dense_half(w, n_out, n_in, x_in, y, relu, mode == 2, col_map, (n_in + 31) / 32, clamp_in);
        return;
    }
    for (int o = 0; o < n_out; ++o) {
        float v = dotf(w + (int64_t)o * n_in, x_in, n_in);
        y[o] = relu ? (v > 0.0f ? v : 0.0f) : v;
    }
}

/* pos: world xyz; t: time; dir: view direction.  Outputs: rgb[3], sigma, geo[15] (any may be NULL). */
static void field_one(const ced_o_field_t *f, const float *pos, float t, const float *dir,
                      float *rgb, float *sigma, float *geo, float *xnorm_out)
{
    /* tcnn Frequency(4) on (x,y,z,t): order [dim][freq][sin,cos]   */
    float enc[32], h0[64], h1[64], mo[6];
    float in4[4] = { pos[0], pos[1], pos[2], t };
    for (int d = 0; d < 4; ++d)
        for (int k = 0; k < 4; ++k) {
            float y = in4[d] * (float)(1 << k);
            enc[d * 8 + k * 2] = ced_o_sinpi_phase(y, 0);
            enc[d * 8 + k * 2 + 1] = ced_o_sinpi_phase(y, 1);
        }
    const int body = f->mlp_half == 3 ? 0 : f->mlp_half, head = f->mlp_half == 3 ? 2 : f->mlp_half;
    const half_layer_t *HL = f->half_cache ? ((const half_cache_t *)f->half_cache)->layer : NULL;
#define LAYER(i) (HL ? &HL[i] : NULL)
    dense(f->m_w0, 64, 32, enc, h0, 1, body, COL_NATURAL, 0, LAYER(0));
    dense(f->m_w1, 64, 64, h0, h1, 1, body, COL_NATURAL, 1, LAYER(1));
    dense(f->m_w2, 64, 64, h1, h0, 1, body, COL_NATURAL, 1, LAYER(2));
    int n_mo = f->use_div_offsets ? 6 : 3;
    dense(f->m_w3, n_mo, 64, h0, mo, 0, body, COL_NATURAL, 1, LAYER(3));
    float move[3], xn[3];
    int sel = 1;
    for (int a = 0; a < 3; ++a) {                                 /* model.py:356-363 */
        float m = mo[a] * f->moving_step;
        if (f->use_div_offsets) {
            float e = ced_o_expf(2.0f * mo[3 + a]);
            float th = 1.0f - 2.0f / (e + 1.0f);                  /* tanh */
            m = m + th * f->moving_step;
        }
        move[a] = m;
        float xm = pos[a] + m;
        xn[a] = (xm - f->aabb[a]) / (f->aabb[3 + a] - f->aabb[a]);   /* model.py:378-379 */
        if (!(xn[a] > 0.0f && xn[a] < 1.0f)) sel = 0;                /* model.py:383 */
    }
    if (xnorm_out) { xnorm_out[0] = xn[0]; xnorm_out[1] = xn[1]; xnorm_out[2] = xn[2]; }
    float bin[41], bout[16];
    hash_encode_one(&f->hash, xn, t, bin);
    if (f->time_mode) {                                           /* model.py:386-403 */
        float mn = sqrtf((move[0] * move[0] + move[1] * move[1]) + move[2] * move[2]);
        ced_o_time_encode(t, mn, f->time_mode == 2, bin + 32);
    }
    dense(f->b_w0, 64, f->base_in, bin, h0, 1, body, COL_HASH, 1, LAYER(4));
    dense(f->b_w1, 16, 64, h0, bout, 0, body, COL_NATURAL, 1, LAYER(5));
    float s = ced_o_expf(bout[0] - 1.0f);                         /* model.py:105,414-417 */
    if (!sel) s = 0.0f;
    if (sigma) *sigma = s;
    if (geo) for (int i = 0; i < 15; ++i) geo[i] = bout[1 + i];
    if (rgb) {
        /* model.py:447-466; SH degree 2 on v = 2*((d/|d|+1)/2) - 1   */
        float nrm = sqrtf((dir[0] * dir[0] + dir[1] * dir[1]) + dir[2] * dir[2]);
        float v[3];
        for (int a = 0; a < 3; ++a) {
            float u = (dir[a] / nrm + 1.0f) / 2.0f;
            v[a] = u * 2.0f - 1.0f;
        }
        float hin[19];
        hin[0] = 0.28209479177387814f;
        hin[1] = -0.48860251190291987f * v[1];
        hin[2] = 0.48860251190291987f * v[2];
        hin[3] = -0.48860251190291987f * v[0];
        for (int i = 0; i < 15; ++i) hin[4 + i] = bout[1 + i];
        float o3[3];
        dense(f->h_w0, 64, 19, hin, h0, 1, head, COL_HEAD, 1, LAYER(6));
        dense(f->h_w1, 64, 64, h0, h1, 1, head, COL_NATURAL, 1, LAYER(7));
        dense(f->h_w2, 3, 64, h1, o3, 0, head, COL_NATURAL, 1, LAYER(8));
        for (int a = 0; a < 3; ++a) rgb[a] = 1.0f / (1.0f + ced_o_expf(-o3[a]));   /* sigmoid */
    }
}

#undef LAYER

/* Decompose the weights of the fp16-operand modes once (f->mlp_half != 0); results are identical without it. */
void ced_o_field_prepare(ced_o_field_t *f)
{
    if (!f->mlp_half || f->half_cache) return;
    half_cache_t *C = (half_cache_t *)calloc(1, sizeof(half_cache_t));
    const int body = f->mlp_half == 3 ? 0 : f->mlp_half, head = f->mlp_half == 3 ? 2 : f->mlp_half;
    const int n_mo = f->use_div_offsets ? 6 : 3;
    if (body) {
        half_layer_build(&C->layer[0], f->m_w0, 64, 32, body == 2, COL_NATURAL);
        half_layer_build(&C->layer[1], f->m_w1, 64, 64, body == 2, COL_NATURAL);
        half_layer_build(&C->layer[2], f->m_w2, 64, 64, body == 2, COL_NATURAL);
        half_layer_build(&C->layer[3], f->m_w3, n_mo, 64, body == 2, COL_NATURAL);
        half_layer_build(&C->layer[4], f->b_w0, 64, f->base_in, body == 2, COL_HASH);
        half_layer_build(&C->layer[5], f->b_w1, 16, 64, body == 2, COL_NATURAL);
    }
    half_layer_build(&C->layer[6], f->h_w0, 64, 19, head == 2, COL_HEAD);
    half_layer_build(&C->layer[7], f->h_w1, 64, 64, head == 2, COL_NATURAL);
    half_layer_build(&C->layer[8], f->h_w2, 3, 64, head == 2, COL_NATURAL);
    f->half_cache = C;
}

void ced_o_field_release(ced_o_field_t *f)
{
    half_cache_t *C = (half_cache_t *)f->half_cache;
    if (!C) return;
    for (int i = 0; i < 9; ++i) {
        free(C->layer[i].blocks);
        for (int h = 0; h < 2; ++h) { free(C->layer[i].we[h]); free(C->layer[i].wm[h]); }
    }
    free(C);
    f->half_cache = NULL;
}

/* explicit positions/dirs/t (DNGPradianceField.forward, model.py:468-488) */
void ced_o_field_forward(const ced_o_field_t *f, int64_t n, const float *pos, const float *t,
                         const float *dir, float *rgb, float *sigma, float *geo, float *xnorm)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i)
        field_one(f, pos + 3 * i, t[i], dir ? dir + 3 * i : NULL, (rgb && dir) ? rgb + 3 * i : NULL,
                  sigma ? sigma + i : NULL, geo ? geo + 15 * i : NULL, xnorm ? xnorm + 3 * i : NULL);
}

/* positions from rays: the rgb_sigma_fn / sigma_fn closures, cednerf/utils.py:74-104,181-195.
 * t_per_ray: 0 -> timestamps[0] for every sample (eval), 1 -> timestamps[ray] (training). */
void ced_o_field_forward_rays(const ced_o_field_t *f, int64_t n, const float *rays_o, const float *rays_d,
                              const int64_t *ray_indices, const float *t_starts, const float *t_ends,
                              const float *timestamps, int t_per_ray, int want_rgb,
                              float *rgb, float *sigma, float *geo)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = ray_indices[i];
        float tm2 = t_starts[i] + t_ends[i];
        float pos[3], dir[3];
        for (int a = 0; a < 3; ++a) {
            dir[a] = rays_d[3 * r + a];
            pos[a] = rays_o[3 * r + a] + (dir[a] * tm2) / 2.0f;   /* utils.py:76-77,184 */
        }
        float t = t_per_ray ? timestamps[r] : timestamps[0];
        field_one(f, pos, t, dir, want_rgb ? rgb + 3 * i : NULL, sigma + i, geo ? geo + 15 * i : NULL, NULL);
    }
}

/* ------------------------------------------------------------------------------------------
 * a15/a4: render_weight_from_density / render_transmittance_from_density (call sites
 * cednerf/render.py:52-54,81-87, cednerf/utils.py:274-281), packed per ray.
 * ---------------------------------------------------------------------------------------- */
void ced_o_render_weights(int64_t n_rays, const int64_t *packed_info /* [n_rays,2] */,
                          const float *t_starts, const float *t_ends, const float *sigmas,
                          const float *prefix_trans /* per SAMPLE, may be NULL */,
                          float *weights, float *trans, float *alphas)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t s0 = packed_info[2 * r], cnt = packed_info[2 * r + 1];
        float acc = 0.0f;
        for (int64_t i = s0; i < s0 + cnt; ++i) {
            float sd = sigmas[i] * (t_ends[i] - t_starts[i]);
            float a = 1.0f - ced_o_expf(-sd);
            float tr = ced_o_expf(-acc);
            if (prefix_trans) tr = tr * prefix_trans[i];
            if (alphas) alphas[i] = a;
            if (trans) trans[i] = tr;
            if (weights) weights[i] = tr * a;
            acc = acc + sd;
        }
    }
}

/* accumulate_along_rays(_): out[ray] += w * v, sequential in sample order (SURVEY A.5;
 * cednerf/render.py:158-169, cednerf/utils.py:282-299). values may be NULL (C = 1). */
void ced_o_accumulate(int64_t n_rays, const int64_t *packed_info, const float *weights,
                      const float *values, int C, float *out /* [n_rays, C], in place */)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t s0 = packed_info[2 * r], cnt = packed_info[2 * r + 1];
        for (int c = 0; c < C; ++c) {
            float acc = out[r * C + c];
            for (int64_t i = s0; i < s0 + cnt; ++i)
                acc = acc + (values ? weights[i] * values[i * C + c] : weights[i]);
            out[r * C + c] = acc;
        }
    }
}

/* Backward of the training-time compositing (next row f2): the chain render_weight_from_density ->
 * accumulate_along_rays x3 of cednerf/render.py:158-169 (colors, opacities, depths), differentiated w.r.t. the
 * per-sample sigmas and rgbs.  Evaluated in DOUBLE (the derivative of the real-valued formula):
 *   w_i = T_i a_i,  T_i = exp(-sum_{j<i} sd_j),  a_i = 1 - exp(-sd_i),  sd_i = sigma_i (t1_i - t0_i)
 *   g_i = dL/dw_i = <d_color, rgb_i> + d_opacity + d_depth * (t0_i + t1_i)/2
 *   dL/drgb_i = w_i d_color;   dL/dsigma_i = (t1_i - t0_i) [ g_i (T_i - w_i) - sum_{k>i} g_k w_k ]. */
void ced_o_composite_backward(int64_t n_rays, const int64_t *packed_info, const float *t_starts, const float *t_ends,
                              const float *sigmas, const float *rgbs, const float *d_color /* [n_rays,3] */,
                              const float *d_opacity /* [n_rays] */, const float *d_depth /* [n_rays] */,
                              double *d_sigmas, double *d_rgbs /* [S,3] */)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t s0 = packed_info[2 * r], cnt = packed_info[2 * r + 1];
        double total = 0.0;
        for (int64_t i = s0; i < s0 + cnt; ++i) total += (double)sigmas[i] * ((double)t_ends[i] - (double)t_starts[i]);
        double acc_after = total, suffix = 0.0;                      /* sum_{j<=i} sd_j, sum_{k>i} g_k w_k */
        for (int64_t i = s0 + cnt - 1; i >= s0; --i) {
            double dt = (double)t_ends[i] - (double)t_starts[i];
            double sd = (double)sigmas[i] * dt;
            double acc_before = acc_after - sd;
            double T = exp(-acc_before), a = 1.0 - exp(-sd), w = T * a;
            double g = (double)d_color[3 * r] * rgbs[3 * i] + (double)d_color[3 * r + 1] * rgbs[3 * i + 1] +
                       (double)d_color[3 * r + 2] * rgbs[3 * i + 2] + (double)d_opacity[r] +
                       (double)d_depth[r] * (((double)t_starts[i] + (double)t_ends[i]) / 2.0);
            d_sigmas[i] = dt * (g * (T - w) - suffix);
            for (int c = 0; c < 3; ++c) d_rgbs[3 * i + c] = w * (double)d_color[3 * r + c];
            suffix += g * w;
            acc_after = acc_before;
        }
    }
}

/* Weight gradient of a bias-free dense layer, dw[o][i] = sum_s dy[s][o] x[s][i], in float64 (checker of
 * ced_weight_grad; the GEMM tiny-cuda-nn runs for the modules of cednerf/model.py:200-222,280-309 in backward). */
void ced_o_weight_grad(int64_t n, const float *x, int32_t n_in, const float *dy, int32_t n_out, double *dw)
{
    for (int64_t e = 0; e < (int64_t)n_out * n_in; ++e) dw[e] = 0.0;
    for (int64_t s = 0; s < n; ++s)
        for (int o = 0; o < n_out; ++o) {
            double g = (double)dy[s * n_out + o];
            for (int i = 0; i < n_in; ++i) dw[o * n_in + i] += g * (double)x[s * n_in + i];
        }
}

/* render_visibility_from_density (inside OccGridEstimator.sampling, utils.py:115-125). */
void ced_o_visibility(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                      const float *t_ends, const float *sigmas, float early_stop_eps, float alpha_thre,
                      uint8_t *mask)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rays; ++r) {
        int64_t s0 = packed_info[2 * r], cnt = packed_info[2 * r + 1];
        float acc = 0.0f;
        for (int64_t i = s0; i < s0 + cnt; ++i) {
            float sd = sigmas[i] * (t_ends[i] - t_starts[i]);
            float a = 1.0f - ced_o_expf(-sd);
            float tr = ced_o_expf(-acc);
            int vis = tr >= early_stop_eps;
            if (alpha_thre > 0.0f) vis = vis && (a >= alpha_thre);
            mask[i] = (uint8_t)vis;
            acc = acc + sd;
        }
    }
}

/* a16: composite_test, cednerf/taichi_kernel/volume_render_test.py:4-59 (never called by the
 * reference; kept as the spec of the fused per-ray compositor). alive[n] set to -1 when done. */
void ced_o_composite_test(int64_t n_alive, const float *sigmas, const float *rgbs, const float *t_start,
                          const float *t_end, const int64_t *pack_info, int64_t *alive_indices,
                          float T_threshold, float alpha_threshold, float *opacity, float *depth, float *rgb)
{
    for (int64_t n = 0; n < n_alive; ++n) {
        int64_t start = pack_info[2 * n], steps = pack_info[2 * n + 1];
        int64_t ray = alive_indices[n];
        if (steps == 0) { alive_indices[n] = -1; continue; }
        float T = 1.0f - opacity[ray];
        float c[3] = { 0, 0, 0 }, dacc = 0.0f, oacc = 0.0f;
        for (int64_t s = 0; s < steps; ++s) {
            int64_t i = start + s;
            float delta = t_end[i] - t_start[i];
            float a = 1.0f - ced_o_expf(-sigmas[i] * delta);
            if (a > alpha_threshold) {
                float w = a * T;
                float tmid = (t_start[i] + t_end[i]) / 2.0f;
                for (int k = 0; k < 3; ++k) c[k] = c[k] + w * rgbs[3 * i + k];
                dacc = dacc + w * tmid;
                oacc = oacc + w;
                T = T * (1.0f - a);
                if (T <= T_threshold) { alive_indices[n] = -1; break; }
            }
        }
        for (int k = 0; k < 3; ++k) rgb[3 * ray + k] += c[k];
        depth[ray] += dacc;
        opacity[ray] += oacc;
    }
}

int ced_o_version(void) { return 1; }
