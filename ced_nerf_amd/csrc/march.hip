// Occupancy-grid ray marching: ray/AABB slab test and multi-level DDA traversal.
// Replaces nerfacc.ray_aabb_intersect / nerfacc.traverse_grids (un-vendored CUDA ops) at the call
// sites cednerf/utils.py:215 and cednerf/utils.py:241-264 (and, through OccGridEstimator.sampling,
// cednerf/utils.py:115-125) of the reference.  One lane per ray; integer outputs are bit-exact
// with the oracle because every float operation below is a single IEEE op in a fixed order
// (-ffp-contract=off) -- do not "simplify" the arithmetic.
#include "ced_common.hpp"

namespace ced {

__device__ __forceinline__ bool ray_aabb_one(const float o[3], const float inv_d[3], const float *__restrict__ aabb,
                                             float near, float far, float &tmin_out, float &tmax_out)
{
    float tmin, tmax, tymin, tymax, tzmin, tzmax;
    if (inv_d[0] >= 0) { tmin = (aabb[0] - o[0]) * inv_d[0]; tmax = (aabb[3] - o[0]) * inv_d[0]; }
    else               { tmin = (aabb[3] - o[0]) * inv_d[0]; tmax = (aabb[0] - o[0]) * inv_d[0]; }
    if (inv_d[1] >= 0) { tymin = (aabb[1] - o[1]) * inv_d[1]; tymax = (aabb[4] - o[1]) * inv_d[1]; }
    else               { tymin = (aabb[4] - o[1]) * inv_d[1]; tymax = (aabb[1] - o[1]) * inv_d[1]; }
    if (tmin > tymax || tymin > tmax) return false;
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    if (inv_d[2] >= 0) { tzmin = (aabb[2] - o[2]) * inv_d[2]; tzmax = (aabb[5] - o[2]) * inv_d[2]; }
    else               { tzmin = (aabb[5] - o[2]) * inv_d[2]; tzmax = (aabb[2] - o[2]) * inv_d[2]; }
    if (tmin > tzmax || tzmin > tmax) return false;
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    if (tmax <= 0) return false;
    tmin_out = __builtin_fmaxf(tmin, near);
    tmax_out = __builtin_fminf(tmax, far);
    return true;
}

__global__ __launch_bounds__(256) void ray_aabb_kernel(int64_t n_rays, const float *__restrict__ rays_o,
                                                       const float *__restrict__ rays_d, int n_aabbs,
                                                       const float *__restrict__ aabbs, float near, float far,
                                                       float miss, float *__restrict__ t_mins,
                                                       float *__restrict__ t_maxs, uint8_t *__restrict__ hits)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float o[3] = { rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2] };
    float inv_d[3] = { 1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2] };
    for (int a = 0; a < n_aabbs; ++a) {
        float t0 = miss, t1 = miss;
        bool hit = ray_aabb_one(o, inv_d, aabbs + 6 * a, near, far, t0, t1);
        if (!hit) { t0 = miss; t1 = miss; }
        t_mins[r * n_aabbs + a] = t0;
        t_maxs[r * n_aabbs + a] = t1;
        hits[r * n_aabbs + a] = hit ? 1 : 0;
    }
}

__device__ __forceinline__ float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    float v = t * cone_angle;
    return __builtin_fminf(__builtin_fmaxf(v, dt_min), dt_max);
}
// march t_last forward in whole steps until the next step's mid-point reaches `target`
__device__ __forceinline__ float skip_march(float t_last, float target, float step_size, float cone_angle)
{
    if (step_size <= 0.0f) return target;
    for (;;) {
        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
        if (t_last + dt * 0.5f >= target) break;
        t_last += dt;
    }
    return t_last;
}
constexpr int kLook = 4;
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct TraverseArgs {
    int64_t n_rays;
    const float *rays_o, *rays_d;
    const uint8_t *binaries;
    int n_grids, res;
    const float *aabbs;
    const float *near_planes, *far_planes;
    float step_size, cone_angle;
    int limit;
    const uint8_t *rays_mask;
    const float *t_sorted;
    const int64_t *t_indices;
    const uint8_t *hits;
    int mode;
    const int64_t *base;
    int64_t *counts;
    float *t_starts, *t_ends;
    int64_t *ray_indices;
    float *termination_planes;
    int64_t *packed_info_out;
};

__global__ __launch_bounds__(256) void traverse_kernel(TraverseArgs A)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= A.n_rays) return;
    const float near = A.near_planes[r], far = A.far_planes[r];
    int64_t out_base = 0;
    if (A.mode == 1) out_base = A.base[r];
    else if (A.mode == 2) out_base = r * (int64_t)A.limit;
    else if (A.mode == 3) out_base = A.base[r] - A.counts[r];      // base = inclusive scan of the counts
    if (A.rays_mask && !A.rays_mask[r]) {
        A.counts[r] = 0;
        if (A.termination_planes) A.termination_planes[r] = near;
        if (A.packed_info_out) { A.packed_info_out[2 * r] = out_base; A.packed_info_out[2 * r + 1] = 0; }
        return;
    }
    const float eps = 1e-6f;
    const float o[3] = { A.rays_o[3 * r], A.rays_o[3 * r + 1], A.rays_o[3 * r + 2] };
    const float d[3] = { A.rays_d[3 * r], A.rays_d[3 * r + 1], A.rays_d[3 * r + 2] };
    const float inv_d[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    const int n_grids = A.n_grids, res = A.res, limit = A.limit;
    const float step_size = A.step_size, cone_angle = A.cone_angle;
    const float resf = (float)res;
    const float *ts_row = A.t_sorted + r * 2 * n_grids;
    const int64_t *ti_row = A.t_indices + r * 2 * n_grids;
    const uint8_t *hit_row = A.hits + r * n_grids;
    const bool fill = A.mode != 0;

    float t_last = near;
    bool continuous = false;
    int64_t n = 0;
    for (int i = 0; i < 2 * n_grids - 1; ++i) {
        int64_t ti = ti_row[i];
        bool entering = ti < n_grids;
        int lvl = (int)(ti % n_grids);
        if (!hit_row[lvl]) continue;
        if (!entering) {
            int64_t tn = ti_row[i + 1];
            if (tn < n_grids) continue;
            lvl = (int)(tn % n_grids);
            if (!hit_row[lvl]) continue;
        }
        float this_tmin = __builtin_fmaxf(ts_row[i], near);
        float this_tmax = __builtin_fminf(ts_row[i + 1], far);
        if (this_tmin >= this_tmax) continue;
        if (!continuous) t_last = skip_march(t_last, this_tmin, step_size, cone_angle);
        const float *ab = A.aabbs + 6 * lvl;
        float tdist[3], delta[3];
        int cur[3], stp[3], ovf[3];
        const float ts = this_tmin + eps, te = this_tmax - eps;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float ext = ab[3 + a] - ab[a];
            float vox = ext / resf;
            float ps = o[a] + d[a] * ts;
            float pe = o[a] + d[a] * te;
            cur[a] = clampi((int)(((ps - ab[a]) / ext) * resf), 0, res - 1);
            int fin = clampi((int)(((pe - ab[a]) / ext) * resf), 0, res - 1);
            int idelta = d[a] > 0.0f ? 1 : 0;
            float tm = ((ab[a] + (((float)(cur[a] + idelta) * vox) - ps)) * inv_d[a]) + this_tmin;
            float stepf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
            stp[a] = (int)stepf;
            tdist[a] = (d[a] == 0.0f) ? this_tmax : tm;
            delta[a] = (d[a] == 0.0f) ? this_tmax : (vox * inv_d[a]) * stepf;
            ovf[a] = fin + stp[a];
        }
        const uint8_t *grid = A.binaries + (int64_t)lvl * res * res * res;
        // The DDA path does not depend on the occupancy values, so it runs kLook cells ahead and the
        // occupancy bytes of those cells are fetched together (one dependent-load latency per kLook
        // cells instead of per cell).  Runs of empty cells only remember the farthest boundary; the
        // skip-march to it happens once, before the next occupied cell or at the end -- the same
        // t_last sequence as marching cell by cell, because the recurrence t_last += dt does not
        // depend on where the intermediate boundaries are.
        bool dda_done = false, stop = false, has_pending = false;
        float pending = 0.0f;
        while (!dda_done && !stop) {
            float tt[kLook];
            int64_t cellv[kLook];
            bool valid[kLook];
#pragma unroll
            for (int b = 0; b < kLook; ++b) {
                valid[b] = !dda_done;
                tt[b] = __builtin_fminf(__builtin_fminf(tdist[0], __builtin_fminf(tdist[1], tdist[2])), this_tmax);
                cellv[b] = ((int64_t)cur[0] * res + cur[1]) * res + cur[2];
                if (!dda_done) {
                    int ax;
                    if (tdist[0] < tdist[1] && tdist[0] < tdist[2]) ax = 0;
                    else if (tdist[1] < tdist[2]) ax = 1;
                    else ax = 2;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        if (a == ax) {
                            cur[a] += stp[a];
                            tdist[a] += delta[a];
                            dda_done = (cur[a] == ovf[a]);
                        }
                    }
                }
            }
            uint8_t occ[kLook];
#pragma unroll
            for (int b = 0; b < kLook; ++b) occ[b] = valid[b] ? grid[cellv[b]] : (uint8_t)0;
#pragma unroll
            for (int b = 0; b < kLook; ++b) {
                if (!valid[b] || stop) continue;
                if (limit > 0 && n >= limit) { stop = true; continue; }
                const float t_trav = tt[b];
                if (!occ[b]) {
                    pending = t_trav;
                    has_pending = true;
                    continuous = false;
                    continue;
                }
                if (has_pending) {
                    t_last = skip_march(t_last, pending, step_size, cone_angle);
                    has_pending = false;
                }
                while (limit <= 0 || n < limit) {
                    float t_next;
                    if (step_size <= 0.0f) {
                        t_next = t_trav;
                    } else {
                        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_trav) break;
                        t_next = t_last + dt;
                    }
                    if (fill) {
                        A.t_starts[out_base + n] = t_last;
                        A.t_ends[out_base + n] = t_next;
                        if (A.ray_indices) A.ray_indices[out_base + n] = r;
                    }
                    n += 1;
                    continuous = true;
                    t_last = t_next;
                    if (t_next >= t_trav) break;
                }
            }
        }
        if (has_pending) t_last = skip_march(t_last, pending, step_size, cone_angle);
    }
    A.counts[r] = n;
    if (A.termination_planes) A.termination_planes[r] = t_last;
    if (A.packed_info_out) { A.packed_info_out[2 * r] = out_base; A.packed_info_out[2 * r + 1] = n; }
}

}  // namespace ced

extern "C" int ced_ray_aabb_intersect(int64_t n_rays, const float *rays_o, const float *rays_d, int32_t n_aabbs,
                                      const float *aabbs, float near_plane, float far_plane, float miss_value,
                                      float *t_mins, float *t_maxs, uint8_t *hits, void *stream)
{
    CED_REQUIRE(n_rays >= 0 && n_aabbs >= 1, "ray_aabb_intersect: bad sizes n_rays=%lld n_aabbs=%d",
                (long long)n_rays, n_aabbs);
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(rays_o && rays_d && aabbs && t_mins && t_maxs && hits, "ray_aabb_intersect: null pointer");
    dim3 block(256), grid((unsigned)((n_rays + 255) / 256));
    hipLaunchKernelGGL(ced::ray_aabb_kernel, grid, block, 0, (hipStream_t)stream, n_rays, rays_o, rays_d,
                       (int)n_aabbs, aabbs, near_plane, far_plane, miss_value, t_mins, t_maxs, hits);
    return ced::check_launch("ray_aabb_intersect");
}

extern "C" int ced_traverse_grids(int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *binaries,
                                  int32_t n_grids, int32_t res, const float *aabbs, const float *near_planes,
                                  const float *far_planes, float step_size, float cone_angle, int32_t limit,
                                  const uint8_t *rays_mask, const float *t_sorted, const int64_t *t_indices,
                                  const uint8_t *hits, int32_t mode, const int64_t *base, int64_t *counts,
                                  float *t_starts, float *t_ends, int64_t *ray_indices, float *termination_planes,
                                  int64_t *packed_info_out, void *stream)
{
    CED_REQUIRE(n_rays >= 0 && n_grids >= 1 && res >= 1, "traverse_grids: bad sizes");
    CED_REQUIRE(mode >= 0 && mode <= 3,
                "traverse_grids: mode must be 0 (count), 1 (fill), 2 (over-allocate) or 3 (fill, inclusive scan)");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(rays_o && rays_d && binaries && aabbs && near_planes && far_planes && t_sorted && t_indices && hits &&
                    counts,
                "traverse_grids: null pointer");
    if (mode == 1 || mode == 3)
        CED_REQUIRE(base && t_starts && t_ends, "traverse_grids: fill mode needs base/t_starts/t_ends");
    if (mode == 2) CED_REQUIRE(limit > 0 && t_starts && t_ends, "traverse_grids: over-allocate needs limit > 0");
    ced::TraverseArgs A{ n_rays, rays_o, rays_d, binaries, n_grids, res, aabbs, near_planes, far_planes, step_size,
                         cone_angle, limit, rays_mask, t_sorted, t_indices, hits, mode, base, counts, t_starts,
                         t_ends, ray_indices, termination_planes, packed_info_out };
    dim3 block(256), grid((unsigned)((n_rays + 255) / 256));
    hipLaunchKernelGGL(ced::traverse_kernel, grid, block, 0, (hipStream_t)stream, A);
    return ced::check_launch("traverse_grids");
}
