// Occupancy-grid ray marching: ray/AABB slab test and multi-level DDA traversal.
// Replaces nerfacc.ray_aabb_intersect / nerfacc.traverse_grids (un-vendored CUDA ops) at the call
// sites cednerf/utils.py:215 and cednerf/utils.py:241-264 (and, through OccGridEstimator.sampling,
// cednerf/utils.py:115-125) of the reference.  One lane per ray; integer outputs are bit-exact
// with the oracle because every float operation below is a single IEEE op in a fixed order
// (-ffp-contract=off) -- do not "simplify" the arithmetic.
#include "ced_common.hpp"
#include "march_core.hpp"

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

struct TraverseArgs {
    int64_t n_rays;
    const float *rays_o, *rays_d;
    GridSpec grid;
    const float *near_planes, *far_planes;
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
    else if (A.mode == 2) out_base = r * (int64_t)A.grid.limit;
    else if (A.mode == 3) out_base = A.base[r] - A.counts[r];      // base = inclusive scan of the counts
    if (A.rays_mask && !A.rays_mask[r]) {
        A.counts[r] = 0;
        if (A.termination_planes) A.termination_planes[r] = near;
        if (A.packed_info_out) { A.packed_info_out[2 * r] = out_base; A.packed_info_out[2 * r + 1] = 0; }
        return;
    }
    const float o[3] = { A.rays_o[3 * r], A.rays_o[3 * r + 1], A.rays_o[3 * r + 2] };
    const float d[3] = { A.rays_d[3 * r], A.rays_d[3 * r + 1], A.rays_d[3 * r + 2] };
    const int m = A.grid.n_grids;
    const bool fill = A.mode != 0;
    float t_term;
    const int n = traverse_ray(
        A.grid, o, d, near, far, A.t_sorted + r * 2 * m, A.t_indices + r * 2 * m, A.hits + r * m,
        [&](int i, float t0, float t1) {
            if (fill) {
                A.t_starts[out_base + i] = t0;
                A.t_ends[out_base + i] = t1;
                if (A.ray_indices) A.ray_indices[out_base + i] = r;
            }
        },
        t_term);
    A.counts[r] = n;
    if (A.termination_planes) A.termination_planes[r] = t_term;
    if (A.packed_info_out) { A.packed_info_out[2 * r] = out_base; A.packed_info_out[2 * r + 1] = n; }
}

// The event list of cednerf/utils.py:219-225: torch.sort(cat([t_mins, t_maxs], -1), stable=True) per ray -- 2 m <= 16 keys
// per ray, a stable insertion sort in registers of one lane (torch's segmented radix sort takes 3.1 ms for the 1.37 M
// rays of a 1352x1014 frame: 39 % of render_image's kernel time there).  NaN keys sort last, as torch's do.
constexpr int kSortMaxGrids = 8;
__global__ __launch_bounds__(256) void sort_intersections_kernel(int64_t n_rays, int m, const float *__restrict__ t_mins,
                                                                 const float *__restrict__ t_maxs,
                                                                 float *__restrict__ t_sorted, int64_t *__restrict__ t_indices)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float ev[2 * kSortMaxGrids];
    int id[2 * kSortMaxGrids];
#pragma unroll
    for (int a = 0; a < kSortMaxGrids; ++a) {
        if (a < m) {
            ev[a] = t_mins[r * m + a]; id[a] = a;
        }
    }
#pragma unroll
    for (int a = 0; a < kSortMaxGrids; ++a) {
        if (a < m) {
            // position m + a: filled through a select chain (the arrays stay in registers)
#pragma unroll
            for (int q = 0; q < 2 * kSortMaxGrids; ++q)
                if (q == m + a) { ev[q] = t_maxs[r * m + a]; id[q] = m + a; }
        }
    }
    // stable insertion sort as a fixed network of compare-and-shift steps over the first 2 m entries
#pragma unroll
    for (int i = 1; i < 2 * kSortMaxGrids; ++i) {
        if (i < 2 * m) {
#pragma unroll
            for (int j = i; j >= 1; --j) {
                const float a0 = ev[j - 1], a1 = ev[j];
                // strictly greater moves behind (equal keys keep their order); NaN counts as greater than any number
                const bool swap = (a0 > a1) || (a0 != a0 && a1 == a1);
                const int i0 = id[j - 1], i1 = id[j];
                ev[j - 1] = swap ? a1 : a0; ev[j] = swap ? a0 : a1;
                id[j - 1] = swap ? i1 : i0; id[j] = swap ? i0 : i1;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2 * kSortMaxGrids; ++q) {
        if (q < 2 * m) {
            t_sorted[r * 2 * m + q] = ev[q];
            t_indices[r * 2 * m + q] = id[q];
        }
    }
}

}  // namespace ced

extern "C" int ced_sort_intersections(int64_t n_rays, int32_t n_grids, const float *t_mins, const float *t_maxs,
                                      float *t_sorted, int64_t *t_indices, void *stream)
{
    CED_REQUIRE(n_rays >= 0 && n_grids >= 1 && n_grids <= ced::kSortMaxGrids, "sort_intersections: n_grids must be 1..%d",
                ced::kSortMaxGrids);
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(t_mins && t_maxs && t_sorted && t_indices, "sort_intersections: null pointer");
    hipLaunchKernelGGL(ced::sort_intersections_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_rays, (int)n_grids, t_mins, t_maxs, t_sorted, t_indices);
    return ced::check_launch("sort_intersections");
}

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
    ced::TraverseArgs A{ n_rays, rays_o, rays_d, ced::GridSpec{ binaries, aabbs, n_grids, res, step_size, cone_angle, limit },
                         near_planes, far_planes, rays_mask, t_sorted, t_indices, hits, mode, base, counts, t_starts,
                         t_ends, ray_indices, termination_planes, packed_info_out };
    dim3 block(256), grid((unsigned)((n_rays + 255) / 256));
    hipLaunchKernelGGL(ced::traverse_kernel, grid, block, 0, (hipStream_t)stream, A);
    return ced::check_launch("traverse_grids");
}

// HOST evaluation of the kernels' empty-space skip (closed form for cone_angle == 0, the sequential
// recurrence otherwise); lets the CPU test-suite check the closed form against the oracle's loop.
extern "C" float ced_host_skip_march(float t_last, float target, float step_size, float cone_angle)
{
    return ced::skip_march(t_last, target, step_size, cone_angle);
}
