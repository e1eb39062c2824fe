// Occupancy-grid maintenance: the device side of
// nerfacc.OccGridEstimator._update / update_every_n_steps as driven at train_real.py:324-336.
// nerfacc itself does these steps with torch ops; here the two per-sample pieces around the density
// query are single launches: cell index + jitter -> world position, and the EMA-max write-back.
#include "ced_common.hpp"

namespace ced {

// x = (grid_coord(idx) + noise) / res  ->  aabb_min + x * (aabb_max - aabb_min); idx = (ix*res + iy)*res + iz
__global__ __launch_bounds__(256) void occ_points_kernel(int64_t n, const int64_t *__restrict__ cell_idx,
                                                         const float *__restrict__ noise, int res, float a0, float a1,
                                                         float a2, float e0, float e1, float e2,
                                                         float *__restrict__ pos)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t c = cell_idx[i];
    const int iz = (int)(c % res), iy = (int)((c / res) % res), ix = (int)(c / ((int64_t)res * res));
    const float resf = (float)res;
    const float x = ((float)ix + noise[3 * i]) / resf;
    const float y = ((float)iy + noise[3 * i + 1]) / resf;
    const float z = ((float)iz + noise[3 * i + 2]) / resf;
    pos[3 * i] = a0 + x * e0;
    pos[3 * i + 1] = a1 + y * e1;
    pos[3 * i + 2] = a2 + z * e2;
}

// occs[id] = max(occs[id] * decay, density * step)   (one writer per sample, like the index_put it replaces)
__global__ __launch_bounds__(256) void occ_ema_kernel(int64_t n, const int64_t *__restrict__ cell_ids,
                                                      const float *__restrict__ density, float step_size, float decay,
                                                      float *__restrict__ occs)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t id = cell_ids[i];
    const float occ = density[i] * step_size;
    occs[id] = __builtin_fmaxf(occs[id] * decay, occ);
}

}  // namespace ced

extern "C" int ced_occ_cell_points(int64_t n, const int64_t *cell_indices, const float *noise, int32_t res,
                                   const float *aabb_host, float *positions, void *stream)
{
    CED_REQUIRE(n >= 0 && res >= 1, "occ_cell_points: bad sizes");
    if (n == 0) return CED_OK;
    CED_REQUIRE(cell_indices && noise && aabb_host && positions, "occ_cell_points: null pointer");
    hipLaunchKernelGGL(ced::occ_points_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,
                       cell_indices, noise, (int)res, aabb_host[0], aabb_host[1], aabb_host[2],
                       aabb_host[3] - aabb_host[0], aabb_host[4] - aabb_host[1], aabb_host[5] - aabb_host[2], positions);
    return ced::check_launch("occ_cell_points");
}

extern "C" int ced_occ_ema_update(int64_t n, const int64_t *cell_ids, const float *density, float step_size,
                                  float ema_decay, float *occs, void *stream)
{
    CED_REQUIRE(n >= 0, "occ_ema_update: n < 0");
    if (n == 0) return CED_OK;
    CED_REQUIRE(cell_ids && density && occs, "occ_ema_update: null pointer");
    hipLaunchKernelGGL(ced::occ_ema_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, n,
                       cell_ids, density, step_size, ema_decay, occs);
    return ced::check_launch("occ_ema_update");
}
