// Occupancy acceleration structure of the frame renderer: a per-brick Chebyshev distance field over the occupancy
// grid (march_accel.hpp explains how the marching uses it).  Built on the device from `binaries` -- once per
// occupancy-grid update when the caller keeps it (ced_build_occupancy_accel), or inside every render call otherwise.
// Also the HOST twins used by the CPU test-suite: the same traversal code (march_accel.hpp is host + device) run
// against the CPU oracle without a GPU.
#include <algorithm>
#include <vector>

#include "ced_common.hpp"
#include "march_accel.hpp"

namespace ced {

constexpr int kDistCap = 15;       // distances are exact up to here; farther bricks read kDistCap + 1 (a lower bound)

// brick_any[b] = any occupied cell in brick b.  One 64-lane wave per brick: lane (x, y) ORs the 8 contiguous z
// bytes of its row, then a wave-wide ballot.
__global__ __launch_bounds__(64) void brick_any_kernel(const uint8_t *__restrict__ binaries, int res, int nb,
                                                       uint8_t *__restrict__ any)
{
    const int idx = blockIdx.x;
    const int bz = idx % nb, by = (idx / nb) % nb, bx = (idx / (nb * nb)) % nb, lvl = idx / (nb * nb * nb);
    const uint8_t *g = binaries + (size_t)lvl * res * res * res;
    const int x = bx * kBrick + (threadIdx.x >> 3), y = by * kBrick + (threadIdx.x & 7);
    uint8_t acc = 0;
    if (x < res && y < res)
        for (int z = bz * kBrick; z < min((bz + 1) * kBrick, res); ++z) acc |= g[((size_t)x * res + y) * res + z];
    const unsigned long long any_lane = __ballot(acc != 0);
    if (threadIdx.x == 0) any[idx] = any_lane ? 0 : 255;            // distance 0, or "not reached yet"
}

// One workgroup per grid level, the level's brick array in LDS (nb <= 32): kDistCap rounds of 3x3x3 dilation; a brick
// reached in round r has distance r.
__global__ __launch_bounds__(1024) void brick_dist_lds_kernel(uint8_t *__restrict__ dist, int nb)
{
    extern __shared__ uint8_t sm[];
    const int n = nb * nb * nb;
    uint8_t *cur = sm, *nxt = sm + n;
    uint8_t *level = dist + (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) cur[i] = level[i];
    __syncthreads();
    for (int r = 1; r <= kDistCap; ++r) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            uint8_t v = cur[i];
            if (v == 255) {
                const int bz = i % nb, by = (i / nb) % nb, bx = i / (nb * nb);
                bool reach = false;
                for (int x = max(bx - 1, 0); x <= min(bx + 1, nb - 1); ++x)
                    for (int y = max(by - 1, 0); y <= min(by + 1, nb - 1); ++y)
                        for (int z = max(bz - 1, 0); z <= min(bz + 1, nb - 1); ++z) reach |= cur[(x * nb + y) * nb + z] != 255;
                if (reach) v = (uint8_t)r;
            }
            nxt[i] = v;
        }
        __syncthreads();
        uint8_t *t = cur; cur = nxt; nxt = t;
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) level[i] = cur[i] == 255 ? (uint8_t)(kDistCap + 1) : cur[i];
}

// Large grids (nb > 32): the same rounds through global memory, one launch per round.
__global__ __launch_bounds__(256) void brick_dist_round_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                               int n_levels, int nb, int r, int final_round)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)nb * nb * nb;
    if (idx >= n * n_levels) return;
    const int i = (int)(idx % n);
    const uint8_t *lv = src + (idx / n) * n;
    uint8_t v = lv[i];
    if (v == 255) {
        const int bz = i % nb, by = (i / nb) % nb, bx = i / (nb * nb);
        bool reach = false;
        for (int x = max(bx - 1, 0); x <= min(bx + 1, nb - 1); ++x)
            for (int y = max(by - 1, 0); y <= min(by + 1, nb - 1); ++y)
                for (int z = max(bz - 1, 0); z <= min(bz + 1, nb - 1); ++z) reach |= lv[(x * nb + y) * nb + z] != 255;
        if (reach) v = (uint8_t)r;
    }
    if (final_round && v == 255) v = (uint8_t)(kDistCap + 1);
    dst[idx] = v;
}

constexpr int kCellCap = 15;       // cell distances are exact up to here; farther cells carry the brick bound

// ---- cell-level field: D(x) = min over occupied p of max_a |x_a - p_a|, separable as three min-max passes ----
// pass z: 1-D distance along z inside every (x, y) column (one thread per column, two sweeps)
__global__ __launch_bounds__(256) void cell_dist_z_kernel(const uint8_t *__restrict__ binaries, int n_levels, int res,
                                                          uint8_t *__restrict__ out)
{
    const int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= (int64_t)n_levels * res * res) return;
    const uint8_t *g = binaries + col * res;
    uint8_t *o = out + col * res;
    int dmin = kCellCap + 1;
    for (int z = 0; z < res; ++z) {
        dmin = g[z] ? 0 : min(dmin + 1, kCellCap + 1);
        o[z] = (uint8_t)dmin;
    }
    dmin = kCellCap + 1;
    for (int z = res - 1; z >= 0; --z) {
        dmin = g[z] ? 0 : min(dmin + 1, kCellCap + 1);
        o[z] = (uint8_t)min((int)o[z], dmin);
    }
}

// passes y and x: g'(c) = min over |k| <= cap of max(|k|, g(c + k * stride)) along one axis
__global__ __launch_bounds__(256) void cell_dist_axis_kernel(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst,
                                                             int n_levels, int res, int axis)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)res * res * res;
    if (idx >= n * n_levels) return;
    const int i = (int)(idx % n);
    const int z = i % res, y = (i / res) % res, x = i / (res * res);
    const int pos = axis == 0 ? x : y;
    const int64_t stride = axis == 0 ? (int64_t)res * res : res;
    int best = src[idx];
    for (int k = 1; k < best && k <= kCellCap; ++k) {        // a neighbour k away can only help while k < best
        if (pos - k >= 0) best = min(best, max(k, (int)src[idx - k * stride]));
        if (pos + k < res) best = min(best, max(k, (int)src[idx + k * stride]));
    }
    (void)z;
    dst[idx] = (uint8_t)best;
}

// beyond the exact range a cell carries the bound of its brick: (R - 1) * 8 + 1 cells
__global__ __launch_bounds__(256) void cell_dist_combine_kernel(uint8_t *__restrict__ cdist, const uint8_t *__restrict__ bdist,
                                                                int n_levels, int res, int nb)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)res * res * res;
    if (idx >= n * n_levels) return;
    int v = cdist[idx];
    if (v > kCellCap) {
        const int i = (int)(idx % n);
        const int z = i % res, y = (i / res) % res, x = i / (res * res);
        const int R = bdist[((idx / n) * nb + (x >> kBrickShift)) * nb * nb + (int64_t)(y >> kBrickShift) * nb + (z >> kBrickShift)];
        const int bound = R >= 1 ? (R - 1) * kBrick + 1 : 0;
        v = max(kCellCap + 1, bound);
        cdist[idx] = (uint8_t)min(v, 255);
    }
}

static inline int64_t align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// layout of the accel buffer: [bdist m*nb^3][brick scratch m*nb^3] pad [cdist m*res^3][cell scratch m*res^3]
AccelSpec accel_view(const void *accel, int n_grids, int res, bool with_cells)
{
    const int nb = (res + kBrick - 1) / kBrick;
    const uint8_t *base = (const uint8_t *)accel;
    return AccelSpec{ base, nb, with_cells ? base + align256(2 * (int64_t)n_grids * nb * nb * nb) : nullptr };
}

int build_brick_accel(const uint8_t *binaries, int n_grids, int res, uint8_t *dist, uint8_t *scratch, hipStream_t stream)
{
    const int nb = (res + kBrick - 1) / kBrick;
    const int n_bricks = n_grids * nb * nb * nb;
    hipLaunchKernelGGL(brick_any_kernel, dim3(n_bricks), dim3(64), 0, stream, binaries, res, nb, dist);
    if (nb <= 32) {
        hipLaunchKernelGGL(brick_dist_lds_kernel, dim3(n_grids), dim3(1024), (size_t)2 * nb * nb * nb, stream, dist, nb);
    } else {
        // ping-pong dist <-> scratch; copied back if the last round landed in the scratch half
        uint8_t *src = dist, *dst = scratch;
        for (int r = 1; r <= kDistCap; ++r) {
            hipLaunchKernelGGL(brick_dist_round_kernel, dim3((n_bricks + 255) / 256), dim3(256), 0, stream, src, dst, n_grids,
                               nb, r, r == kDistCap ? 1 : 0);
            uint8_t *t = src; src = dst; dst = t;
        }
        if (src != dist && hipMemcpyAsync(dist, src, (size_t)n_bricks, hipMemcpyDeviceToDevice, stream) != hipSuccess)
            return check_launch("build_occupancy_accel (copy)");
    }
    return check_launch("build_occupancy_accel (bricks)");
}

int build_full_accel(const uint8_t *binaries, int n_grids, int res, uint8_t *accel, hipStream_t stream)
{
    const int64_t nb = (res + kBrick - 1) / kBrick;
    const int64_t n_bricks = n_grids * nb * nb * nb, n_cells = (int64_t)n_grids * res * res * res;
    uint8_t *bdist = accel, *cdist = accel + align256(2 * n_bricks), *cscr = cdist + n_cells;
    int rc = build_brick_accel(binaries, n_grids, res, bdist, bdist + n_bricks, stream);
    if (rc) return rc;
    const dim3 blk(256), grd((unsigned)((n_cells + 255) / 256));
    hipLaunchKernelGGL(cell_dist_z_kernel, dim3((unsigned)(((int64_t)n_grids * res * res + 255) / 256)), blk, 0, stream, binaries,
                       n_grids, res, cdist);
    hipLaunchKernelGGL(cell_dist_axis_kernel, grd, blk, 0, stream, cdist, cscr, n_grids, res, 1);
    hipLaunchKernelGGL(cell_dist_axis_kernel, grd, blk, 0, stream, cscr, cdist, n_grids, res, 0);
    hipLaunchKernelGGL(cell_dist_combine_kernel, grd, blk, 0, stream, cdist, bdist, n_grids, res, (int)nb);
    return check_launch("build_occupancy_accel (cells)");
}

static void host_build_bricks(const uint8_t *binaries, int n_grids, int res, uint8_t *dist)
{
    const int nb = (res + kBrick - 1) / kBrick;
    const size_t n = (size_t)nb * nb * nb;
    for (int lvl = 0; lvl < n_grids; ++lvl) {
        const uint8_t *g = binaries + (size_t)lvl * res * res * res;
        uint8_t *lv = dist + lvl * n;
        for (size_t i = 0; i < n; ++i) lv[i] = 255;
        for (int x = 0; x < res; ++x)
            for (int y = 0; y < res; ++y)
                for (int z = 0; z < res; ++z)
                    if (g[((size_t)x * res + y) * res + z]) lv[((size_t)(x / kBrick) * nb + y / kBrick) * nb + z / kBrick] = 0;
        std::vector<uint8_t> nxt(n);
        for (int r = 1; r <= kDistCap; ++r) {
            for (size_t i = 0; i < n; ++i) {
                uint8_t v = lv[i];
                if (v == 255) {
                    const int bz = (int)(i % nb), by = (int)((i / nb) % nb), bx = (int)(i / ((size_t)nb * nb));
                    bool reach = false;
                    for (int x = bx > 0 ? bx - 1 : 0; x <= (bx + 1 < nb ? bx + 1 : nb - 1); ++x)
                        for (int y = by > 0 ? by - 1 : 0; y <= (by + 1 < nb ? by + 1 : nb - 1); ++y)
                            for (int z = bz > 0 ? bz - 1 : 0; z <= (bz + 1 < nb ? bz + 1 : nb - 1); ++z)
                                reach |= lv[((size_t)x * nb + y) * nb + z] != 255;
                    if (reach) v = (uint8_t)r;
                }
                nxt[i] = v;
            }
            for (size_t i = 0; i < n; ++i) lv[i] = nxt[i];
        }
        for (size_t i = 0; i < n; ++i)
            if (lv[i] == 255) lv[i] = (uint8_t)(kDistCap + 1);
    }
}

// the same three passes as the device kernels, on the host
static void host_build_cells(const uint8_t *binaries, int n_grids, int res, const uint8_t *bdist, uint8_t *cdist)
{
    const int nb = (res + kBrick - 1) / kBrick;
    const int64_t n = (int64_t)res * res * res;
    std::vector<uint8_t> tmp((size_t)n);
    for (int lvl = 0; lvl < n_grids; ++lvl) {
        const uint8_t *g = binaries + lvl * n;
        uint8_t *out = cdist + lvl * n;
        for (int64_t col = 0; col < (int64_t)res * res; ++col) {
            int dmin = kCellCap + 1;
            for (int z = 0; z < res; ++z) { dmin = g[col * res + z] ? 0 : std::min(dmin + 1, kCellCap + 1); out[col * res + z] = (uint8_t)dmin; }
            dmin = kCellCap + 1;
            for (int z = res - 1; z >= 0; --z) {
                dmin = g[col * res + z] ? 0 : std::min(dmin + 1, kCellCap + 1);
                out[col * res + z] = (uint8_t)std::min((int)out[col * res + z], dmin);
            }
        }
        for (int axis = 1; axis >= 0; --axis) {
            const int64_t stride = axis == 0 ? (int64_t)res * res : res;
            for (int64_t i = 0; i < n; ++i) {
                const int y = (int)((i / res) % res), x = (int)(i / ((int64_t)res * res));
                const int pos = axis == 0 ? x : y;
                int best = out[i];
                for (int k = 1; k < best && k <= kCellCap; ++k) {
                    if (pos - k >= 0) best = std::min(best, std::max(k, (int)out[i - k * stride]));
                    if (pos + k < res) best = std::min(best, std::max(k, (int)out[i + k * stride]));
                }
                tmp[(size_t)i] = (uint8_t)best;
            }
            for (int64_t i = 0; i < n; ++i) out[i] = tmp[(size_t)i];
        }
        for (int64_t i = 0; i < n; ++i) {
            int v = out[i];
            if (v > kCellCap) {
                const int z = (int)(i % res), y = (int)((i / res) % res), x = (int)(i / ((int64_t)res * res));
                const int R = bdist[(((size_t)lvl * nb + (x >> kBrickShift)) * nb + (y >> kBrickShift)) * nb + (z >> kBrickShift)];
                const int bound = R >= 1 ? (R - 1) * kBrick + 1 : 0;
                out[i] = (uint8_t)std::min(std::max(kCellCap + 1, bound), 255);
            }
        }
    }
}

}  // namespace ced

extern "C" int64_t ced_occupancy_accel_bytes(int32_t n_grids, int32_t res)
{
    if (n_grids < 1 || res < 1 || res > 1024) return -1;
    const int64_t nb = (res + ced::kBrick - 1) / ced::kBrick;
    // brick field + scratch, then cell field + scratch (layout: ced::accel_view)
    return ced::align256(2 * (int64_t)n_grids * nb * nb * nb) + 2 * (int64_t)n_grids * res * res * res;
}

extern "C" int ced_build_occupancy_accel(const uint8_t *binaries, int32_t n_grids, int32_t res, void *accel,
                                         int64_t accel_bytes, void *stream)
{
    CED_REQUIRE(binaries && accel, "build_occupancy_accel: null pointer");
    const int64_t need = ced_occupancy_accel_bytes(n_grids, res);
    CED_REQUIRE(need > 0 && accel_bytes >= need, "build_occupancy_accel: bad sizes (need %lld bytes, got %lld)",
                (long long)need, (long long)accel_bytes);
    return ced::build_full_accel(binaries, n_grids, res, (uint8_t *)accel, (hipStream_t)stream);
}

// ---- HOST twins (no GPU needed): validation aids of the CPU test-suite --------------------------------------------
extern "C" int ced_host_build_occupancy_accel(const uint8_t *binaries_host, int32_t n_grids, int32_t res,
                                              uint8_t *accel_host)
{
    CED_REQUIRE(binaries_host && accel_host && n_grids >= 1 && res >= 1 && res <= 1024, "host_build_occupancy_accel: bad arguments");
    const int64_t nb = (res + ced::kBrick - 1) / ced::kBrick;
    ced::host_build_bricks(binaries_host, n_grids, res, accel_host);
    ced::host_build_cells(binaries_host, n_grids, res, accel_host, accel_host + ced::align256(2 * (int64_t)n_grids * nb * nb * nb));
    return CED_OK;
}

extern "C" int32_t ced_host_count_steps(float *x, float d, float tau, int32_t kcap, float *prev)
{
    return ced::count_steps(*x, d, tau, kcap, *prev);
}

extern "C" int ced_host_march_frame(int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *binaries,
                                    int32_t n_grids, int32_t res, const float *aabbs, const float *near_planes,
                                    float far_plane, float step_size, float cone_angle, int32_t limit,
                                    const float *t_sorted, const int64_t *t_indices, const uint8_t *hits,
                                    const uint8_t *accel_host, int32_t accel_mode, int32_t use_lattice, int32_t start_coarse,
                                    int32_t *counts, float *t_starts, float *t_ends, float *t_term)
{
    CED_REQUIRE(n_rays >= 0 && n_grids >= 1 && res >= 1 && limit >= 1, "host_march_frame: bad sizes");
    CED_REQUIRE(rays_o && rays_d && binaries && aabbs && near_planes && t_sorted && t_indices && hits && counts &&
                    t_starts && t_ends && t_term, "host_march_frame: null pointer");
    CED_REQUIRE(accel_mode >= 0 && accel_mode <= 2 && (accel_mode == 0 || accel_host), "host_march_frame: accel_mode 0 (none), 1 (bricks), 2 (cells)");
    // use_lattice: every near plane must be a point of the lattice that starts at near_planes[0] (as in a frame:
    // the frame's near plane, or termination planes of earlier iterations)
    float lattice[256];
    if (use_lattice) ced::build_lattice(near_planes[0], step_size, lattice);
    const ced::GridSpec G{ binaries, aabbs, n_grids, res, step_size, cone_angle, limit,
                           (use_lattice && cone_angle == 0.0f && step_size > 0.0f) ? lattice : nullptr };
    ced::AccelSpec S{ nullptr, (res + ced::kBrick - 1) / ced::kBrick, nullptr };
    if (accel_mode) S = ced::accel_view(accel_host, n_grids, res, accel_mode == 2);
    for (int64_t r = 0; r < n_rays; ++r) {
        const float o[3] = { rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2] };
        const float d[3] = { rays_d[3 * r], rays_d[3 * r + 1], rays_d[3 * r + 2] };
        float *t0 = t_starts + r * limit, *t1 = t_ends + r * limit;
        auto emit = [&](int i, float a, float b) { t0[i] = a; t1[i] = b; };
        const float *ts = t_sorted + r * 2 * n_grids;
        const int64_t *ti = t_indices + r * 2 * n_grids;
        const uint8_t *hr = hits + r * n_grids;
        // the kernels' instantiations: one grid level recomputes the ray/box interval, LOOK = 4 cells
        if (n_grids == 1)
            counts[r] = start_coarse == 1 ? ced::traverse_ray_frame<ced::kFrameLook, true, true>(G, S, true, o, d, near_planes[r], far_plane, ts, ti, hr, emit, t_term[r])
                                          : ced::traverse_ray_frame<ced::kFrameLook, true, false>(G, S, start_coarse != 0, o, d, near_planes[r], far_plane, ts, ti, hr, emit, t_term[r]);
        else
            counts[r] = start_coarse == 1 ? ced::traverse_ray_frame<ced::kFrameLook, false, true>(G, S, true, o, d, near_planes[r], far_plane, ts, ti, hr, emit, t_term[r])
                                        : ced::traverse_ray_frame<ced::kFrameLook, false, false>(G, S, start_coarse != 0, o, d, near_planes[r], far_plane, ts, ti, hr, emit, t_term[r]);
    }
    return CED_OK;
}
