// Frame renderer: the whole of render_image_test (cednerf/utils.py:153-318) behind one C call.
//
// Same algorithm and the same per-ray sample sets as the reference's host loop (image-global
// schedule N_samples = clamp(N_rays // N_alive, min, 64), termination checked between iterations),
// but an iteration is four launches instead of ~15 kernels + torch glue:
//   march_alloc : one lane per ALIVE ray marches it ONCE, stages the (t_start, t_end) pairs in LDS; the
//                 workgroup reserves one contiguous output range (wave prefix sums + a single returning
//                 atomic) and writes the ray-packed samples (no count pass, no scan);
//   field       : the fused field kernel (field.hip / field_half.hip) on those samples (count read from device memory);
//   composite   : per-ray front-to-back compositing over the alive list; survivors are appended to the next
//                 iteration's list, one range reservation per workgroup;
//   publish     : one thread copies the iteration's two counters into mapped pinned host memory and raises a
//                 sequence number.
// The host spins on that sequence number -- one round trip per iteration, the same single sync the
// reference pays at cednerf/utils.py:231.  Sample order in memory differs from nerfacc's (waves
// reserve ranges in arrival order) but every ray's samples are contiguous and in order, so pixels,
// sample counts and the schedule are bit-identical.
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdlib>

#include "ced_common.hpp"
#include "field_args.hpp"
#include "march_core.hpp"

namespace ced {

__device__ __forceinline__ bool slab_test(const float o[3], const float inv_d[3], const float *__restrict__ aabb,
                                          float &tmin_out, float &tmax_out)
{
    // nerfacc.ray_aabb_intersect with near = -inf, far = +inf (cednerf/utils.py:215)
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
    tmin_out = tmin;
    tmax_out = tmax;
    return true;
}

constexpr int kMaxGrids = 8;
constexpr int kMaxFrames = 8;       // frames rendered by one call (ced_render_frames_test)

// Several frames in one call: the rays of all frames are one array (frame f owns ray ids [f*rays_per_frame,
// (f+1)*rays_per_frame)), every frame keeps its OWN schedule (N_samples = clamp(N_rays // N_alive, min, 64) on its
// own counts, its own loop end) and its own alive list; a launch covers the frames' alive rays back to back, each
// frame's slot range starting at a multiple of 256 so that a workgroup never straddles two frames.
struct BatchMap {
    int n_frames;
    int rays_per_frame;
    int base[kMaxFrames];       // first slot of the frame in this launch
    int count[kMaxFrames];      // alive rays of the frame in this launch (0: the frame has finished)
    int limit[kMaxFrames];      // the frame's N_samples in this iteration
    int last[kMaxFrames];       // the frame's loop ends after this iteration (max_samples reached)
};

// frame of the workgroup whose first slot is s0 (block-uniform), -1 for a workgroup in the padding between frames
__device__ __forceinline__ int frame_of_slot(const BatchMap &B, int64_t s0)
{
    int f = -1;
    for (int k = 0; k < B.n_frames; ++k)
        if (B.count[k] > 0 && s0 >= B.base[k] && s0 < (int64_t)B.base[k] + B.count[k]) f = k;
    return f;
}

__global__ __launch_bounds__(256) void frame_times_kernel(int64_t n_rays, int rays_per_frame,
                                                          const float *__restrict__ frame_times, float *__restrict__ ts_ray)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rays) ts_ray[r] = frame_times[r / rays_per_frame];
}

// Per-ray setup of cednerf/utils.py:197-225: zero the pixel accumulators, all rays alive, near
// planes, ray/AABB intersection per grid level and the stably sorted entry/exit event list.
__global__ __launch_bounds__(256) void frame_prep_kernel(int64_t n_rays, const float *__restrict__ rays_o,
                                                         const float *__restrict__ rays_d, int m,
                                                         const float *__restrict__ aabbs, float near_plane,
                                                         float *__restrict__ t_sorted, int64_t *__restrict__ t_indices,
                                                         uint8_t *__restrict__ hits, float *__restrict__ near_planes,
                                                         float *__restrict__ rgb,
                                                         float *__restrict__ opacity, float *__restrict__ depth)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const float o[3] = { rays_o[3 * r], rays_o[3 * r + 1], rays_o[3 * r + 2] };
    const float inv_d[3] = { 1.0f / rays_d[3 * r], 1.0f / rays_d[3 * r + 1], 1.0f / rays_d[3 * r + 2] };
    float ev[2 * kMaxGrids];
    int id[2 * kMaxGrids];
    for (int a = 0; a < m; ++a) {
        float t0 = __builtin_inff(), t1 = __builtin_inff();
        bool hit = slab_test(o, inv_d, aabbs + 6 * a, t0, t1);
        if (!hit) { t0 = __builtin_inff(); t1 = __builtin_inff(); }
        hits[r * m + a] = hit ? 1 : 0;
        ev[a] = t0; id[a] = a;
        ev[m + a] = t1; id[m + a] = m + a;
    }
    if (m > 1) {            // stable insertion sort of the 2m events (torch.sort(..., stable=True))
        for (int i = 1; i < 2 * m; ++i) {
            float v = ev[i];
            int k = id[i], j = i - 1;
            while (j >= 0 && ev[j] > v) { ev[j + 1] = ev[j]; id[j + 1] = id[j]; --j; }
            ev[j + 1] = v; id[j + 1] = k;
        }
    }
    for (int i = 0; i < 2 * m; ++i) {
        t_sorted[r * 2 * m + i] = ev[i];
        t_indices[r * 2 * m + i] = id[i];
    }
    near_planes[r] = near_plane;
    rgb[3 * r] = 0.0f; rgb[3 * r + 1] = 0.0f; rgb[3 * r + 2] = 0.0f;
    opacity[r] = 0.0f;
    depth[r] = 0.0f;
}

// brick_any[b] = any occupied cell in brick b (kBrick^3 cells).  One 64-lane wave per brick: lane
// (x, y) ORs the 8 contiguous z bytes of its row, then a wave-wide ballot.
__global__ __launch_bounds__(64) void brick_any_kernel(const uint8_t *__restrict__ binaries, int m, int res, int nb,
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
    if (threadIdx.x == 0) any[idx] = any_lane ? 1 : 0;
}

// dilated[b] = any of the 3x3x3 bricks around b
__global__ __launch_bounds__(256) void brick_dilate_kernel(const uint8_t *__restrict__ any, int m, int nb,
                                                           uint8_t *__restrict__ dil)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * nb * nb * nb) return;
    const int bz = idx % nb, by = (idx / nb) % nb, bx = (idx / (nb * nb)) % nb, lvl = idx / (nb * nb * nb);
    const uint8_t *a = any + (size_t)lvl * nb * nb * nb;
    uint8_t acc = 0;
    for (int x = max(bx - 1, 0); x <= min(bx + 1, nb - 1); ++x)
        for (int y = max(by - 1, 0); y <= min(by + 1, nb - 1); ++y)
            for (int z = max(bz - 1, 0); z <= min(bz + 1, nb - 1); ++z) acc |= a[(x * nb + y) * nb + z];
    dil[idx] = acc;
}

struct MarchArgs {
    int64_t n_rays;
    const float *rays_o, *rays_d;
    GridSpec grid;
    float *near_planes;            // in: near plane, out: termination plane (cednerf/utils.py:301)
    float far_plane;
    const int32_t *alive;          // per-frame lists of the rays still alive (NULL: all rays, first iteration)
    BatchMap map;
    const float *t_sorted;
    const int64_t *t_indices;
    const uint8_t *hits;
    float *t_starts, *t_ends;
    int32_t *ray_idx;
    int32_t *packed;               // [n_rays, 2] (start, count)
    unsigned long long *counter;   // samples reserved so far in this iteration
};

// Workgroup of T threads (T/64 waves).  Each ray marches once and stages its (t_start, t_end) pairs
// in LDS (dynamic LDS = T * limit * 8 bytes, slot-major per wave so a wave's stores of slot i are 512
// contiguous bytes); the workgroup then reserves ONE contiguous output range with a single returning
// atomic -- a returning atomic on one word sustains only ~88 ops/us on this chip, so one per wave
// (10^4 per launch) would cost more than the marching itself.
__global__ __launch_bounds__(256) void march_alloc_kernel(MarchArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float2 stage_all[];
    __shared__ int wave_tot[16];
    __shared__ long long block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    const int64_t s0 = (int64_t)blockIdx.x * blockDim.x;
    const int f = frame_of_slot(A.map, s0);
    if (f < 0) return;                                   // padding between two frames' slot ranges (whole workgroup)
    const int limit = A.map.limit[f];
    float2 *stage = stage_all + (size_t)wave * limit * 64;
    const int64_t idx = s0 + threadIdx.x - A.map.base[f];
    const bool active = idx < A.map.count[f];
    const int64_t first = (int64_t)f * A.map.rays_per_frame;
    const int64_t r = active ? (A.alive ? (int64_t)A.alive[first + idx] : first + idx) : 0;
    GridSpec grid = A.grid;
    grid.limit = limit;
    int n = 0;
#ifdef CED_MARCH_PROFILE
    unsigned long long mp_acc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    unsigned long long mp_t = __builtin_readcyclecounter();
#endif
    if (active) {
        const float o[3] = { A.rays_o[3 * r], A.rays_o[3 * r + 1], A.rays_o[3 * r + 2] };
        const float d[3] = { A.rays_d[3 * r], A.rays_d[3 * r + 1], A.rays_d[3 * r + 2] };
        const int m = grid.n_grids;
        float t_term;
        n = traverse_ray(
            grid, o, d, A.near_planes[r], A.far_plane, A.t_sorted + r * 2 * m, A.t_indices + r * 2 * m,
            A.hits + r * m, [&](int i, float t0, float t1) { stage[i * 64 + lane] = make_float2(t0, t1); }, t_term
#ifdef CED_MARCH_PROFILE
            , mp_acc, mp_t
#endif
        );
        A.near_planes[r] = t_term;
    }
    // wave-inclusive prefix sum of the counts
    int incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < n_waves; ++w) { int t = wave_tot[w]; wave_tot[w] = run; run += t; }
        block_base = run > 0 ? (long long)atomicAdd(A.counter, (unsigned long long)run) : 0;
    }
    __syncthreads();
    const int64_t start = (int64_t)block_base + wave_tot[wave] + (incl - n);
    if (active) {
        A.packed[2 * r] = (int32_t)start;
        A.packed[2 * r + 1] = n;
    }
    for (int i = 0; i < n; ++i) {
        const float2 v = stage[i * 64 + lane];
        A.t_starts[start + i] = v.x;
        A.t_ends[start + i] = v.y;
        A.ray_idx[start + i] = (int32_t)r;
    }
#ifdef CED_MARCH_PROFILE
    CED_MP(7)                       // [7] prefix sum, range reservation, copy-out
    CED_MP_FLUSH
#endif
}

#ifdef CED_MARCH_PROFILE
extern "C" int ced_debug_march_profile(unsigned long long *out16, int reset)
{
    if (out16 && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_march_prof), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = { 0 };
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_march_prof), z, sizeof z) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// composite_prefix (cednerf/utils.py:274-299) + ray bookkeeping (utils.py:301-307) over the list of
// alive rays; survivors (opacity <= threshold and a full sample budget) are appended to the next
// iteration's list, one range reservation per workgroup.
__global__ __launch_bounds__(256) void frame_composite_kernel(BatchMap map, const int32_t *__restrict__ alive_list,
                                                              int32_t *__restrict__ next_list,
                                                              unsigned long long *__restrict__ next_count,
                                                              const int32_t *__restrict__ packed,
                                                              const float *__restrict__ t0,
                                                              const float *__restrict__ t1,
                                                              const float *__restrict__ sig,
                                                              const float *__restrict__ rgbs, float *__restrict__ rgb,
                                                              float *__restrict__ opacity, float *__restrict__ depth,
                                                              float opc_thres)
{
    __shared__ int wave_alive[4], wave_samples[4];
    __shared__ long long block_base;
    const int64_t s0 = (int64_t)blockIdx.x * blockDim.x;
    const int f = frame_of_slot(map, s0);
    if (f < 0) return;                                   // padding between two frames' slot ranges (whole workgroup)
    const int n_samples_iter = map.limit[f];
    const int64_t idx = s0 + threadIdx.x - map.base[f];
    const bool active = idx < map.count[f];
    const int64_t first = (int64_t)f * map.rays_per_frame;
    const int64_t r = active ? (alive_list ? (int64_t)alive_list[first + idx] : first + idx) : 0;
    int cnt = 0;
    bool alive = false;
    if (active) {
        const int s0 = packed[2 * r];
        cnt = packed[2 * r + 1];
        float op = opacity[r];
        if (cnt > 0) {
            const float prefix = 1.0f - op;
            float c0 = rgb[3 * r], c1 = rgb[3 * r + 1], c2 = rgb[3 * r + 2], dp = depth[r];
            float acc = 0.0f;
            // Samples are consumed strictly in order (the per-ray sums are sequential by contract), but
            // their loads are issued kU at a time so one memory round trip feeds kU samples.
            constexpr int kU = 4;
            int i = s0;
            const int end = s0 + cnt;
            for (; i + kU <= end; i += kU) {
                float ts[kU], te[kU], sg[kU], cr[kU], cg[kU], cb[kU];
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    ts[u] = t0[i + u]; te[u] = t1[i + u]; sg[u] = sig[i + u];
                    cr[u] = rgbs[3 * (i + u)]; cg[u] = rgbs[3 * (i + u) + 1]; cb[u] = rgbs[3 * (i + u) + 2];
                }
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    float sd = sg[u] * (te[u] - ts[u]);
                    float a = 1.0f - det_expf(-sd);
                    float t = det_expf(-acc) * prefix;
                    float w = t * a;
                    c0 = c0 + w * cr[u];
                    c1 = c1 + w * cg[u];
                    c2 = c2 + w * cb[u];
                    op = op + w;
                    dp = dp + w * ((ts[u] + te[u]) / 2.0f);
                    acc = acc + sd;
                }
            }
            for (; i < end; ++i) {
                float ts = t0[i], te = t1[i];
                float sd = sig[i] * (te - ts);
                float a = 1.0f - det_expf(-sd);
                float t = det_expf(-acc) * prefix;
                float w = t * a;
                c0 = c0 + w * rgbs[3 * i];
                c1 = c1 + w * rgbs[3 * i + 1];
                c2 = c2 + w * rgbs[3 * i + 2];
                op = op + w;
                dp = dp + w * ((ts + te) / 2.0f);
                acc = acc + sd;
            }
            rgb[3 * r] = c0; rgb[3 * r + 1] = c1; rgb[3 * r + 2] = c2;
            opacity[r] = op;
            depth[r] = dp;
        }
        alive = !map.last[f] && (op <= opc_thres) && (cnt == n_samples_iter);
    }
    const unsigned long long ballot = __ballot(alive);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wave_alive[wave] = __builtin_popcountll(ballot);
    // several frames per call: the frame's sample count of this iteration rides in the high word of the same atomic
    // that reserves the survivors' range (low word), so the per-frame totals cost no extra atomic
    const bool count_samples = map.n_frames > 1;
    if (count_samples) {
        int wsum = cnt;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wsum += __shfl_xor(wsum, off, 64);
        if (lane == 0) wave_samples[wave] = wsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < 4; ++w) { int t = wave_alive[w]; wave_alive[w] = run; run += t; }
        unsigned long long add = (unsigned long long)run;
        if (count_samples)
            add += (unsigned long long)(wave_samples[0] + wave_samples[1] + wave_samples[2] + wave_samples[3]) << 32;
        block_base = add != 0 ? (long long)(atomicAdd(next_count + f, add) & 0xffffffffull) : 0;
    }
    __syncthreads();
    if (alive) {
        const int rank = __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
        next_list[first + block_base + wave_alive[wave] + rank] = (int32_t)r;
    }
}

__global__ __launch_bounds__(256) void frame_finalize_kernel(int64_t n_rays, const float *__restrict__ bkgd,
                                                             float *__restrict__ rgb, const float *__restrict__ opacity,
                                                             float *__restrict__ depth)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float op = opacity[r];
    if (bkgd) {
        float rem = 1.0f - op;
        rgb[3 * r] = rgb[3 * r] + bkgd[0] * rem;
        rgb[3 * r + 1] = rgb[3 * r + 1] + bkgd[1] * rem;
        rgb[3 * r + 2] = rgb[3 * r + 2] + bkgd[2] * rem;
    }
    depth[r] = depth[r] / __builtin_fmaxf(op, FLT_EPSILON);
}

// Hands the iteration's two counters to the host through mapped pinned memory and raises a sequence
// number; the host spins on it instead of paying a copy + hipStreamSynchronize round trip.
__global__ void frame_publish_kernel(const unsigned long long *__restrict__ it_counters,
                                     const unsigned long long *__restrict__ next_counters, int n_frames, long long *host,
                                     long long seq)
{
    // per-iteration counter block: [0] samples reserved in the iteration; [2+f]: low word = rays of frame f alive
    // entering the iteration, high word (several frames per call) = samples of frame f in the PREVIOUS iteration
    long long alive_next = 0;
    for (int f = 0; f < n_frames; ++f) alive_next += (long long)(next_counters[2 + f] & 0xffffffffull);
    host[0] = (long long)it_counters[0];    // samples reserved in this iteration
    host[1] = alive_next;                   // rays alive entering the next iteration
    if (n_frames > 1)
        for (int f = 0; f < n_frames; ++f) {
            host[3 + f] = (long long)(next_counters[2 + f] & 0xffffffffull);
            host[3 + n_frames + f] = (long long)(next_counters[2 + f] >> 32);
        }
    __threadfence_system();
    __hip_atomic_store(&host[2], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct FrameWorkspace {
    float *t_sorted; int64_t *t_indices; uint8_t *hits; float *near; int32_t *packed;
    int32_t *alive_a, *alive_b;     // double-buffered list of alive ray ids
    unsigned long long *counters;   // [iters+2][2+F]: see frame_publish_kernel
    float *ts_ray;                  // per-ray time of a multi-frame call
    float *t0, *t1; int32_t *ridx; float *sigma, *rgbs;
    uint8_t *brick_any, *brick_dil;
    size_t bytes;
};

static FrameWorkspace carve(void *base, int64_t n, int m, int64_t cap, int max_iters, int res, int n_frames = 1,
                            bool per_ray_times = false)
{
    FrameWorkspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return (char *)base + o; };
    w.t_sorted = (float *)take((size_t)n * 2 * m * 4);
    w.t_indices = (int64_t *)take((size_t)n * 2 * m * 8);
    w.hits = (uint8_t *)take((size_t)n * m);
    w.near = (float *)take((size_t)n * 4);
    w.packed = (int32_t *)take((size_t)n * 8);
    w.alive_a = (int32_t *)take((size_t)n * 4);
    w.alive_b = (int32_t *)take((size_t)n * 4);
    w.counters = (unsigned long long *)take((size_t)(max_iters + 2) * (2 + n_frames) * 8);
    w.ts_ray = (float *)take(per_ray_times ? (size_t)n * 4 : 0);
    w.t0 = (float *)take((size_t)cap * 4);
    w.t1 = (float *)take((size_t)cap * 4);
    w.ridx = (int32_t *)take((size_t)cap * 4);
    w.sigma = (float *)take((size_t)cap * 4);
    w.rgbs = (float *)take((size_t)cap * 12);
    const int nb = (res + kBrick - 1) / kBrick;
    w.brick_any = (uint8_t *)take((size_t)m * nb * nb * nb);
    w.brick_dil = (uint8_t *)take((size_t)m * nb * nb * nb);
    w.bytes = off;
    return w;
}

static std::atomic<long long> g_publish_seq{ 0 };

static inline int min_samples_of(float cone_angle) { return cone_angle == 0.0f ? 1 : 4; }

// The frame loop for `n_frames` frames of `rays_per_frame` rays each (ced_render_image_test: one frame).
// frame_times == nullptr: `timestamps` is what the field kernel gets ([1] shared or [n_rays] per ray, t_per_ray);
// otherwise frame_times [n_frames] holds one time per frame and is expanded to a per-ray array.
static int render_frames_impl(const ced_field_desc *field, int n_frames, int64_t rays_per_frame, const float *rays_o,
                              const float *rays_d, const uint8_t *binaries, int32_t n_grids, int32_t res,
                              const float *aabbs, float near_plane, float far_plane, float step_size, float cone_angle,
                              float early_stop_eps, int32_t max_samples, const float *timestamps, int32_t t_per_ray,
                              const float *frame_times, const float *bkgd, float *rgb, float *opacity, float *depth,
                              void *workspace, int64_t workspace_bytes, int64_t *host_stats, int64_t *total_samples_out,
                              ced_frame_trace *trace, void *field_stream_, void *stream_, const char *who)
{
    hipStream_t stream = (hipStream_t)stream_;
    // Optional separate stream for the field kernel: callers that keep several frames in flight hand
    // every frame the same field stream, so the MFMA-bound field launches of different frames queue
    // behind each other (overlapping them buys nothing) while the latency-bound marching / compositing
    // launches and the host hand-shake of one frame run beside the field kernel of another.
    hipStream_t fstream = field_stream_ ? (hipStream_t)field_stream_ : stream;
    const bool split = fstream != stream;
    static thread_local hipEvent_t ev_to_field = nullptr, ev_from_field = nullptr;
    if (split && !ev_to_field) {
        if (hipEventCreateWithFlags(&ev_to_field, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&ev_from_field, hipEventDisableTiming) != hipSuccess)
            return check_launch("render_image_test (event create)");
    }
    const int64_t n_rays = (int64_t)n_frames * rays_per_frame;
    CED_REQUIRE(field != nullptr, "%s: null field descriptor", who);
    CED_REQUIRE(n_frames >= 1 && n_frames <= kMaxFrames, "%s: n_frames must be 1..%d", who, kMaxFrames);
    CED_REQUIRE(rays_per_frame >= 0 && n_grids >= 1 && n_grids <= kMaxGrids && res >= 1 && res <= 1024, "%s: bad sizes", who);
    CED_REQUIRE(n_rays < (1ll << 31) / 4, "%s: too many rays for 32-bit sample indices", who);
    CED_REQUIRE(max_samples >= 0, "%s: max_samples < 0", who);
    if (total_samples_out)
        for (int f = 0; f < n_frames; ++f) total_samples_out[f] = 0;
    if (trace) trace->n_iters = 0;
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(rays_o && rays_d && binaries && aabbs && (timestamps || frame_times) && rgb && opacity && depth &&
                    workspace && host_stats,
                "%s: null pointer", who);
    const int min_samples = min_samples_of(cone_angle);
    const int64_t cap = n_rays * min_samples;
    FrameWorkspace W = carve(workspace, n_rays, n_grids, cap, max_samples + 1, res, n_frames, frame_times != nullptr);
    CED_REQUIRE((int64_t)W.bytes <= workspace_bytes, "%s: workspace too small (%lld < %lld bytes)", who,
                (long long)workspace_bytes, (long long)W.bytes);
    const dim3 blk(256), grd((unsigned)((n_rays + 255) / 256));
    static bool lds_attr_set = false;
    if (!lds_attr_set) {        // the marching kernel stages up to 128 KB of samples per workgroup
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(march_alloc_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) != hipSuccess)
            return check_launch("render_image_test (LDS attribute)");
        lds_attr_set = true;
    }

    hipLaunchKernelGGL(frame_prep_kernel, grd, blk, 0, stream, n_rays, rays_o, rays_d, (int)n_grids, aabbs, near_plane,
                       W.t_sorted, W.t_indices, W.hits, W.near, rgb, opacity, depth);
    if (frame_times)
        hipLaunchKernelGGL(frame_times_kernel, grd, blk, 0, stream, n_rays, (int)rays_per_frame, frame_times, W.ts_ray);
    const size_t cstride = 2 + (size_t)n_frames;                  // counters per iteration (see frame_publish_kernel)
    if (hipMemsetAsync(W.counters, 0, (size_t)(max_samples + 3) * cstride * 8, stream) != hipSuccess)
        return check_launch("render_image_test (memset)");
    const int nb = (res + kBrick - 1) / kBrick;
    const int n_bricks = n_grids * nb * nb * nb;
    hipLaunchKernelGGL(brick_any_kernel, dim3(n_bricks), dim3(64), 0, stream, binaries, (int)n_grids, (int)res, nb,
                       W.brick_any);
    hipLaunchKernelGGL(brick_dilate_kernel, dim3((n_bricks + 255) / 256), blk, 0, stream, W.brick_any, (int)n_grids, nb,
                       W.brick_dil);
    int rc = check_launch("render_image_test (prep)");
    if (rc) return rc;

    const float opc_thres = (float)(1.0 - (double)early_stop_eps);
    int64_t alive[kMaxFrames], total[kMaxFrames];
    int iter_samples[kMaxFrames];
    for (int f = 0; f < n_frames; ++f) { alive[f] = rays_per_frame; total[f] = 0; iter_samples[f] = 0; }
    int it = 0;
    for (;;) {
        // every frame advances its own reference loop (cednerf/utils.py:227-238): while iteration < max_samples and
        // rays are alive, N_samples = clamp(N_rays // N_alive, min, 64), iteration += N_samples
        BatchMap B{};
        B.n_frames = n_frames;
        B.rays_per_frame = (int)rays_per_frame;
        int64_t slots = 0, upper = 0, alive_total = 0;
        int max_limit = 0;
        for (int f = 0; f < n_frames; ++f) {
            if (alive[f] <= 0 || iter_samples[f] >= max_samples) continue;
            const int64_t q = rays_per_frame / alive[f];
            int n_samples = (int)(q < 64 ? q : 64);
            if (n_samples < min_samples) n_samples = min_samples;
            iter_samples[f] += n_samples;
            B.base[f] = (int)slots;
            B.count[f] = (int)alive[f];
            B.limit[f] = n_samples;
            B.last[f] = iter_samples[f] >= max_samples ? 1 : 0;
            slots = (slots + alive[f] + 255) & ~(int64_t)255;
            upper += alive[f] * n_samples;
            alive_total += alive[f];
            if (n_samples > max_limit) max_limit = n_samples;
        }
        if (alive_total == 0) break;
        unsigned long long *it_counters = W.counters + (size_t)it * cstride;
        unsigned long long *next_counters = W.counters + (size_t)(it + 1) * cstride;
        unsigned long long *counter = it_counters;                                  // samples reserved in this iteration
        const int32_t *cur_list = it == 0 ? nullptr : ((it & 1) ? W.alive_a : W.alive_b);
        int32_t *next_list = (it & 1) ? W.alive_b : W.alive_a;

        MarchArgs M{ n_rays, rays_o, rays_d,
                     GridSpec{ binaries, aabbs, n_grids, res, step_size, cone_angle, max_limit,
                               g_march_early_out ? W.brick_dil : nullptr, nb, it > 0 ? 1 : 0 },
                     W.near, far_plane, cur_list, B, W.t_sorted, W.t_indices, W.hits, W.t0, W.t1, W.ridx, W.packed,
                     counter };
        // only alive rays get a lane; 128 rays per workgroup (CED_MARCH_THREADS): a workgroup lives as long as its
        // slowest ray (its waves meet at the range reservation), while one reservation still serves 128 rays
        static const int threads_env = [] {
            const char *e = getenv("CED_MARCH_THREADS");
            const int t = e ? atoi(e) : 128;
            return (t == 64 || t == 128 || t == 256) ? t : 128;
        }();
        const int threads = threads_env;
        hipLaunchKernelGGL(march_alloc_kernel, dim3((unsigned)(slots / threads)), dim3(threads),
                           (size_t)threads * max_limit * sizeof(float2), stream, M);
        rc = check_launch("render_image_test (march)");
        if (rc) return rc;

        FieldArgs F{};
        F.n = upper;                        // host-side upper bound; the kernel reads the exact count
        F.n_dev = reinterpret_cast<const int64_t *>(counter);
        F.rays_o = rays_o; F.rays_d = rays_d; F.ray_idx32 = W.ridx;
        F.t0 = W.t0; F.t1 = W.t1;
        F.timestamps = frame_times ? W.ts_ray : timestamps;
        F.rays_mode = 1; F.t_per_ray = (frame_times || t_per_ray) ? 1 : 0; F.want_rgb = 1;
        F.rgb = W.rgbs; F.sigma = W.sigma; F.geo = nullptr;
        if (split) {
            (void)hipEventRecord(ev_to_field, stream);
            (void)hipStreamWaitEvent(fstream, ev_to_field, 0);
        }
        if (trace && it < trace->capacity && trace->field_begin)
            (void)hipEventRecord((hipEvent_t)trace->field_begin[it], fstream);
        rc = launch_field(field, F, (void *)fstream);
        if (rc) return rc;
        if (trace && it < trace->capacity && trace->field_end)
            (void)hipEventRecord((hipEvent_t)trace->field_end[it], fstream);
        if (split) {
            (void)hipEventRecord(ev_from_field, fstream);
            (void)hipStreamWaitEvent(stream, ev_from_field, 0);
        }

        hipLaunchKernelGGL(frame_composite_kernel, dim3((unsigned)(slots / 256)), blk, 0, stream, B, cur_list, next_list,
                           next_counters + 2, W.packed, W.t0, W.t1, W.sigma, W.rgbs, rgb, opacity, depth, opc_thres);
        rc = check_launch("render_image_test (composite)");
        if (rc) return rc;
        // {samples of this iteration, rays alive for the next} -> pinned host memory; spin on the sequence
        // number (falls back to a stream synchronise if the flag does not show up)
        const long long seq = ++g_publish_seq;
        hipLaunchKernelGGL(frame_publish_kernel, dim3(1), dim3(1), 0, stream, it_counters, next_counters, n_frames,
                           (long long *)host_stats, seq);
        rc = check_launch("render_image_test (publish)");
        if (rc) return rc;
        {
            volatile long long *flag = (volatile long long *)host_stats + 2;
            const auto t_start = std::chrono::steady_clock::now();
            long spins = 0;
            while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 0xfff) == 0 &&
                    std::chrono::steady_clock::now() - t_start > std::chrono::milliseconds(200)) {
                    if (hipStreamSynchronize(stream) != hipSuccess) return check_launch("render_image_test (sync)");
                    break;
                }
            }
        }
        const int64_t samples_it = host_stats[0];
        if (trace && it < trace->capacity) {
            if (trace->iter_alive) trace->iter_alive[it] = alive_total;
            if (trace->iter_n_samples) trace->iter_n_samples[it] = max_limit;
            if (trace->iter_samples) trace->iter_samples[it] = samples_it;
        }
        for (int f = 0; f < n_frames; ++f) {
            if (B.count[f] == 0) continue;
            alive[f] = n_frames > 1 ? host_stats[3 + f] : host_stats[1];
            total[f] += n_frames > 1 ? host_stats[3 + n_frames + f] : samples_it;
        }
        ++it;
    }
    if (trace) trace->n_iters = it;
    hipLaunchKernelGGL(frame_finalize_kernel, grd, blk, 0, stream, n_rays, bkgd, rgb, opacity, depth);
    rc = check_launch("render_image_test (finalize)");
    if (rc) return rc;
    if (total_samples_out)
        for (int f = 0; f < n_frames; ++f) total_samples_out[f] = total[f];
    return CED_OK;
}

}  // namespace ced

extern "C" int64_t ced_render_image_test_workspace_bytes(int64_t n_rays, int32_t n_grids, int32_t res,
                                                         float cone_angle, int32_t max_samples)
{
    if (n_rays < 0 || n_grids < 1 || n_grids > ced::kMaxGrids || res < 1 || res > 1024 || max_samples < 0) return -1;
    const int64_t cap = n_rays * ced::min_samples_of(cone_angle);
    return (int64_t)ced::carve(nullptr, n_rays, n_grids, cap, max_samples + 1, res).bytes;
}

extern "C" int ced_render_image_test(const ced_field_desc *field, int64_t n_rays, const float *rays_o,
                                     const float *rays_d, const uint8_t *binaries, int32_t n_grids, int32_t res,
                                     const float *aabbs, float near_plane, float far_plane, float step_size,
                                     float cone_angle, float early_stop_eps, int32_t max_samples,
                                     const float *timestamps, int32_t t_per_ray, const float *bkgd, float *rgb,
                                     float *opacity, float *depth, void *workspace, int64_t workspace_bytes,
                                     int64_t *host_stats, int64_t *total_samples_out, ced_frame_trace *trace,
                                     void *field_stream_, void *stream_)
{
    return ced::render_frames_impl(field, 1, n_rays, rays_o, rays_d, binaries, n_grids, res, aabbs, near_plane, far_plane,
                                   step_size, cone_angle, early_stop_eps, max_samples, timestamps, t_per_ray, nullptr, bkgd,
                                   rgb, opacity, depth, workspace, workspace_bytes, host_stats, total_samples_out, trace,
                                   field_stream_, stream_, "render_image_test");
}

extern "C" int64_t ced_render_frames_test_workspace_bytes(int32_t n_frames, int64_t rays_per_frame, int32_t n_grids,
                                                          int32_t res, float cone_angle, int32_t max_samples)
{
    if (n_frames < 1 || n_frames > ced::kMaxFrames || rays_per_frame < 0 || n_grids < 1 || n_grids > ced::kMaxGrids ||
        res < 1 || res > 1024 || max_samples < 0)
        return -1;
    const int64_t n_rays = (int64_t)n_frames * rays_per_frame;
    const int64_t cap = n_rays * ced::min_samples_of(cone_angle);
    return (int64_t)ced::carve(nullptr, n_rays, n_grids, cap, max_samples + 1, res, n_frames, true).bytes;
}

extern "C" int ced_render_frames_test(const ced_field_desc *field, int32_t n_frames, int64_t rays_per_frame,
                                      const float *rays_o, const float *rays_d, const uint8_t *binaries, int32_t n_grids,
                                      int32_t res, const float *aabbs, float near_plane, float far_plane, float step_size,
                                      float cone_angle, float early_stop_eps, int32_t max_samples,
                                      const float *frame_times, const float *bkgd, float *rgb, float *opacity,
                                      float *depth, void *workspace, int64_t workspace_bytes, int64_t *host_stats,
                                      int64_t *total_samples_out, ced_frame_trace *trace, void *field_stream_,
                                      void *stream_)
{
    CED_REQUIRE(frame_times != nullptr, "render_frames_test: null frame_times");
    return ced::render_frames_impl(field, n_frames, rays_per_frame, rays_o, rays_d, binaries, n_grids, res, aabbs,
                                   near_plane, far_plane, step_size, cone_angle, early_stop_eps, max_samples, nullptr, 0,
                                   frame_times, bkgd, rgb, opacity, depth, workspace, workspace_bytes, host_stats,
                                   total_samples_out, trace, field_stream_, stream_, "render_frames_test");
}
