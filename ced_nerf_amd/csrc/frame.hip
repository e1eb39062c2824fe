// Frame renderer: the whole of render_image_test (cednerf/utils.py:153-318) behind one C call.
//
// Same algorithm and the same per-ray sample sets as the reference's host loop (image-global schedule
// N_samples = clamp(N_rays // N_alive, min, 64), termination checked between iterations), but
//   * an iteration is four launches instead of ~15 kernels + torch glue:
//       march     : one lane per ALIVE ray (march_accel.hpp: distance fields through empty space, exact DDA where it
//                   matters); a ray remembers its samples as runs in registers, the workgroup reserves one contiguous
//                   range of the iteration's sample array with a single atomic and the samples are regenerated
//                   there: ray-packed and dense, no count pass, no scan, no staging;
//       field     : the fused field kernel (field.hip / field_half.hip) on those samples (count read from device memory);
//       composite : per-ray front-to-back compositing over the alive list; survivors are appended to the next
//                   iteration's list, one range reservation per workgroup;
//       schedule  : one thread turns the survivor counts into the NEXT iteration's plan on the device -- rays alive,
//                   N_samples, slot and sample bases per frame -- and publishes (alive, done) to pinned host memory;
//   * THE HOST NEVER WAITS FOR AN ITERATION (the reference syncs at utils.py:231 every time): every kernel reads its
//     sizes from the plan in device memory, grids are sized by the last PUBLISHED alive count (an upper bound: the
//     count only falls) and the host enqueues up to `run_ahead` iterations beyond the last one it has seen
//     published; iterations enqueued after the frame has finished are empty launches.
// Pixels, per-iteration schedule and sample counts are bit-identical to the reference loop.
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cstdlib>
#include <vector>

#include "ced_common.hpp"
#include "field_args.hpp"
#include "march_accel.hpp"

#ifdef CED_MARCH_DIAG
// [phase][0] passes of a wave through the phase, [1] lanes active in those passes, [2] cycles of the wave between this
// tick and its next one.  Phases: 0 ray set-up, 1 segment set-up, 2 distance-field probe, 3 closed-form re-entry,
// 4 exact walk (one look-ahead batch), 5 emission at an occupied cell, 6 reservation + regeneration, 7 idle tail.
__device__ unsigned long long g_march_diag[8][3];
// per-wave accumulators in LDS (a tick costs a handful of instructions of one lane), flushed once per wave
__device__ __forceinline__ unsigned long long (*ced_diag_acc())[8][3]
{
    __shared__ unsigned long long acc[16][8][3];
    return acc;
}
__device__ __forceinline__ unsigned long long *ced_diag_state()
{
    __shared__ unsigned long long st[16][2];        // last tick's time and phase
    return &st[0][0];
}
__device__ void ced_diag_tick(int phase)
{
    const unsigned long long m = __ballot(1);
    const int wave = threadIdx.x >> 6;
    if ((int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) {
        unsigned long long(*acc)[8][3] = ced_diag_acc();
        unsigned long long *st = ced_diag_state() + 2 * wave;
        const unsigned long long now = __builtin_readcyclecounter();
        if (st[1] != 0x1234) {                         // first tick of the wave: clear
            for (int p = 0; p < 8; ++p) for (int k = 0; k < 3; ++k) acc[wave][p][k] = 0;
            st[1] = 0x1234;
        } else {
            acc[wave][(int)(st[0] >> 60) & 7][2] += (now - st[0]) & 0x0fffffffffffffffull;
        }
        st[0] = (now & 0x0fffffffffffffffull) | ((unsigned long long)phase << 60);
        acc[wave][phase][0] += 1;
        acc[wave][phase][1] += (unsigned long long)__popcll(m);
    }
}
// per-wave log: passes through phases 0..7, lanes in them, cycles in them (one row per wave and launch)
constexpr int kDiagWaves = 1 << 17;
__device__ unsigned int g_march_wave_log[kDiagWaves][24];
__device__ unsigned int g_march_wave_n;
__device__ void ced_diag_flush()
{
    const int wave = threadIdx.x >> 6;
    unsigned long long(*acc)[8][3] = ced_diag_acc();
    unsigned long long *st = ced_diag_state() + 2 * wave;
    if ((threadIdx.x & 63) == 0 && st[1] == 0x1234) {
        for (int p = 0; p < 8; ++p) for (int k = 0; k < 3; ++k) if (acc[wave][p][k]) atomicAdd(&g_march_diag[p][k], acc[wave][p][k]);
        const unsigned int row = atomicAdd(&g_march_wave_n, 1u);
        if (row < (unsigned)kDiagWaves)
            for (int p = 0; p < 8; ++p) for (int k = 0; k < 3; ++k) g_march_wave_log[row][3 * p + k] = (unsigned int)acc[wave][p][k];
        st[1] = 0;
    }
}
#endif

namespace ced {

int build_brick_accel(const uint8_t *binaries, int n_grids, int res, uint8_t *dist, uint8_t *scratch, hipStream_t stream);
AccelSpec accel_view(const void *accel, int n_grids, int res, bool with_cells);

constexpr int kMaxGrids = 8;
constexpr int kMaxFrames = 64;      // frames rendered by one call (ced_render_frames_test); one lane of the scheduling wave each
constexpr int kSlotAlign = 256;     // a frame's ray slots start at a multiple of this: no workgroup straddles two frames
#ifndef CED_MARCH_THREADS
#define CED_MARCH_THREADS 128
#endif
constexpr int kMarchThreads = CED_MARCH_THREADS;
constexpr int kCompositeThreads = 256;
#ifndef CED_COMPOSITE_KU
#define CED_COMPOSITE_KU 4
#endif
constexpr int kHostLatticeWord = 8;    // host_stats: words 0..2 publish {alive, done, seq}; the lattice table from word 8 on
constexpr int kHostIterWord = 8 + 128; // sharded calls: {alive here, done, seq} of plan k at word kHostIterWord + 3k

// The plan of ONE iteration, in device memory (written by make_next_plan, read by that iteration's launches).
// Several frames in one call: the rays of all frames are one array (frame f owns ray ids [f*rays_per_frame,
// (f+1)*rays_per_frame)); every frame keeps its OWN reference loop (N_samples on its own counts, its own end) and its
// own alive list; a launch covers the frames' alive rays back to back.
struct IterPlan {
    int32_t count[kMaxFrames];       // rays of frame f alive entering the iteration, in THIS process (0: none left here)
    int32_t gcount[kMaxFrames];      // the same over all processes that share the frame (= count unless the frame's rays
                                     // are sharded, ced_shard_exchange); 0: the frame has finished
    int32_t limit[kMaxFrames];       // the frame's N_samples in this iteration (cednerf/utils.py:235)
    int32_t last[kMaxFrames];        // the frame's loop ends after this iteration (max_samples reached, utils.py:229)
    int32_t used[kMaxFrames];        // the frame's iter_samples, this iteration included (utils.py:236)
    int32_t slot_base[kMaxFrames];   // first ray slot of the frame in this iteration's launches (multiple of kSlotAlign)
    int32_t samp_bound[kMaxFrames];  // count * limit: the frame's samples of the iteration are at most this many
    int32_t total_slots;             // end of the last frame's ray-slot range
    int32_t done;                    // nothing left: this and every later iteration is an empty launch
    int32_t n_cand;                  // first iteration: rays the culling pass could not rule out (march_cull_kernel)
    int32_t cursor;                  // first iteration, candidate list: next entry the persistent marching waves take
    int64_t sample_base;             // render_image: first entry of the iteration in the call's persistent sample arrays
    int64_t total_samples;           // samples RESERVED in the iteration so far: the marching workgroups add their totals
                                     // (ray-packed allocation); the field kernel's n once the marching launch is over
    // filled in DURING the iteration by the compositing kernel: low word = survivors appended to the frame's next
    // alive list, high word = samples the frame marched in this iteration
    unsigned long long next[kMaxFrames];
};

// frame of the workgroup whose first ray slot is s0 (block-uniform), -1 for the padding between two frames
__device__ __forceinline__ int frame_of_slot(const IterPlan &P, int n_frames, int64_t s0)
{
    // slot_base is non-decreasing and a frame without rays shares its base with the next one: the LAST frame whose
    // base is <= s0 is the only candidate (binary search; n_frames <= 64)
    int lo = 0, hi = n_frames - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)P.slot_base[mid] <= s0) lo = mid; else hi = mid - 1;
    }
    return (P.count[lo] > 0 && s0 >= P.slot_base[lo] && s0 < (int64_t)P.slot_base[lo] + P.count[lo]) ? lo : -1;
}

__global__ __launch_bounds__(256) void frame_times_kernel(int64_t n_rays, int rays_per_frame,
                                                          const float *__restrict__ frame_times, float *__restrict__ ts_ray)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_rays) ts_ray[r] = frame_times[r / rays_per_frame];
}

// Per-ray setup of cednerf/utils.py:197-225: zero the pixel accumulators, near planes, ray/AABB intersection per grid
// level and the stably sorted entry/exit event list.
__global__ __launch_bounds__(256) void frame_prep_kernel(int64_t n_rays, const float *__restrict__ rays_o,
                                                         const float *__restrict__ rays_d, int m,
                                                         const float *__restrict__ aabbs, float near_plane,
                                                         float *__restrict__ t_sorted, uint8_t *__restrict__ t_indices,
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
        t_indices[r * 2 * m + i] = (uint8_t)id[i];
    }
    near_planes[r] = near_plane;
    rgb[3 * r] = 0.0f; rgb[3 * r + 1] = 0.0f; rgb[3 * r + 2] = 0.0f;
    opacity[r] = 0.0f;
    depth[r] = 0.0f;
}

// One grid level: the marching recomputes the ray/box interval itself (march_accel.hpp, SINGLE), so the set-up is only
// the zeroing of the accumulators and the near planes.
__global__ __launch_bounds__(256) void frame_prep_single_kernel(int64_t n_rays, float near_plane, float *__restrict__ near_planes,
                                                                float *__restrict__ rgb, float *__restrict__ opacity,
                                                                float *__restrict__ depth)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    near_planes[r] = near_plane;
    rgb[3 * r] = 0.0f; rgb[3 * r + 1] = 0.0f; rgb[3 * r + 2] = 0.0f;
    opacity[r] = 0.0f;
    depth[r] = 0.0f;
}

// Every frame advances its own reference loop (cednerf/utils.py:227-238): while iter_samples < max_samples and rays
// are alive, N_samples = clamp(N_rays // N_alive, min, 64), iter_samples += N_samples.  it = -1 writes the plan of
// iteration 0 (all rays alive); otherwise the plan of iteration it + 1 from the survivor counts of iteration it.
// Then (alive rays, done) of the new plan go to pinned host memory behind a sequence number.
// ONE WAVE: lane f works out frame f, the frames' slot ranges are an exclusive prefix sum across the lanes.
//
// Sharded frames (ced_shard_exchange): the rays of frame f are dealt over several processes and the reference's loop is
// ONE loop per image -- N_rays and N_alive in utils.py:235 are the whole image's.  `xcounts` then holds, for every
// iteration and frame, the survivors summed over the processes (the exchange step between the compositing and this
// launch), `global_rays` the image's ray count: every process computes the same N_samples, the same last-iteration flag
// and the same end of the loop, so each ray gets exactly the samples it gets when one process renders the image.
struct ScheduleArgs {
    IterPlan *plans;
    int it, n_frames, rays_per_frame, min_samples, max_samples;
    long long *host;
    long long seq;
    const int32_t *local_rays;      // device [n_frames] or NULL: rays of frame f that are real (the rest of its
                                    // rays_per_frame slots is padding and never alive)
    const int64_t *xcounts;         // device [(max_iters + 1) * n_frames] or NULL: survivors over all processes
    int64_t global_rays;            // N_rays of the whole image when sharded, else 0 (= rays_per_frame)
    long long *host_iter;           // pinned, or NULL: {alive here, done, seq} of EVERY plan, plan k at word 3k
};

__device__ __forceinline__ void make_next_plan(const ScheduleArgs &S)
{
    const int f = threadIdx.x & 63;
    IterPlan &N = S.plans[S.it + 1];
    int local = 0, used = 0;
    long long global = 0;
    if (f < S.n_frames) {
        if (S.it < 0) {
            local = S.local_rays ? S.local_rays[f] : S.rays_per_frame;
            global = S.global_rays > 0 ? S.global_rays : local;
        } else {
            const IterPlan &P = S.plans[S.it];
            // the survivor counts were accumulated by device-scope atomics of other workgroups
            const unsigned long long nx = __hip_atomic_load(&P.next[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool finished = P.last[f] || P.gcount[f] == 0;
            local = (finished || P.count[f] == 0) ? 0 : (int)(nx & 0xffffffffull);
            global = finished ? 0 : (S.xcounts ? S.xcounts[(int64_t)S.it * S.n_frames + f] : local);
            used = P.used[f];
        }
    }
    int count = 0, gcount = 0, limit = 0, last = 0;
    if (global > 0 && used < S.max_samples) {
        const long long n_total = S.global_rays > 0 ? S.global_rays : S.rays_per_frame;
        const long long q = n_total / global;
        limit = q < 64 ? (int)q : 64;
        if (limit < S.min_samples) limit = S.min_samples;
        used += limit;
        last = used >= S.max_samples ? 1 : 0;
        count = local;
        gcount = (int)(global < 0x7fffffffll ? global : 0x7fffffffll);
    }
    // slot ranges: exclusive prefix sum of the counts rounded up to kSlotAlign
    const int padded = (count + kSlotAlign - 1) & ~(kSlotAlign - 1);
    int incl = padded;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (f >= off) incl += v;
    }
    long long alive_here = count, alive_all = gcount;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        alive_here += __shfl_xor(alive_here, off, 64);
        alive_all += __shfl_xor(alive_all, off, 64);
    }
    const int total_slots = __shfl(incl, 63, 64);
    N.count[f] = count; N.gcount[f] = gcount; N.limit[f] = limit; N.last[f] = last; N.used[f] = used;
    N.slot_base[f] = incl - padded; N.samp_bound[f] = count * limit;
    N.next[f] = 0;
    if (f == 0) {
        N.total_slots = total_slots;
        N.sample_base = S.it < 0 ? 0 : S.plans[S.it].sample_base + S.plans[S.it].total_samples;
        N.total_samples = 0;
        N.n_cand = 0;
        N.cursor = 0;
        N.done = alive_all == 0 ? 1 : 0;
        S.host[0] = alive_here;                 // rays alive entering iteration it + 1: an upper bound for every later one
        S.host[1] = N.done;
        if (S.host_iter) {
            S.host_iter[3 * (S.it + 1)] = alive_here;
            S.host_iter[3 * (S.it + 1) + 1] = N.done;
        }
        __threadfence_system();
        __hip_atomic_store(&S.host[2], S.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        if (S.host_iter) __hip_atomic_store(&S.host_iter[3 * (S.it + 1) + 2], S.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// The plan of iteration 0.  The launch also brings the call's lattice table (256 floats the host wrote behind the
// publish words of the pinned buffer, march_accel.hpp: build_lattice) into device memory.
__global__ void frame_init_kernel(ScheduleArgs S, float *__restrict__ lattice_dev, unsigned long long *stamps = nullptr,
                                  int n_stamps = 0)
{
    for (int i = threadIdx.x; i < n_stamps; i += blockDim.x) {      // {min start, max end} per iteration
        stamps[2 * i] = ~0ull;
        stamps[2 * i + 1] = 0ull;
    }
    if (lattice_dev) {
        const float *src = reinterpret_cast<const float *>(S.host + kHostLatticeWord);
        for (int i = threadIdx.x; i < 256; i += blockDim.x) lattice_dev[i] = src[i];
    }
    if (threadIdx.x < 64) make_next_plan(S);        // the first wave (launched with exactly one)
}

struct MarchArgs {
    const float *rays_o, *rays_d;
    GridSpec grid;                 // limit is per frame (plan)
    AccelSpec accel;
    float *near_planes;            // in: near plane, out: termination plane (cednerf/utils.py:301)
    float far_plane;
    const int32_t *alive;          // per-frame lists of the rays still alive (NULL: all rays, first iteration)
    int n_frames, rays_per_frame;
    const float *t_sorted;
    const uint8_t *t_indices;      // event ids as bytes (nerfacc's int64 layout would be 64 B per ray at four levels)
    const uint8_t *hits;
    float *t_starts, *t_ends;      // ray-packed samples of the iteration
    int32_t *ray_idx;              // ray of every sample
    int32_t *packed;               // [n_rays, 2] (first sample, count) of the ray in this iteration
    int32_t *cand;                 // first iteration in two passes: the rays march_cull_kernel could not rule out
};

constexpr int kMaxRuns = 4;        // runs of consecutive samples a ray's walk is remembered by (more: the ray walks twice)

// One lane per alive ray.  The walk does not store its samples: within a run of consecutive samples t_start[j+1] ==
// t_end[j] and t_end[j] = t_start[j] + dt(t_start[j]), so a ray remembers (first t, length) of up to kMaxRuns runs in
// registers.  Then the workgroup reserves ONE contiguous range of the iteration's sample array (wave prefix sums +
// a single returning atomic: the samples stay ray-packed and dense, with no count pass and no staging) and every
// lane regenerates its samples from its runs -- the same recurrence, the same floats.  A ray with more runs than fit
// (alternating single occupied cells), or a walk without a step size, simply walks again, storing directly.
// First iteration of a frame, pass one of two.  Nine rays in ten see nothing (the scene fills a few percent of the
// grid), and what decides that is only the sphere trace through the distance field: this pass runs the trace of every
// segment of every ray -- traverse_ray_frame's own first step, the same function on the same arguments -- on a lean
// kernel (no DDA state, twice the waves per SIMD of the marching kernel to hide the probes' latency).  A ray whose
// segments are all traced to their ends marches no sample (packed = {0, 0}); the others are compacted into a list
// that the marching kernel then walks with full waves.
constexpr int kCullThreads = 256;
static_assert(kSlotAlign % kCullThreads == 0, "a culling workgroup's slots belong to one frame");
template <bool SINGLE>
__global__ __launch_bounds__(kCullThreads) void march_cull_kernel(MarchArgs A, IterPlan *__restrict__ plan)
{
    constexpr int kWaves = kCullThreads / 64;
    __shared__ int wave_cnt[kWaves];
    __shared__ int block_base;
    const IterPlan &P = *plan;
    const int64_t total = P.total_slots;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = SINGLE ? 1 : A.grid.n_grids, res = A.grid.res;
    for (int64_t s0 = (int64_t)blockIdx.x * kCullThreads; s0 < total; s0 += (int64_t)gridDim.x * kCullThreads) {
        // kSlotAlign == kCullThreads: a workgroup's slots belong to one frame
        const int f = frame_of_slot(P, A.n_frames, s0);
        if (f < 0) continue;
        const int64_t idx = s0 + threadIdx.x - P.slot_base[f];
        const bool active = idx < P.count[f];
        const int64_t r = (int64_t)f * A.rays_per_frame + (active ? idx : 0);
        bool cand = false;
        if (active) {
            float o[3], d[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) { o[a] = A.rays_o[3 * r + a]; d[a] = A.rays_d[3 * r + a]; }
            const float near = A.near_planes[r];
            if constexpr (SINGLE) {
                const float inv_d[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
                float seg_a, seg_b;
                if (slab_test(o, inv_d, A.grid.aabbs, seg_a, seg_b)) {
                    const float t0 = fmaxf(seg_a, near), t1 = fminf(seg_b, A.far_plane);
                    float t_stop, cells;
                    if (t0 < t1) cand = !coarse_advance(A.accel, 0, res, A.grid.aabbs, o, d, t0, t1, t_stop, cells);
                }
            } else {
                // the segments traverse_ray_frame visits, in its order (nerfacc's sorted entry / exit events)
                const float *ts_row = A.t_sorted + r * 2 * m;
                const uint8_t *ti_row = A.t_indices + r * 2 * m;
                const uint8_t *hit_row = A.hits + r * m;
                for (int i = 0; i < 2 * m - 1 && !cand; ++i) {
                    const int ti = ti_row[i];
                    int lvl = (int)(ti % m);
                    if (!hit_row[lvl]) continue;
                    if (!(ti < m)) {
                        const int tn = ti_row[i + 1];
                        if (tn < m) continue;
                        lvl = (int)(tn % m);
                        if (!hit_row[lvl]) continue;
                    }
                    const float t0 = fmaxf(ts_row[i], near), t1 = fminf(ts_row[i + 1], A.far_plane);
                    if (t0 >= t1) continue;
                    float t_stop, cells;
                    cand = !coarse_advance(A.accel, lvl, res, A.grid.aabbs + 6 * lvl, o, d, t0, t1, t_stop, cells);
                }
            }
            if (!cand) { A.packed[2 * r] = 0; A.packed[2 * r + 1] = 0; }
        }
        const unsigned long long ballot = __ballot(cand);
        if (lane == 0) wave_cnt[wave] = __builtin_popcountll(ballot);
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int w = 0; w < kWaves; ++w) { const int t = wave_cnt[w]; wave_cnt[w] = run; run += t; }
            block_base = run > 0 ? atomicAdd(&plan->n_cand, run) : 0;
        }
        __syncthreads();
        if (cand) A.cand[block_base + wave_cnt[wave] + __builtin_popcountll(ballot & ((1ull << lane) - 1ull))] = (int32_t)r;
        __syncthreads();
    }
}

// FIRST: a frame's first iteration (every segment starts with a sphere trace; the walk in its looking-loop form,
// march_accel.hpp).  CAND: the rays are those of the culling pass's list, any frame's in any order, kCandLanes of them
// per wave (64; measured with 32 and 16 -- more, emptier waves in case a wave alone on its SIMD were bound by the latency
// of its own instruction stream: 118 and 164 us against 91, the kernel is bound by instruction issue, not by that).
#ifndef CED_CAND_LANES
#define CED_CAND_LANES 64
#endif
constexpr int kCandLanes = CED_CAND_LANES;
template <bool SINGLE, bool CAND, bool FIRST>
__global__ __launch_bounds__(kMarchThreads, (SINGLE && !CAND) ? 4 : 3) void march_frame_kernel(MarchArgs A, IterPlan *__restrict__ plan)
{
    static_assert(!CAND || FIRST, "the candidate list belongs to the first iteration");
    constexpr int kWaves = kMarchThreads / 64;
    constexpr int kPerGroup = CAND ? kWaves * kCandLanes : kMarchThreads;        // rays of one workgroup pass
    __shared__ int wave_tot[kWaves];
    __shared__ long long block_base;
    const IterPlan &P = *plan;
    const int64_t total = CAND ? (int64_t)P.n_cand : (int64_t)P.total_slots;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t s0 = (int64_t)blockIdx.x * kPerGroup; s0 < total; s0 += (int64_t)gridDim.x * kPerGroup) {
        int f = 0;
        bool active;
        int64_t r = 0;
        if constexpr (CAND) {
            const int64_t slot = s0 + wave * kCandLanes + lane;
            active = lane < kCandLanes && slot < total;
            if (active) { r = A.cand[slot]; f = (int)(r / A.rays_per_frame); }
        } else {
            f = frame_of_slot(P, A.n_frames, s0);
            if (f < 0) continue;                             // padding between two frames' slot ranges (whole workgroup)
            const int64_t idx = s0 + threadIdx.x - P.slot_base[f];
            active = idx < P.count[f];
            const int64_t first = (int64_t)f * A.rays_per_frame;
            r = active ? (A.alive ? (int64_t)A.alive[first + idx] : first + idx) : 0;
        }
        const int limit = P.limit[f];
        GridSpec grid = A.grid;
        grid.limit = limit;
        const int m = grid.n_grids;
        float o[3] = { 0.0f, 0.0f, 0.0f }, d[3] = { 0.0f, 0.0f, 1.0f };
        float near = 0.0f, t_term = 0.0f;
        float run_t[kMaxRuns];
        int run_n[kMaxRuns];
#pragma unroll
        for (int k = 0; k < kMaxRuns; ++k) { run_t[k] = 0.0f; run_n[k] = 0; }
        int n = 0, n_runs = 0;
        CED_DIAG_TICK(0);
        if (active) {
#pragma unroll
            for (int a = 0; a < 3; ++a) { o[a] = A.rays_o[3 * r + a]; d[a] = A.rays_d[3 * r + a]; }
            near = A.near_planes[r];
            float prev_end = 0.0f;
            n = traverse_ray_frame<kFrameLook, SINGLE, FIRST>(
                grid, A.accel, FIRST, o, d, near, A.far_plane,
                SINGLE ? nullptr : A.t_sorted + r * 2 * m, SINGLE ? (const uint8_t *)nullptr : A.t_indices + r * 2 * m,
                SINGLE ? nullptr : A.hits + r * m,
                [&](int i, float t0, float t1) {
                    const bool fresh = i == 0 || t0 != prev_end;
                    if (fresh) ++n_runs;
                    prev_end = t1;
                    // run slots as a chain of selects (run_t / run_n stay in registers)
#pragma unroll
                    for (int k = 0; k < kMaxRuns; ++k) {
                        const bool here = n_runs == k + 1;
                        run_t[k] = (here && fresh) ? t0 : run_t[k];
                        run_n[k] += here ? 1 : 0;
                    }
                },
                t_term);
            A.near_planes[r] = t_term;
        }
        // wave-inclusive prefix sum of the counts, one reservation per workgroup
        CED_DIAG_TICK(6);
        int incl = n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int w = 0; w < kWaves; ++w) { const int t = wave_tot[w]; wave_tot[w] = run; run += t; }
            block_base = run > 0 ? (long long)atomicAdd(reinterpret_cast<unsigned long long *>(&plan->total_samples),
                                                        (unsigned long long)run) : 0;
        }
        __syncthreads();
        const int64_t start = (int64_t)block_base + wave_tot[wave] + (incl - n);
        if (active) {
            A.packed[2 * r] = (int32_t)start;
            A.packed[2 * r + 1] = n;
            float *const p0 = A.t_starts + start, *const p1 = A.t_ends + start;
            int32_t *const pr = A.ray_idx + start;
            if (n > 0 && (n_runs > kMaxRuns || !(grid.step_size > 0.0f))) {
                float unused;                               // the walk again, storing at the final position
                (void)traverse_ray_frame<kFrameLook, SINGLE, FIRST>(
                    grid, A.accel, FIRST, o, d, near, A.far_plane,
                    SINGLE ? nullptr : A.t_sorted + r * 2 * m, SINGLE ? (const uint8_t *)nullptr : A.t_indices + r * 2 * m,
                    SINGLE ? nullptr : A.hits + r * m,
                    [&](int i, float t0, float t1) { p0[i] = t0; p1[i] = t1; pr[i] = (int32_t)r; }, unused);
            } else {
                int pos = 0;
#pragma unroll
                for (int k = 0; k < kMaxRuns; ++k) {
                    float t = run_t[k];
                    for (int j = 0; j < run_n[k]; ++j) {
                        const float t1 = t + calc_dt(t, grid.cone_angle, grid.step_size, 1e10f);
                        p0[pos] = t; p1[pos] = t1; pr[pos] = (int32_t)r;
                        ++pos;
                        t = t1;
                    }
                }
            }
        }
        CED_DIAG_TICK(7);
        __syncthreads();                                     // wave_tot / block_base are reused by the next chunk
    }
#ifdef CED_MARCH_DIAG
    ced_diag_flush();
#endif
}

#ifdef CED_MARCH_DIAG
}  // namespace ced
extern "C" int ced_diag_march_read(unsigned long long *out, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_diag), sizeof(unsigned long long) * 24) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[24] = { 0 };
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_march_diag), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
// rows of the per-wave log since the last reset (out: [max_rows][24] uint32); returns the number of rows
extern "C" int ced_diag_march_waves(unsigned int *out, int max_rows, int reset)
{
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    unsigned int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_march_wave_n), 4) != hipSuccess) return -1;
    if (n > (unsigned)kDiagWaves) n = kDiagWaves;
    if ((int)n > max_rows) n = max_rows;
    if (out && n && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_wave_log), (size_t)n * 96) != hipSuccess) return -1;
    if (reset) {
        unsigned int z = 0;
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_march_wave_n), &z, 4) != hipSuccess) return -1;
    }
    return (int)n;
}
namespace ced {
#endif

// composite_prefix (cednerf/utils.py:274-299) + ray bookkeeping (utils.py:301-307) over the list of alive rays;
// survivors (opacity <= threshold and a full sample budget) are appended to the next iteration's list, one range
// reservation per workgroup; the frame's sample count of the iteration rides in the high word of the same atomic.
__global__ __launch_bounds__(kCompositeThreads) void frame_composite_kernel(
    IterPlan *plan, int n_frames, int rays_per_frame, const int32_t *__restrict__ alive_list,
    int32_t *__restrict__ next_list, const int32_t *__restrict__ packed, const float *__restrict__ t0,
    const float *__restrict__ t1, const float *__restrict__ sig, const float *__restrict__ rgbs, float *__restrict__ rgb,
    float *__restrict__ opacity, float *__restrict__ depth, float opc_thres)
{
    constexpr int kWaves = kCompositeThreads / 64;
    __shared__ int wave_alive[kWaves], wave_samples[kWaves];
    __shared__ long long block_base;
    const IterPlan &P = *plan;
    const int64_t total = P.total_slots;
    for (int64_t s0 = (int64_t)blockIdx.x * kCompositeThreads; s0 < total; s0 += (int64_t)gridDim.x * kCompositeThreads) {
        const int f = frame_of_slot(P, n_frames, s0);
        if (f < 0) continue;                                 // padding between two frames' slot ranges (whole workgroup)
        const int limit = P.limit[f];
        const int64_t idx = s0 + threadIdx.x - P.slot_base[f];
        const bool active = idx < P.count[f];
        const int64_t first = (int64_t)f * rays_per_frame;
        const int64_t r = active ? (alive_list ? (int64_t)alive_list[first + idx] : first + idx) : 0;
        int cnt = 0;
        bool alive = false;
        if (active) {
            const int64_t sb = packed[2 * r];
            cnt = packed[2 * r + 1];
            float op = opacity[r];
            if (cnt > 0) {
                const float prefix = 1.0f - op;
                float c0 = rgb[3 * r], c1 = rgb[3 * r + 1], c2 = rgb[3 * r + 2], dp = depth[r];
                float acc = 0.0f;
                // Samples are consumed strictly in order (the per-ray sums are sequential by contract), but
                // their loads are issued kU at a time so one memory round trip feeds kU samples.
                constexpr int kU = CED_COMPOSITE_KU;
                int64_t i = sb;
                const int64_t end = sb + cnt;
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
            alive = !P.last[f] && (op <= opc_thres) && (cnt == limit);
        }
        const unsigned long long ballot = __ballot(alive);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int wsum = cnt;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) wsum += __shfl_xor(wsum, off, 64);
        if (lane == 0) { wave_alive[wave] = __builtin_popcountll(ballot); wave_samples[wave] = wsum; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0, samples = 0;
            for (int w = 0; w < kWaves; ++w) { int t = wave_alive[w]; wave_alive[w] = run; run += t; samples += wave_samples[w]; }
            const unsigned long long add = (unsigned long long)run + ((unsigned long long)samples << 32);
            block_base = add != 0 ? (long long)(atomicAdd(&plan->next[f], add) & 0xffffffffull) : 0;
        }
        __syncthreads();
        if (alive) {
            const int rank = __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
            next_list[first + block_base + wave_alive[wave] + rank] = (int32_t)r;
        }
        __syncthreads();                                     // wave_alive / block_base are reused by the next chunk
    }
}

// One wave turns the survivor counts of iteration `it` into the plan of iteration it + 1.  (Folding this into the
// compositing kernel's last workgroup was tried: the system-scope publish inside that kernel cost more than this
// launch -- 24 us instead of 14 + 5 per iteration.)
__global__ __launch_bounds__(64) void frame_schedule_kernel(ScheduleArgs S)
{
    make_next_plan(S);
}

// Sharded frames: this process's survivors of iteration `it`, per frame, as the int64 row the exchange step sums over
// the processes (ced_shard_exchange.reduce) before frame_schedule_kernel reads it.
__global__ __launch_bounds__(64) void frame_pack_counts_kernel(const IterPlan *__restrict__ plan, int n_frames,
                                                               int64_t *__restrict__ row)
{
    const int f = threadIdx.x;
    if (f >= n_frames) return;
    const unsigned long long nx = __hip_atomic_load(&plan->next[f], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool finished = plan->last[f] || plan->gcount[f] == 0 || plan->count[f] == 0;
    row[f] = finished ? 0 : (int64_t)(nx & 0xffffffffull);
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

// ------------------------------------------------------------------------------------------------------------------
// render_image in eval mode (cednerf/utils.py:46-150): `sampling` (march every ray to the far plane, evaluate the
// density of EVERY sample, keep those whose transmittance so far is >= early_stop_eps) followed by `rendering` (evaluate
// the field again on the survivors, weights, per-ray sums).  The kept samples of a ray are a prefix of its march (the
// transmittance only falls), so the same result comes from walking each ray's samples front to back in chunks, the field
// evaluated ONCE per sample (colour and density together), and stopping a ray at the first sample that fails the test:
// identical floats in the identical order => bit-identical outputs, with 3.9 M instead of 15 M + 3.9 M field
// evaluations on the 800x800 frame.  The chunk sizes follow the N_samples rule of the test loop (any would do).
// Per sample the call keeps (persistent arrays in the workspace, iteration after iteration, entry = sample_base + i):
// t0, t1, ray, sigma, rgb, weight, transmittance, alpha, and the sample's rank among the ray's kept ones (-1: dropped);
// ced_render_image_gather moves the kept ones into the reference's ray-major order.
struct ImageState {
    int32_t *cursor;     // samples of the ray consumed so far
    float *acc_all;      // sum of sigma * dt over ALL consumed samples      (visibility: render_visibility_from_density)
    float *acc_kept;     // sum of sigma * dt over the KEPT samples           (weights:    render_weight_from_density)
    int32_t *kept;       // kept samples of the ray so far
};

__global__ __launch_bounds__(256) void image_prep_kernel(int64_t n_rays, ImageState st, float *__restrict__ rgb,
                                                         float *__restrict__ opacity, float *__restrict__ depth)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    st.cursor[r] = 0; st.acc_all[r] = 0.0f; st.acc_kept[r] = 0.0f; st.kept[r] = 0;
    if (rgb) {
        rgb[3 * r] = 0.0f; rgb[3 * r + 1] = 0.0f; rgb[3 * r + 2] = 0.0f;
        opacity[r] = 0.0f;
        depth[r] = 0.0f;
    }
}

// The iteration's samples: the next `limit` samples of every alive ray, copied from the one-shot march into the
// iteration's window of the persistent arrays (ray-packed, one range reservation per workgroup as in march_frame_kernel).
// samples of ray r in the one-shot march; a range that passes the end of the arrays (a one-pass march that overflowed
// its capacity: the caller redoes the frame) counts as empty, so that nothing out of bounds is ever read
__device__ __forceinline__ int64_t ray_count(const int64_t *__restrict__ packed_all, int64_t r, int64_t n_all)
{
    const int64_t start = packed_all[2 * r], cnt = packed_all[2 * r + 1];
    return (start < 0 || cnt < 0 || start + cnt > n_all) ? 0 : cnt;
}

__global__ __launch_bounds__(kMarchThreads) void image_chunk_kernel(
    IterPlan *__restrict__ plan, int64_t n_all, const int32_t *__restrict__ alive, const int64_t *__restrict__ packed_all,
    const float *__restrict__ t0_all, const float *__restrict__ t1_all, const int32_t *__restrict__ cursor,
    float *__restrict__ t0s, float *__restrict__ t1s, int32_t *__restrict__ ridx, int32_t *__restrict__ packed)
{
    constexpr int kWaves = kMarchThreads / 64;
    __shared__ int wave_tot[kWaves];
    __shared__ long long block_base;
    const IterPlan &P = *plan;
    const int64_t total = P.count[0];
    const int limit = P.limit[0];
    const int64_t base = P.sample_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t s0 = (int64_t)blockIdx.x * kMarchThreads; s0 < total; s0 += (int64_t)gridDim.x * kMarchThreads) {
        const int64_t idx = s0 + threadIdx.x;
        const bool active = idx < total;
        const int64_t r = active ? (alive ? (int64_t)alive[idx] : idx) : 0;
        int n = 0;
        int64_t src = 0;
        if (active) {
            const int cur = cursor[r];
            const int64_t left = ray_count(packed_all, r, n_all) - cur;
            n = (int)(left < limit ? left : limit);
            src = packed_all[2 * r] + cur;
        }
        int incl = n;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int v = __shfl_up(incl, off, 64);
            if (lane >= off) incl += v;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0;
            for (int w = 0; w < kWaves; ++w) { const int t = wave_tot[w]; wave_tot[w] = run; run += t; }
            block_base = run > 0 ? (long long)atomicAdd(reinterpret_cast<unsigned long long *>(&plan->total_samples),
                                                        (unsigned long long)run) : 0;
        }
        __syncthreads();
        if (active) {
            const int64_t start = base + (int64_t)block_base + wave_tot[wave] + (incl - n);
            packed[2 * r] = (int32_t)start;
            packed[2 * r + 1] = n;
            for (int i = 0; i < n; ++i) {
                t0s[start + i] = t0_all[src + i];
                t1s[start + i] = t1_all[src + i];
                ridx[start + i] = (int32_t)r;
            }
        }
        __syncthreads();
    }
}

// Visibility (nerfacc render_visibility_from_density: exp(-sum so far) >= early_stop_eps, alpha >= alpha_thre), weights
// of the kept samples (render_weight_from_density over the kept ones only) and the three per-ray sums of
// cednerf/render.py:158-169 -- every sum sequential in sample order, carried across iterations in the ray's state.
template <bool FULL>
__global__ __launch_bounds__(kCompositeThreads) void image_composite_kernel(
    IterPlan *plan, const int32_t *__restrict__ alive_list, int32_t *__restrict__ next_list,
    const int32_t *__restrict__ packed, const int64_t *__restrict__ packed_all, const float *__restrict__ t0,
    const float *__restrict__ t1, const float *__restrict__ sig, const float *__restrict__ rgbs, ImageState st,
    float *__restrict__ w_out, float *__restrict__ tr_out, float *__restrict__ al_out, int32_t *__restrict__ rank_out,
    float *__restrict__ rgb, float *__restrict__ opacity, float *__restrict__ depth, float eps, float alpha_thre,
    int64_t n_all)
{
    constexpr int kWaves = kCompositeThreads / 64;
    __shared__ int wave_alive[kWaves], wave_samples[kWaves], wave_kept[kWaves];
    __shared__ long long block_base;
    const IterPlan &P = *plan;
    const int64_t total = P.count[0];
    for (int64_t s0 = (int64_t)blockIdx.x * kCompositeThreads; s0 < total; s0 += (int64_t)gridDim.x * kCompositeThreads) {
        const int64_t idx = s0 + threadIdx.x;
        const bool active = idx < total;
        const int64_t r = active ? (alive_list ? (int64_t)alive_list[idx] : idx) : 0;
        int cnt = 0, kept_new = 0;
        bool alive = false;
        if (active) {
            const int64_t sb = packed[2 * r];
            cnt = packed[2 * r + 1];
            if (cnt > 0) {
                float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, op = 0.0f, dp = 0.0f;
                if constexpr (FULL) { c0 = rgb[3 * r]; c1 = rgb[3 * r + 1]; c2 = rgb[3 * r + 2]; op = opacity[r]; dp = depth[r]; }
                float acc = st.acc_all[r], acck = st.acc_kept[r];
                int kept = st.kept[r];
                const int kept_before = kept;
                bool open = true;                   // false once a sample failed the transmittance test: the rest fail too
                constexpr int kU = 4;
                const int64_t end = sb + cnt;
                for (int64_t i = sb; i < end; i += kU) {
                    float ts[kU], te[kU], sg[kU], cr[kU], cg[kU], cb[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const int64_t j = i + u < end ? i + u : end - 1;
                        ts[u] = t0[j]; te[u] = t1[j]; sg[u] = sig[j];
                        if constexpr (FULL) { cr[u] = rgbs[3 * j]; cg[u] = rgbs[3 * j + 1]; cb[u] = rgbs[3 * j + 2]; }
                    }
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        if (i + u >= end) break;
                        const float sd = sg[u] * (te[u] - ts[u]);
                        const float a = 1.0f - det_expf(-sd);
                        const float tv = det_expf(-acc);
                        open = open && (tv >= eps);
                        const bool vis = open && !(alpha_thre > 0.0f && !(a >= alpha_thre));
                        acc = acc + sd;
                        int rank = -1;
                        if (vis) {
                            if constexpr (FULL) {
                                const float t = alpha_thre > 0.0f ? det_expf(-acck) : tv;   // no alpha test: the two sums coincide
                                const float w = t * a;
                                c0 = c0 + w * cr[u];
                                c1 = c1 + w * cg[u];
                                c2 = c2 + w * cb[u];
                                op = op + w;
                                dp = dp + w * ((ts[u] + te[u]) / 2.0f);
                                acck = acck + sd;
                                w_out[i + u] = w; tr_out[i + u] = t; al_out[i + u] = a;
                            }
                            rank = kept++;
                        }
                        rank_out[i + u] = rank;
                    }
                }
                if constexpr (FULL) {
                    rgb[3 * r] = c0; rgb[3 * r + 1] = c1; rgb[3 * r + 2] = c2;
                    opacity[r] = op;
                    depth[r] = dp;
                }
                const int cur = st.cursor[r] + cnt;
                st.cursor[r] = cur;
                st.acc_all[r] = acc; st.acc_kept[r] = acck; st.kept[r] = kept;
                kept_new = kept - kept_before;
                // alive: samples left, and the next one would pass the transmittance test
                alive = open && (int64_t)cur < ray_count(packed_all, r, n_all) && (det_expf(-acc) >= eps);
            }
        }
        const unsigned long long ballot = __ballot(alive);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        int wsum = cnt, ksum = kept_new;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { wsum += __shfl_xor(wsum, off, 64); ksum += __shfl_xor(ksum, off, 64); }
        if (lane == 0) { wave_alive[wave] = __builtin_popcountll(ballot); wave_samples[wave] = wsum; wave_kept[wave] = ksum; }
        __syncthreads();
        if (threadIdx.x == 0) {
            int run = 0, samples = 0, keptw = 0;
            for (int w = 0; w < kWaves; ++w) {
                int t = wave_alive[w]; wave_alive[w] = run; run += t; samples += wave_samples[w]; keptw += wave_kept[w];
            }
            const unsigned long long add = (unsigned long long)run + ((unsigned long long)samples << 32);
            block_base = add != 0 ? (long long)(atomicAdd(&plan->next[0], add) & 0xffffffffull) : 0;
            if (keptw) atomicAdd(&plan->next[1], (unsigned long long)keptw);       // kept samples of the iteration
        }
        __syncthreads();
        if (alive) {
            const int rank = __builtin_popcountll(ballot & ((1ull << lane) - 1ull));
            next_list[block_base + wave_alive[wave] + rank] = (int32_t)r;
        }
        __syncthreads();
    }
}

// Kept samples -> the reference's order (by ray, then along the ray): entry i of the persistent arrays goes to
// ray_offset[ray] + rank.  chunk_rays > 0: ray indices relative to the ray's chunk of that many rays (the `extras` of
// the reference's chunked eval loop, cednerf/utils.py:108-133).
struct ImageOut {
    int64_t *ray_indices;
    float *t_starts, *t_ends, *sigmas, *rgbs, *weights, *trans, *alphas;
};

__global__ __launch_bounds__(256) void image_gather_kernel(int64_t n, const int32_t *__restrict__ rank,
                                                           const int32_t *__restrict__ ridx, const float *__restrict__ t0,
                                                           const float *__restrict__ t1, const float *__restrict__ sig,
                                                           const float *__restrict__ rgbs, const float *__restrict__ w,
                                                           const float *__restrict__ tr, const float *__restrict__ al,
                                                           const int64_t *__restrict__ ray_offset, int64_t chunk_rays,
                                                           ImageOut out)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = rank[i];
        if (k < 0) continue;
        const int64_t r = ridx[i];
        const int64_t d = ray_offset[r] + k;
        out.ray_indices[d] = chunk_rays > 0 ? r % chunk_rays : r;
        out.t_starts[d] = t0[i]; out.t_ends[d] = t1[i];
        if (out.sigmas) out.sigmas[d] = sig[i];
        if (out.rgbs) {                       // (NULL after a sampling-only call: those arrays were never written)
            out.rgbs[3 * d] = rgbs[3 * i]; out.rgbs[3 * d + 1] = rgbs[3 * i + 1]; out.rgbs[3 * d + 2] = rgbs[3 * i + 2];
            out.weights[d] = w[i]; out.trans[d] = tr[i]; out.alphas[d] = al[i];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// One-shot march of every ray to the far plane (nerfacc traverse_grids without a step limit: the marching half of
// OccGridEstimator.sampling, call sites cednerf/utils.py:115-125 and train_real.py:339-350) on the frame renderer's
// accelerated walk: sphere tracing over the brick / cell distance fields through empty space, closed-form DDA re-entry,
// the emitted samples those of the cell-by-cell walk bit for bit (march_accel.hpp).  Count pass -> caller's scan ->
// fill pass, as ced_traverse_grids; one grid level (the multi-level case stays on ced_traverse_grids).
struct MarchAllArgs {
    int64_t n_rays;
    const float *rays_o, *rays_d;
    GridSpec grid;
    AccelSpec accel;
    const float *near_planes;
    float far_plane;
    const float *t_sorted;         // several grid levels: the rays' sorted entry / exit events (cednerf/utils.py:215-225)
    const int64_t *t_indices;
    const uint8_t *hits;
    int64_t *packed;               // [n_rays, 2]: FILL reads the first sample, COUNT writes the count
    float *t_starts, *t_ends;
    int64_t *ray_indices;          // optional
};

template <bool SINGLE, bool FILL>
__global__ __launch_bounds__(kMarchThreads, SINGLE ? 4 : 3) void march_all_kernel(MarchAllArgs A)
{
    const int64_t r = (int64_t)blockIdx.x * kMarchThreads + threadIdx.x;
    if (r >= A.n_rays) return;
    float o[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { o[a] = A.rays_o[3 * r + a]; d[a] = A.rays_d[3 * r + a]; }
    const int m = A.grid.n_grids;
    const float *const ts_row = SINGLE ? nullptr : A.t_sorted + r * 2 * m;
    const int64_t *const ti_row = SINGLE ? nullptr : A.t_indices + r * 2 * m;
    const uint8_t *const hit_row = SINGLE ? nullptr : A.hits + r * m;
    float t_term;
    if constexpr (!FILL) {
        const int n = traverse_ray_frame<kFrameLook, SINGLE, false>(A.grid, A.accel, true, o, d, A.near_planes[r], A.far_plane, ts_row,
                                                             ti_row, hit_row, [](int, float, float) {}, t_term);
        A.packed[2 * r + 1] = n;
    } else {
        const int64_t start = A.packed[2 * r];
        if (A.packed[2 * r + 1] == 0) return;
        float *const p0 = A.t_starts + start, *const p1 = A.t_ends + start;
        int64_t *const pr = A.ray_indices ? A.ray_indices + start : nullptr;
        (void)traverse_ray_frame<kFrameLook, SINGLE, false>(A.grid, A.accel, true, o, d, A.near_planes[r], A.far_plane, ts_row, ti_row,
                                                     hit_row,
                                                     [&](int i, float t0, float t1) {
                                                         p0[i] = t0; p1[i] = t1;
                                                         if (pr) pr[i] = r;
                                                     },
                                                     t_term);
    }
}

// The same march in ONE pass, for callers that can bound the total (a video's frames: the previous frame's count):
// every lane walks once remembering its samples as runs (as march_frame_kernel does), the workgroup reserves one
// range of the caller's arrays from a device counter and the samples are regenerated there.  Rays land in workgroup
// arrival order (packed_info says where); a ray whose range would pass `capacity` stores nothing -- the counter then
// exceeds the capacity and the caller falls back to the two-pass form.
template <bool SINGLE>
__global__ __launch_bounds__(kMarchThreads, SINGLE ? 4 : 3) void march_all_onepass_kernel(MarchAllArgs A, int64_t capacity,
                                                                                         unsigned long long *total)
{
    constexpr int kWaves = kMarchThreads / 64;
    __shared__ int wave_tot[kWaves];
    __shared__ long long block_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * kMarchThreads + threadIdx.x;
    const bool active = r < A.n_rays;
    const int m = A.grid.n_grids;
    float o[3] = { 0.0f, 0.0f, 0.0f }, d[3] = { 0.0f, 0.0f, 1.0f };
    float near = 0.0f, t_term = 0.0f;
    float run_t[kMaxRuns];
    int run_n[kMaxRuns];
#pragma unroll
    for (int k = 0; k < kMaxRuns; ++k) { run_t[k] = 0.0f; run_n[k] = 0; }
    int n = 0, n_runs = 0;
    const float *const ts_row = (SINGLE || !active) ? nullptr : A.t_sorted + r * 2 * m;
    const int64_t *const ti_row = (SINGLE || !active) ? nullptr : A.t_indices + r * 2 * m;
    const uint8_t *const hit_row = (SINGLE || !active) ? nullptr : A.hits + r * m;
    if (active) {
#pragma unroll
        for (int a = 0; a < 3; ++a) { o[a] = A.rays_o[3 * r + a]; d[a] = A.rays_d[3 * r + a]; }
        near = A.near_planes[r];
        float prev_end = 0.0f;
        n = traverse_ray_frame<kFrameLook, SINGLE, false>(
            A.grid, A.accel, true, o, d, near, A.far_plane, ts_row, ti_row, hit_row,
            [&](int i, float t0, float t1) {
                const bool fresh = i == 0 || t0 != prev_end;
                if (fresh) ++n_runs;
                prev_end = t1;
#pragma unroll
                for (int k = 0; k < kMaxRuns; ++k) {
                    const bool here = n_runs == k + 1;
                    run_t[k] = (here && fresh) ? t0 : run_t[k];
                    run_n[k] += here ? 1 : 0;
                }
            },
            t_term);
    }
    int incl = n;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int v = __shfl_up(incl, off, 64);
        if (lane >= off) incl += v;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < kWaves; ++w) { const int t = wave_tot[w]; wave_tot[w] = run; run += t; }
        block_base = run > 0 ? (long long)atomicAdd(total, (unsigned long long)run) : 0;
    }
    __syncthreads();
    if (!active) return;
    const int64_t start = (int64_t)block_base + wave_tot[wave] + (incl - n);
    A.packed[2 * r] = start;
    A.packed[2 * r + 1] = n;
    if (n == 0 || start + n > capacity) return;
    float *const p0 = A.t_starts + start, *const p1 = A.t_ends + start;
    if (n_runs > kMaxRuns || !(A.grid.step_size > 0.0f)) {
        float unused;
        (void)traverse_ray_frame<kFrameLook, SINGLE, false>(A.grid, A.accel, true, o, d, near, A.far_plane, ts_row, ti_row, hit_row,
                                                     [&](int i, float t0, float t1) { p0[i] = t0; p1[i] = t1; }, unused);
    } else {
        int pos = 0;
#pragma unroll
        for (int k = 0; k < kMaxRuns; ++k) {
            float t = run_t[k];
            for (int j = 0; j < run_n[k]; ++j) {
                const float t1 = t + calc_dt(t, A.grid.cone_angle, A.grid.step_size, 1e10f);
                p0[pos] = t; p1[pos] = t1;
                ++pos;
                t = t1;
            }
        }
    }
}

static inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct FrameWorkspace {
    float *t_sorted; uint8_t *t_indices; uint8_t *hits; float *near; int32_t *packed;
    int32_t *alive_a, *alive_b;     // double-buffered list of alive ray ids
    IterPlan *plans;                // [max_iters + 1]
    float *lattice;                 // [256] first lattice point per binade (cone_angle == 0)
    float *ts_ray;                  // per-ray time of a multi-frame call
    float *t0, *t1; int32_t *ridx; float *sigma, *rgbs;
    uint8_t *accel;                 // brick distance field when the caller did not bring one
    size_t bytes;
};

static inline int min_samples_of(float cone_angle) { return cone_angle == 0.0f ? 1 : 4; }
// iterations the reference loop can take: each one uses at least min_samples of the max_samples budget
static inline int max_iterations(int max_samples, int min_samples) { return (max_samples + min_samples - 1) / min_samples; }

// Samples one iteration can march, over all frames of a call.  A frame alone: alive * N_samples with N_samples =
// clamp(N // alive, min, 64), i.e. at most N * min_samples.  A SHARE of a frame whose loop is the whole image's
// (ced_shard_exchange): N_samples = N_image // alive_image, and this process's alive rays may be any part of
// alive_image -- at most min(local rays * 64, N_image) samples (or local rays * min_samples when that clamp binds).
static inline int64_t sample_capacity(int n_frames, int64_t rays_per_frame, int min_samples, int64_t global_rays)
{
    int64_t per = rays_per_frame * min_samples;
    if (global_rays > rays_per_frame) {
        const int64_t worst = rays_per_frame * 64 < global_rays ? rays_per_frame * 64 : global_rays;
        if (worst > per) per = worst;
    }
    return per * n_frames;
}

static FrameWorkspace carve(void *base, int64_t n, int m, int64_t cap, int max_iters, int res, bool per_ray_times = false)
{
    FrameWorkspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return (char *)base + o; };
    w.t_sorted = (float *)take((size_t)n * 2 * m * 4);
    w.t_indices = (uint8_t *)take((size_t)n * 2 * m);
    w.hits = (uint8_t *)take((size_t)n * m);
    w.near = (float *)take((size_t)n * 4);
    w.packed = (int32_t *)take((size_t)n * 8);
    w.alive_a = (int32_t *)take((size_t)n * 4);
    w.alive_b = (int32_t *)take((size_t)n * 4);
    w.plans = (IterPlan *)take((size_t)(max_iters + 2) * sizeof(IterPlan));
    w.lattice = (float *)take(256 * 4);
    w.ts_ray = (float *)take(per_ray_times ? (size_t)n * 4 : 0);
    w.t0 = (float *)take((size_t)cap * 4);
    w.t1 = (float *)take((size_t)cap * 4);
    w.ridx = (int32_t *)take((size_t)cap * 4);
    w.sigma = (float *)take((size_t)cap * 4);
    w.rgbs = (float *)take((size_t)cap * 12);
    const int64_t nbk = (res + kBrick - 1) / kBrick;
    w.accel = (uint8_t *)take((size_t)(2 * m * nbk * nbk * nbk));       // brick field + scratch (no accel from the caller)
    w.bytes = off;
    return w;
}

static std::atomic<long long> g_publish_seq{ 0 };

#if defined(__HIP_DEVICE_COMPILE__)
#define CED_CPU_PAUSE()
#else
#define CED_CPU_PAUSE() __builtin_ia32_pause()
#endif

// Waits until the sequence number in pinned host memory has reached `seq` (spin; after 200 ms a stream synchronise,
// after which the number must be there -- otherwise a launch of the chain has failed).
static int wait_published(volatile long long *flag, long long seq, hipStream_t stream, const char *who)
{
    const auto t_start = std::chrono::steady_clock::now();
    long spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < seq) {
        if ((++spins & 0x3ff) == 0) {
            if (std::chrono::steady_clock::now() - t_start > std::chrono::milliseconds(200)) {
                if (hipStreamSynchronize(stream) != hipSuccess) return check_launch("render_image_test (sync)");
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < seq) {
                    set_error("%s: iteration %lld was not published although the stream has drained", who, seq);
                    return CED_E_LAUNCH;
                }
                break;
            }
        } else {
            CED_CPU_PAUSE();
        }
    }
    return CED_OK;
}

// The frame loop for `n_frames` frames of `rays_per_frame` rays each (ced_render_image_test: one frame).
// frame_times == nullptr: `timestamps` is what the field kernel gets ([1] shared or [n_rays] per ray, t_per_ray);
// otherwise frame_times [n_frames] holds one time per frame and is expanded to a per-ray array.
static int render_frames_impl(const ced_field_desc *field, int n_frames, int64_t rays_per_frame, const float *rays_o,
                              const float *rays_d, const uint8_t *binaries, int32_t n_grids, int32_t res,
                              const float *aabbs, const void *accel, float near_plane, float far_plane, float step_size,
                              float cone_angle, float early_stop_eps, int32_t max_samples, const float *timestamps,
                              int32_t t_per_ray, const float *frame_times, const float *bkgd, float *rgb, float *opacity,
                              float *depth, void *workspace, int64_t workspace_bytes, int64_t *host_stats,
                              int64_t *total_samples_out, ced_frame_trace *trace, void *field_stream_, void *stream_,
                              const char *who, const ced_shard_exchange *xch = nullptr)
{
    hipStream_t stream = (hipStream_t)stream_;
    // Optional separate stream for the field kernel: callers that keep several frames in flight may hand every frame
    // the same field stream, so that the field launches of different frames queue behind each other.
    hipStream_t fstream = field_stream_ ? (hipStream_t)field_stream_ : stream;
    const bool split = fstream != stream;
    constexpr int kMaxDevices = 64;
    static thread_local hipEvent_t ev_to_field[kMaxDevices] = {}, ev_from_field[kMaxDevices] = {};
    int dev = 0;
    if (split) {                // events belong to the device they were created on
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return check_launch("render_image_test (device)");
        if (!ev_to_field[dev] &&
            (hipEventCreateWithFlags(&ev_to_field[dev], hipEventDisableTiming) != hipSuccess ||
             hipEventCreateWithFlags(&ev_from_field[dev], hipEventDisableTiming) != hipSuccess))
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
    const int64_t cap = sample_capacity(n_frames, rays_per_frame, min_samples, xch ? xch->global_rays_per_frame : 0);
    CED_REQUIRE(cap < (1ll << 31) - 64, "%s: too many samples per iteration for 32-bit sample indices", who);
    const int max_iters = max_iterations(max_samples, min_samples);
    FrameWorkspace W = carve(workspace, n_rays, n_grids, cap, max_iters, res, frame_times != nullptr);
    CED_REQUIRE((int64_t)W.bytes <= workspace_bytes, "%s: workspace too small (%lld < %lld bytes)", who,
                (long long)workspace_bytes, (long long)W.bytes);
    if (xch) {
        CED_REQUIRE(xch->reduce && xch->counts && xch->global_rays_per_frame >= rays_per_frame &&
                        xch->global_rays_per_frame < (1ll << 31),
                    "%s: bad ced_shard_exchange (reduce / counts NULL, or global_rays_per_frame < rays_per_frame)", who);
        CED_REQUIRE(!field_stream_ || field_stream_ == stream_, "%s: a sharded call takes no separate field stream", who);
    }
    long long *const host_iter = xch ? (long long *)host_stats + kHostIterWord : nullptr;
    const dim3 blk(256), grd((unsigned)((n_rays + 255) / 256));

    if (n_grids == 1)
        hipLaunchKernelGGL(frame_prep_single_kernel, grd, blk, 0, stream, n_rays, near_plane, W.near, rgb, opacity, depth);
    else
        hipLaunchKernelGGL(frame_prep_kernel, grd, blk, 0, stream, n_rays, rays_o, rays_d, (int)n_grids, aabbs, near_plane,
                           W.t_sorted, W.t_indices, W.hits, W.near, rgb, opacity, depth);
    if (frame_times)
        hipLaunchKernelGGL(frame_times_kernel, grd, blk, 0, stream, n_rays, (int)rays_per_frame, frame_times, W.ts_ray);
    const int nb = (res + kBrick - 1) / kBrick;
    AccelSpec acc{ nullptr, nb, nullptr };
    if (g_march_early_out) {
        if (accel) {
            acc = accel_view(accel, n_grids, res, true);             // the caller's: brick and cell fields
        } else {                                                     // none brought: the cheap brick field, per call
            int rc = build_brick_accel(binaries, n_grids, res, W.accel, W.accel + (size_t)n_grids * nb * nb * nb, stream);
            if (rc) return rc;
            acc = AccelSpec{ W.accel, nb, nullptr };
        }
    }
    volatile long long *pub = (volatile long long *)host_stats;
    std::vector<long long> seq_of((size_t)max_iters + 1);
    long long seq = ++g_publish_seq;
    const bool use_lattice = cone_angle == 0.0f && step_size > 0.0f && near_plane >= 0.0f;
    if (use_lattice) build_lattice(near_plane, step_size, reinterpret_cast<float *>(host_stats + kHostLatticeWord));
    hipLaunchKernelGGL(frame_init_kernel, dim3(1), dim3(64), 0, stream,
                       ScheduleArgs{ W.plans, -1, n_frames, (int)rays_per_frame, min_samples, (int)max_samples,
                                     (long long *)host_stats, seq, xch ? xch->local_rays : nullptr, nullptr,
                                     xch ? xch->global_rays_per_frame : 0, host_iter },
                       use_lattice ? W.lattice : (float *)nullptr,
                       (unsigned long long *)(trace ? trace->field_stamps : nullptr),
                       (trace && trace->field_stamps) ? (int)trace->capacity : 0);
    int rc = check_launch("render_image_test (prep)");
    if (rc) return rc;

    const float opc_thres = (float)(1.0 - (double)early_stop_eps);
    static const int run_ahead_env = [] {
        const char *e = getenv("CED_FRAME_RUN_AHEAD");
        const int v = e ? atoi(e) : 1;
        return v < 0 ? 0 : (v > 64 ? 64 : v);
    }();
    const int run_ahead = run_ahead_env;
    const long long seq_plan0 = seq;              // publication of plan[0]
    long long alive_bound = n_rays;               // rays alive in the iteration being enqueued: never more than this
    int it = 0;
    for (; it < max_iters; ++it) {
        // run-ahead control: the plan of iteration it - run_ahead must have been published (it is published by the
        // schedule launch of iteration it - run_ahead - 1, or by the initial one)
        const int need = it - run_ahead;
        if (xch) {
            // Sharded frames: every process must enqueue the same number of iterations (each holds a collective), so
            // the loop ends on the plan of iteration `need` -- identical on all processes -- and on nothing later that
            // happens to have been published already.
            if (need >= 0) {
                volatile long long *rec = host_iter + 3 * need;
                rc = wait_published(rec + 2, need == 0 ? seq_plan0 : seq_of[need - 1], stream, who);
                if (rc) return rc;
                if (rec[1]) break;
                const long long a = rec[0];
                if (a >= 0 && a < alive_bound) alive_bound = a;
            }
        } else {
            if (need >= 0) {
                rc = wait_published(pub + 2, need == 0 ? seq_plan0 : seq_of[need - 1], stream, who);
                if (rc) return rc;
            }
            if (__atomic_load_n(pub + 2, __ATOMIC_ACQUIRE) >= seq_plan0) {       // something of THIS call has been published
                if (pub[1]) break;                                              // nothing left: stop enqueueing
                const long long a = pub[0];
                if (a >= 0 && a < alive_bound) alive_bound = a;
            }
        }
        IterPlan *plan = W.plans + it;
        const int32_t *cur_list = it == 0 ? nullptr : ((it & 1) ? W.alive_a : W.alive_b);
        int32_t *next_list = (it & 1) ? W.alive_b : W.alive_a;

        // slots <= alive rays + padding between frames; the kernels stride over what the plan really holds
        const int64_t slot_bound = alive_bound + (int64_t)n_frames * kSlotAlign;
        MarchArgs M{ rays_o, rays_d, GridSpec{ binaries, aabbs, n_grids, res, step_size, cone_angle, 0, use_lattice ? W.lattice : nullptr },
                     acc, W.near, far_plane, cur_list, n_frames,
                     (int)rays_per_frame, W.t_sorted, W.t_indices, W.hits, W.t0, W.t1, W.ridx, W.packed, W.alive_b };
        int64_t mgrid = (slot_bound + kMarchThreads - 1) / kMarchThreads;
        if (mgrid > 8192) mgrid = 8192;
        const dim3 mblk(kMarchThreads);
        // Two passes pay where the marching kernel is heavy (several grid levels: sorted event lists, 160 registers,
        // three waves per SIMD): C4 +3 % pipelined, first iteration 370 -> 335 us; one level: 103 us in one pass
        // against 29 + 91 in two.
        const int two_pass_opt = g_march_two_pass;
        const bool two_pass = two_pass_opt < 0 ? n_grids > 1 : two_pass_opt != 0;
        if (it == 0 && acc.bdist && two_pass) {
            // first iteration in two passes: cull by the sphere trace alone, then march the rays that are left
            int64_t kgrid = (slot_bound + kCullThreads - 1) / kCullThreads;
            if (kgrid > 8192) kgrid = 8192;
            if (mgrid > 4096) mgrid = 4096;                  // the list's length is only known on the device
            if (n_grids == 1) {
                hipLaunchKernelGGL(march_cull_kernel<true>, dim3((unsigned)kgrid), dim3(kCullThreads), 0, stream, M, plan);
                hipLaunchKernelGGL((march_frame_kernel<true, true, true>), dim3((unsigned)mgrid), mblk, 0, stream, M, plan);
            } else {
                hipLaunchKernelGGL(march_cull_kernel<false>, dim3((unsigned)kgrid), dim3(kCullThreads), 0, stream, M, plan);
                hipLaunchKernelGGL((march_frame_kernel<false, true, true>), dim3((unsigned)mgrid), mblk, 0, stream, M, plan);
            }
        } else if (it == 0) {
            if (n_grids == 1) hipLaunchKernelGGL((march_frame_kernel<true, false, true>), dim3((unsigned)mgrid), mblk, 0, stream, M, plan);
            else hipLaunchKernelGGL((march_frame_kernel<false, false, true>), dim3((unsigned)mgrid), mblk, 0, stream, M, plan);
        } else {
            if (n_grids == 1) hipLaunchKernelGGL((march_frame_kernel<true, false, false>), dim3((unsigned)mgrid), mblk, 0, stream, M, plan);
            else hipLaunchKernelGGL((march_frame_kernel<false, false, false>), dim3((unsigned)mgrid), mblk, 0, stream, M, plan);
        }
        rc = check_launch("render_image_test (march)");
        if (rc) return rc;

        FieldArgs F{};
        F.n = cap;                          // host-side upper bound; the kernel reads the exact count from the plan
        F.n_dev = &plan->total_samples;
        F.rays_o = rays_o; F.rays_d = rays_d; F.ray_idx32 = W.ridx;
        F.t0 = W.t0; F.t1 = W.t1;
        F.timestamps = frame_times ? W.ts_ray : timestamps;
        F.rays_mode = 1; F.t_per_ray = (frame_times || t_per_ray) ? 1 : 0; F.want_rgb = 1;
        F.rgb = W.rgbs; F.sigma = W.sigma; F.geo = nullptr;
        if (trace && trace->field_stamps && it < trace->capacity)
            F.stamp = reinterpret_cast<unsigned long long *>(trace->field_stamps) + 2 * it;
        if (split) {
            (void)hipEventRecord(ev_to_field[dev], stream);
            (void)hipStreamWaitEvent(fstream, ev_to_field[dev], 0);
        }
        if (trace && it < trace->capacity && trace->field_begin)
            (void)hipEventRecord((hipEvent_t)trace->field_begin[it], fstream);
        rc = launch_field(field, F, (void *)fstream);
        if (rc) return rc;
        if (trace && it < trace->capacity && trace->field_end)
            (void)hipEventRecord((hipEvent_t)trace->field_end[it], fstream);
        if (split) {
            (void)hipEventRecord(ev_from_field[dev], fstream);
            (void)hipStreamWaitEvent(stream, ev_from_field[dev], 0);
        }

        int64_t cgrid = (slot_bound + kCompositeThreads - 1) / kCompositeThreads;
        if (cgrid > 4096) cgrid = 4096;
        seq = ++g_publish_seq;
        seq_of[it] = seq;
        hipLaunchKernelGGL(frame_composite_kernel, dim3((unsigned)cgrid), dim3(kCompositeThreads), 0, stream,
                           plan, n_frames, (int)rays_per_frame, cur_list, next_list, W.packed, W.t0, W.t1, W.sigma, W.rgbs,
                           rgb, opacity, depth, opc_thres);
        int64_t *xrow = nullptr;
        if (xch) {
            // the exchange step: this process's survivors per frame -> summed over the processes, on the stream
            xrow = xch->counts + (int64_t)it * n_frames;
            hipLaunchKernelGGL(frame_pack_counts_kernel, dim3(1), dim3(64), 0, stream, plan, n_frames, xrow);
            rc = check_launch("render_image_test (pack counts)");
            if (rc) return rc;
            const int xrc = xch->reduce(xch->user, xrow, n_frames, it, (void *)stream);
            if (xrc != 0) {
                set_error("%s: the exchange step of iteration %d failed (code %d)", who, it, xrc);
                (void)hipStreamSynchronize(stream);
                return CED_E_LAUNCH;
            }
        }
        hipLaunchKernelGGL(frame_schedule_kernel, dim3(1), dim3(64), 0, stream,
                           ScheduleArgs{ W.plans, it, n_frames, (int)rays_per_frame, min_samples, (int)max_samples,
                                         (long long *)host_stats, seq, xch ? xch->local_rays : nullptr,
                                         xch ? xch->counts : nullptr, xch ? xch->global_rays_per_frame : 0, host_iter });
        rc = check_launch("render_image_test (composite / schedule)");
        if (rc) return rc;
    }
    const int enqueued = it;
    hipLaunchKernelGGL(frame_finalize_kernel, grd, blk, 0, stream, n_rays, bkgd, rgb, opacity, depth);
    rc = check_launch("render_image_test (finalize)");
    if (rc) return rc;
    // the per-iteration record (plans) comes back in one copy; the call blocks here, once, for the sample totals
    std::vector<IterPlan> plans((size_t)enqueued + 1);
    if (hipMemcpyAsync(plans.data(), W.plans, plans.size() * sizeof(IterPlan), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return check_launch("render_image_test (read-back)");
    int64_t total[kMaxFrames] = { 0 };
    int n_iters = 0;
    for (int k = 0; k < enqueued; ++k) {
        const IterPlan &P = plans[k];
        if (P.done || P.total_slots == 0) break;
        int64_t alive_total = 0, samples_it = 0;
        int max_limit = 0;
        for (int f = 0; f < n_frames; ++f) {
            alive_total += P.count[f];
            samples_it += (int64_t)(P.next[f] >> 32);
            total[f] += (int64_t)(P.next[f] >> 32);
            if (P.count[f] > 0 && P.limit[f] > max_limit) max_limit = P.limit[f];
        }
        if (trace && k < trace->capacity) {
            if (trace->iter_alive) trace->iter_alive[k] = alive_total;
            if (trace->iter_n_samples) trace->iter_n_samples[k] = max_limit;
            if (trace->iter_samples) trace->iter_samples[k] = samples_it;
        }
        ++n_iters;
    }
    if (n_iters == enqueued && enqueued < max_iters && !plans[enqueued].done) {
        set_error("%s: the frame loop stopped after %d iterations with rays still alive", who, enqueued);
        return CED_E_LAUNCH;
    }
    if (trace) trace->n_iters = n_iters;
    if (total_samples_out)
        for (int f = 0; f < n_frames; ++f) total_samples_out[f] = total[f];
    return CED_OK;
}

struct ImageWorkspace {
    ImageState st;
    int32_t *packed, *alive_a, *alive_b;
    IterPlan *plans;
    float *t0, *t1; int32_t *ridx; float *sigma, *rgbs, *w, *tr, *al; int32_t *rank;
    size_t bytes;
};

constexpr int kImageMaxIters = 1024;

static ImageWorkspace carve_image(void *base, int64_t n, int64_t cap)
{
    ImageWorkspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes); return (char *)base + o; };
    w.st.cursor = (int32_t *)take((size_t)n * 4);
    w.st.acc_all = (float *)take((size_t)n * 4);
    w.st.acc_kept = (float *)take((size_t)n * 4);
    w.packed = (int32_t *)take((size_t)n * 8);
    w.alive_a = (int32_t *)take((size_t)n * 4);
    w.alive_b = (int32_t *)take((size_t)n * 4);
    w.plans = (IterPlan *)take((size_t)(kImageMaxIters + 2) * sizeof(IterPlan));
    w.t0 = (float *)take((size_t)cap * 4);
    w.t1 = (float *)take((size_t)cap * 4);
    w.ridx = (int32_t *)take((size_t)cap * 4);
    w.sigma = (float *)take((size_t)cap * 4);
    w.rgbs = (float *)take((size_t)cap * 12);
    w.w = (float *)take((size_t)cap * 4);
    w.tr = (float *)take((size_t)cap * 4);
    w.al = (float *)take((size_t)cap * 4);
    w.rank = (int32_t *)take((size_t)cap * 4);
    w.bytes = off;
    return w;
}

// a ray consumes at most its whole march; an iteration at most n_rays samples (N_samples <= n_rays // alive) plus the
// field kernel's tile padding
static inline int64_t image_capacity(int64_t n_rays, int64_t n_all) { return n_all + 64; }

static int render_image_impl(const ced_field_desc *field, int64_t n_rays, const float *rays_o, const float *rays_d,
                             int64_t n_all, const int64_t *packed_all, const float *t0_all, const float *t1_all,
                             float early_stop_eps, float alpha_thre, const float *timestamps, int32_t t_per_ray,
                             const float *bkgd, float *rgb, float *opacity, float *depth, int32_t *kept, void *workspace,
                             int64_t workspace_bytes, int64_t *host_stats, int64_t *stats_out, void *field_stream_,
                             void *stream_)
{
    const char *who = "render_image";
    hipStream_t stream = (hipStream_t)stream_;
    hipStream_t fstream = field_stream_ ? (hipStream_t)field_stream_ : stream;
    CED_REQUIRE(fstream == stream, "%s: a separate field stream is not supported", who);
    CED_REQUIRE(field != nullptr, "%s: null field descriptor", who);
    CED_REQUIRE(n_rays >= 0 && n_all >= 0, "%s: bad sizes", who);
    CED_REQUIRE(n_rays < (1ll << 31) / 4 && n_all < (1ll << 31) - 128, "%s: too many rays / samples for 32-bit indices", who);
    if (stats_out) { stats_out[0] = 0; stats_out[1] = 0; stats_out[2] = 0; }
    if (n_rays == 0) return CED_OK;
    const bool full = rgb != nullptr;               // rgb == NULL: sampling only (density-only field, no pixel sums)
    CED_REQUIRE(rays_o && rays_d && packed_all && (n_all == 0 || (t0_all && t1_all)) && timestamps && kept && workspace &&
                    host_stats && (!full || (opacity && depth)),
                "%s: null pointer", who);
    const int64_t cap = image_capacity(n_rays, n_all);
    ImageWorkspace W = carve_image(workspace, n_rays, cap);
    W.st.kept = kept;
    CED_REQUIRE((int64_t)W.bytes <= workspace_bytes, "%s: workspace too small (%lld < %lld bytes)", who,
                (long long)workspace_bytes, (long long)W.bytes);
    const dim3 blk(256), grd((unsigned)((n_rays + 255) / 256));
    hipLaunchKernelGGL(image_prep_kernel, grd, blk, 0, stream, n_rays, W.st, rgb, opacity, depth);
    volatile long long *pub = (volatile long long *)host_stats;
    std::vector<long long> seq_of((size_t)kImageMaxIters + 1);
    long long seq = ++g_publish_seq;
    const int big = 1 << 30;                               // no sample budget: the loop ends when no ray is alive
    // smallest chunk of an iteration (any chunking gives the same result; fewer, larger iterations against samples
    // evaluated behind a ray's end): measured on the 800x800 frame / the 262 k-ray batch, 4 for the full pass (4.26 ->
    // 4.03 ms against chunks from 1) and 8 for the density-only sampling pass
    static const int min_chunk_env = [] { const char *e = getenv("CED_IMAGE_MIN_CHUNK"); return e ? atoi(e) : 0; }();
    const int min_chunk = min_chunk_env > 0 ? (min_chunk_env > 64 ? 64 : min_chunk_env) : (full ? 4 : 8);
    hipLaunchKernelGGL(frame_init_kernel, dim3(1), dim3(64), 0, stream,
                       ScheduleArgs{ W.plans, -1, 1, (int)n_rays, min_chunk, big, (long long *)host_stats, seq }, (float *)nullptr,
                       (unsigned long long *)nullptr, 0);
    int rc = check_launch("render_image (prep)");
    if (rc) return rc;
    static const int run_ahead_env = [] {
        const char *e = getenv("CED_FRAME_RUN_AHEAD");
        const int v = e ? atoi(e) : 1;
        return v < 0 ? 0 : (v > 64 ? 64 : v);
    }();
    const int run_ahead = run_ahead_env;
    const long long seq_plan0 = seq;
    long long alive_bound = n_rays;
    int it = 0;
    for (; it < kImageMaxIters; ++it) {
        const int need = it - run_ahead;
        if (need >= 0) {
            rc = wait_published(pub + 2, need == 0 ? seq_plan0 : seq_of[need - 1], stream, who);
            if (rc) return rc;
        }
        if (__atomic_load_n(pub + 2, __ATOMIC_ACQUIRE) >= seq_plan0) {
            if (pub[1]) break;
            const long long a = pub[0];
            if (a >= 0 && a < alive_bound) alive_bound = a;
        }
        IterPlan *plan = W.plans + it;
        const int32_t *cur_list = it == 0 ? nullptr : ((it & 1) ? W.alive_a : W.alive_b);
        int32_t *next_list = (it & 1) ? W.alive_b : W.alive_a;
        int64_t mgrid = (alive_bound + kMarchThreads - 1) / kMarchThreads;
        if (mgrid > 8192) mgrid = 8192;
        if (mgrid < 1) mgrid = 1;
        hipLaunchKernelGGL(image_chunk_kernel, dim3((unsigned)mgrid), dim3(kMarchThreads), 0, stream, plan, n_all, cur_list,
                           packed_all, t0_all, t1_all, W.st.cursor, W.t0, W.t1, W.ridx, W.packed);
        FieldArgs F{};
        F.n = n_rays * (int64_t)min_chunk;  // an iteration's samples: alive * N_samples <= n_rays * min_chunk; exact count from the plan
        F.n_dev = &plan->total_samples;
        F.base_dev = &plan->sample_base;
        F.rays_o = rays_o; F.rays_d = rays_d; F.ray_idx32 = W.ridx;
        F.t0 = W.t0; F.t1 = W.t1;
        F.timestamps = timestamps;
        F.rays_mode = 1; F.t_per_ray = t_per_ray ? 1 : 0; F.want_rgb = full ? 1 : 0;
        F.rgb = full ? W.rgbs : nullptr; F.sigma = W.sigma; F.geo = nullptr;
        rc = launch_field(field, F, (void *)stream);
        if (rc) return rc;
        int64_t cgrid = (alive_bound + kCompositeThreads - 1) / kCompositeThreads;
        if (cgrid > 4096) cgrid = 4096;
        if (cgrid < 1) cgrid = 1;
        seq = ++g_publish_seq;
        seq_of[it] = seq;
        if (full)
            hipLaunchKernelGGL(image_composite_kernel<true>, dim3((unsigned)cgrid), dim3(kCompositeThreads), 0, stream, plan,
                               cur_list, next_list, W.packed, packed_all, W.t0, W.t1, W.sigma, W.rgbs, W.st, W.w, W.tr, W.al,
                               W.rank, rgb, opacity, depth, early_stop_eps, alpha_thre, n_all);
        else
            hipLaunchKernelGGL(image_composite_kernel<false>, dim3((unsigned)cgrid), dim3(kCompositeThreads), 0, stream, plan,
                               cur_list, next_list, W.packed, packed_all, W.t0, W.t1, W.sigma, (const float *)nullptr, W.st,
                               (float *)nullptr, (float *)nullptr, (float *)nullptr, W.rank, (float *)nullptr,
                               (float *)nullptr, (float *)nullptr, early_stop_eps, alpha_thre, n_all);
        hipLaunchKernelGGL(frame_schedule_kernel, dim3(1), dim3(64), 0, stream,
                           ScheduleArgs{ W.plans, it, 1, (int)n_rays, min_chunk, big, (long long *)host_stats, seq });
        rc = check_launch("render_image (iteration)");
        if (rc) return rc;
    }
    const int enqueued = it;
    if (full) hipLaunchKernelGGL(frame_finalize_kernel, grd, blk, 0, stream, n_rays, bkgd, rgb, opacity, depth);
    rc = check_launch("render_image (finalize)");
    if (rc) return rc;
    std::vector<IterPlan> plans((size_t)enqueued + 1);
    if (hipMemcpyAsync(plans.data(), W.plans, plans.size() * sizeof(IterPlan), hipMemcpyDeviceToHost, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess)
        return check_launch("render_image (read-back)");
    int64_t processed = 0, kept_total = 0;
    int n_iters = 0;
    for (int k = 0; k < enqueued; ++k) {
        const IterPlan &P = plans[k];
        if (P.done || P.count[0] == 0) break;
        processed = P.sample_base + P.total_samples;
        kept_total += (int64_t)P.next[1];
        ++n_iters;
    }
    if (n_iters == enqueued && enqueued == kImageMaxIters && !plans[enqueued].done) {
        set_error("%s: rays still alive after %d iterations", who, enqueued);
        return CED_E_LAUNCH;
    }
    CED_REQUIRE(processed <= cap, "%s: internal: %lld samples processed, capacity %lld", who, (long long)processed, (long long)cap);
    if (stats_out) { stats_out[0] = processed; stats_out[1] = n_iters; stats_out[2] = kept_total; }
    return CED_OK;
}

}  // namespace ced

extern "C" int64_t ced_render_image_test_workspace_bytes(int64_t n_rays, int32_t n_grids, int32_t res,
                                                         float cone_angle, int32_t max_samples)
{
    if (n_rays < 0 || n_grids < 1 || n_grids > ced::kMaxGrids || res < 1 || res > 1024 || max_samples < 0) return -1;
    const int ms = ced::min_samples_of(cone_angle);
    return (int64_t)ced::carve(nullptr, n_rays, n_grids, n_rays * ms, ced::max_iterations(max_samples, ms), res).bytes;
}

extern "C" int ced_render_image_test(const ced_field_desc *field, int64_t n_rays, const float *rays_o,
                                     const float *rays_d, const uint8_t *binaries, int32_t n_grids, int32_t res,
                                     const float *aabbs, const void *accel, float near_plane, float far_plane,
                                     float step_size, float cone_angle, float early_stop_eps, int32_t max_samples,
                                     const float *timestamps, int32_t t_per_ray, const float *bkgd, float *rgb,
                                     float *opacity, float *depth, void *workspace, int64_t workspace_bytes,
                                     int64_t *host_stats, int64_t *total_samples_out, ced_frame_trace *trace,
                                     void *field_stream_, void *stream_)
{
    return ced::render_frames_impl(field, 1, n_rays, rays_o, rays_d, binaries, n_grids, res, aabbs, accel, near_plane,
                                   far_plane, step_size, cone_angle, early_stop_eps, max_samples, timestamps, t_per_ray,
                                   nullptr, bkgd, rgb, opacity, depth, workspace, workspace_bytes, host_stats,
                                   total_samples_out, trace, field_stream_, stream_, "render_image_test");
}

extern "C" int64_t ced_render_frames_test_workspace_bytes(int32_t n_frames, int64_t rays_per_frame, int32_t n_grids,
                                                          int32_t res, float cone_angle, int32_t max_samples)
{
    if (n_frames < 1 || n_frames > ced::kMaxFrames || rays_per_frame < 0 || n_grids < 1 || n_grids > ced::kMaxGrids ||
        res < 1 || res > 1024 || max_samples < 0)
        return -1;
    const int64_t n_rays = (int64_t)n_frames * rays_per_frame;
    const int ms = ced::min_samples_of(cone_angle);
    return (int64_t)ced::carve(nullptr, n_rays, n_grids, n_rays * ms, ced::max_iterations(max_samples, ms), res, true).bytes;
}

extern "C" int ced_render_frames_test(const ced_field_desc *field, int32_t n_frames, int64_t rays_per_frame,
                                      const float *rays_o, const float *rays_d, const uint8_t *binaries, int32_t n_grids,
                                      int32_t res, const float *aabbs, const void *accel, float near_plane,
                                      float far_plane, float step_size, float cone_angle, float early_stop_eps,
                                      int32_t max_samples, const float *frame_times, const float *bkgd, float *rgb,
                                      float *opacity, float *depth, void *workspace, int64_t workspace_bytes,
                                      int64_t *host_stats, int64_t *total_samples_out, ced_frame_trace *trace,
                                      void *field_stream_, void *stream_)
{
    CED_REQUIRE(frame_times != nullptr, "render_frames_test: null frame_times");
    return ced::render_frames_impl(field, n_frames, rays_per_frame, rays_o, rays_d, binaries, n_grids, res, aabbs, accel,
                                   near_plane, far_plane, step_size, cone_angle, early_stop_eps, max_samples, nullptr, 0,
                                   frame_times, bkgd, rgb, opacity, depth, workspace, workspace_bytes, host_stats,
                                   total_samples_out, trace, field_stream_, stream_, "render_frames_test");
}


extern "C" int64_t ced_render_frames_test_sharded_workspace_bytes(int32_t n_frames, int64_t rays_per_frame,
                                                                  int64_t global_rays_per_frame, int32_t n_grids,
                                                                  int32_t res, float cone_angle, int32_t max_samples)
{
    if (n_frames < 1 || n_frames > ced::kMaxFrames || rays_per_frame < 0 || global_rays_per_frame < rays_per_frame ||
        n_grids < 1 || n_grids > ced::kMaxGrids || res < 1 || res > 1024 || max_samples < 0)
        return -1;
    const int64_t n_rays = (int64_t)n_frames * rays_per_frame;
    const int ms = ced::min_samples_of(cone_angle);
    return (int64_t)ced::carve(nullptr, n_rays, n_grids, ced::sample_capacity(n_frames, rays_per_frame, ms, global_rays_per_frame),
                               ced::max_iterations(max_samples, ms), res, true).bytes;
}

extern "C" int32_t ced_render_frames_test_iterations(float cone_angle, int32_t max_samples)
{
    if (max_samples < 0) return -1;
    return ced::max_iterations(max_samples, ced::min_samples_of(cone_angle));
}

extern "C" int64_t ced_render_frames_test_host_bytes(float cone_angle, int32_t max_samples)
{
    if (max_samples < 0) return -1;
    const int64_t iters = ced::max_iterations(max_samples, ced::min_samples_of(cone_angle));
    return (ced::kHostIterWord + 3 * (iters + 2)) * (int64_t)sizeof(int64_t);
}

extern "C" int ced_render_frames_test_sharded(const ced_field_desc *field, int32_t n_frames, int64_t rays_per_frame,
                                              const float *rays_o, const float *rays_d, const uint8_t *binaries,
                                              int32_t n_grids, int32_t res, const float *aabbs, const void *accel,
                                              float near_plane, float far_plane, float step_size, float cone_angle,
                                              float early_stop_eps, int32_t max_samples, const float *frame_times,
                                              const float *bkgd, float *rgb, float *opacity, float *depth, void *workspace,
                                              int64_t workspace_bytes, int64_t *host_stats, int64_t *total_samples_out,
                                              ced_frame_trace *trace, void *field_stream_, void *stream_,
                                              const ced_shard_exchange *exchange)
{
    CED_REQUIRE(frame_times != nullptr, "render_frames_test_sharded: null frame_times");
    return ced::render_frames_impl(field, n_frames, rays_per_frame, rays_o, rays_d, binaries, n_grids, res, aabbs, accel,
                                   near_plane, far_plane, step_size, cone_angle, early_stop_eps, max_samples, nullptr, 0,
                                   frame_times, bkgd, rgb, opacity, depth, workspace, workspace_bytes, host_stats,
                                   total_samples_out, trace, field_stream_, stream_, "render_frames_test_sharded", exchange);
}

extern "C" int64_t ced_render_image_workspace_bytes(int64_t n_rays, int64_t n_all)
{
    if (n_rays < 0 || n_all < 0) return -1;
    return (int64_t)ced::carve_image(nullptr, n_rays, ced::image_capacity(n_rays, n_all)).bytes;
}

extern "C" int ced_render_image(const ced_field_desc *field, int64_t n_rays, const float *rays_o, const float *rays_d,
                                int64_t n_all, const int64_t *packed_info, const float *t_starts, const float *t_ends,
                                float early_stop_eps, float alpha_thre, const float *timestamps, int32_t t_per_ray,
                                const float *bkgd, float *rgb, float *opacity, float *depth, int32_t *kept,
                                void *workspace, int64_t workspace_bytes, int64_t *host_stats, int64_t *stats_out,
                                void *field_stream, void *stream)
{
    return ced::render_image_impl(field, n_rays, rays_o, rays_d, n_all, packed_info, t_starts, t_ends, early_stop_eps,
                                  alpha_thre, timestamps, t_per_ray, bkgd, rgb, opacity, depth, kept, workspace,
                                  workspace_bytes, host_stats, stats_out, field_stream, stream);
}

extern "C" int ced_render_image_gather(int64_t n_rays, int64_t n_all, int64_t processed, const void *workspace,
                                       int64_t workspace_bytes, const int64_t *ray_offsets, int64_t chunk_rays,
                                       int64_t *ray_indices, float *t_starts, float *t_ends, float *sigmas, float *rgbs,
                                       float *weights, float *trans, float *alphas, void *stream)
{
    CED_REQUIRE(n_rays >= 0 && n_all >= 0 && processed >= 0, "render_image_gather: bad sizes");
    if (processed == 0) return CED_OK;
    CED_REQUIRE(workspace && ray_offsets && ray_indices && t_starts && t_ends, "render_image_gather: null pointer");
    CED_REQUIRE((rgbs && weights && trans && alphas) || (!rgbs && !weights && !trans && !alphas),
                "render_image_gather: rgbs / weights / trans / alphas go together");
    const ced::ImageWorkspace W = ced::carve_image(const_cast<void *>(workspace), n_rays, ced::image_capacity(n_rays, n_all));
    CED_REQUIRE((int64_t)W.bytes <= workspace_bytes && processed <= ced::image_capacity(n_rays, n_all),
                "render_image_gather: workspace does not match");
    int64_t grid = (processed + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(ced::image_gather_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, processed, W.rank,
                       W.ridx, W.t0, W.t1, W.sigma, W.rgbs, W.w, W.tr, W.al, ray_offsets, chunk_rays,
                       ced::ImageOut{ ray_indices, t_starts, t_ends, sigmas, rgbs, weights, trans, alphas });
    return ced::check_launch("render_image_gather");
}


extern "C" int ced_march_all(int64_t n_rays, const float *rays_o, const float *rays_d, const uint8_t *binaries,
                             int32_t n_grids, int32_t res, const float *aabbs, const void *accel, const float *near_planes,
                             float far_plane, float step_size, float cone_angle, const float *t_sorted,
                             const int64_t *t_indices, const uint8_t *hits, int32_t fill, int64_t *packed_info,
                             float *t_starts, float *t_ends, int64_t *ray_indices, int64_t capacity, int64_t *total,
                             void *stream)
{
    CED_REQUIRE(n_rays >= 0 && res >= 1 && res <= 1024 && n_grids >= 1 && n_grids <= ced::kMaxGrids, "march_all: bad sizes");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(rays_o && rays_d && binaries && aabbs && accel && near_planes && packed_info, "march_all: null pointer");
    CED_REQUIRE(n_grids == 1 || (t_sorted && t_indices && hits), "march_all: %d grid levels need the sorted intersections", n_grids);
    CED_REQUIRE(!fill || (t_starts && t_ends), "march_all: fill pass without sample arrays");
    ced::MarchAllArgs A{ n_rays, rays_o, rays_d,
                         ced::GridSpec{ binaries, aabbs, n_grids, res, step_size, cone_angle, 0x7fffffff, nullptr },
                         ced::accel_view(accel, n_grids, res, true), near_planes, far_plane, t_sorted, t_indices, hits,
                         packed_info, t_starts, t_ends, ray_indices };
    const dim3 grid((unsigned)((n_rays + ced::kMarchThreads - 1) / ced::kMarchThreads)), blk(ced::kMarchThreads);
    hipStream_t st = (hipStream_t)stream;
    if (fill == 2) {
        CED_REQUIRE(capacity >= 0 && total != nullptr, "march_all: the one-pass form needs a capacity and a device counter");
        if (n_grids == 1)
            hipLaunchKernelGGL((ced::march_all_onepass_kernel<true>), grid, blk, 0, st, A, capacity,
                               reinterpret_cast<unsigned long long *>(total));
        else
            hipLaunchKernelGGL((ced::march_all_onepass_kernel<false>), grid, blk, 0, st, A, capacity,
                               reinterpret_cast<unsigned long long *>(total));
        return ced::check_launch("march_all (one pass)");
    }
    if (n_grids == 1) {
        if (fill) hipLaunchKernelGGL((ced::march_all_kernel<true, true>), grid, blk, 0, st, A);
        else hipLaunchKernelGGL((ced::march_all_kernel<true, false>), grid, blk, 0, st, A);
    } else {
        if (fill) hipLaunchKernelGGL((ced::march_all_kernel<false, true>), grid, blk, 0, st, A);
        else hipLaunchKernelGGL((ced::march_all_kernel<false, false>), grid, blk, 0, st, A);
    }
    return ced::check_launch("march_all");
}
