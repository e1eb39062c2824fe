// Per-ray occupancy-grid traversal shared by the nerfacc-shaped kernels (march.hip) and the frame
// renderer (frame.hip).  Restates nerfacc.traverse_grids (un-vendored CUDA op; call sites
// cednerf/utils.py:241-264 and, through OccGridEstimator.sampling, cednerf/utils.py:115-125).
// Every float operation is a single IEEE op in a fixed order (-ffp-contract=off): sample
// boundaries are bit-exact with the CPU oracle -- do not "simplify" the arithmetic.
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#define CED_HD __host__ __device__ __forceinline__
#else
#define CED_HD inline
#endif

namespace ced {

CED_HD float bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
CED_HD uint32_t float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

CED_HD float calc_dt(float t, float cone_angle, float dt_min, float dt_max)
{
    float v = t * cone_angle;
    return fminf(fmaxf(v, dt_min), dt_max);
}

// The reference recurrence: advance t in whole steps until the next step's mid-point reaches
// `target` (nerfacc: "march until t_mid is right after t_traverse").
CED_HD float skip_march_sequential(float t_last, float target, float step_size, float cone_angle)
{
    if (step_size <= 0.0f) return target;
    for (;;) {
        float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
        if (t_last + dt * 0.5f >= target) break;
        t_last += dt;
    }
    return t_last;
}

// Same result as skip_march_sequential for cone_angle == 0, in O(#binades) instead of O(#steps).
// Inside one binade [2^e, 2^(e+1)) every t is a multiple of u = ulp, so fl(t + step) adds the same
// whole number of ulps each time (step = q*u + r rounds to q or q+1 ulps, the same way for every t
// of the binade, unless r is exactly u/2): the recurrence is an exact arithmetic progression
// t_k = t_0 + k*delta there.  Binade crossings, the tie case and tiny t take real single steps.
CED_HD float skip_march_const_step(float t, float target, float step)
{
    const float h = step * 0.5f;
    for (;;) {
        if (t + h >= target) return t;
        const float t1 = t + step;
        const uint32_t bt = float_to_bits(t), b1 = float_to_bits(t1);
        const uint32_t et = bt & 0x7f800000u;
        const bool same_binade = (et == (b1 & 0x7f800000u)) && (bt >> 31) == 0 && et > (24u << 23) && et < (254u << 23);
        if (!same_binade) { t = t1; continue; }
        const float delta = t1 - t;                         // exact: both are multiples of u in one binade
        const float err = step - delta;                     // exact rounding error of the add (|t| > |step| here)
        const float u = bits_to_float(et - (23u << 23));    // ulp of the binade
        if (!(delta > 0.0f) || fabsf(err) == 0.5f * u) { t = t1; continue; }
        const float hi = bits_to_float(et + (1u << 23));    // 2^(e+1)
        const double td = (double)t, dd = (double)delta;
        const double kmax = floor(((double)hi - (double)u - td) / dd);            // t_kmax is still inside the binade
        double k = ceil(((double)target - (double)h - td) / dd);
        k = k < 1.0 ? 1.0 : k;
        if (k >= kmax) {
            // the target is not reached inside this binade (or right at its end): go to the last
            // in-binade term and let the loop test it and take the crossing step for real
            k = kmax;
            // walk back while the predecessor already satisfies the predicate
            while (k > 1.0 && ((float)(td + (k - 1.0) * dd) + h >= target)) k -= 1.0;
            t = (float)(td + k * dd);
            continue;
        }
        // fix the estimate up against the exact float predicate (at most a couple of moves)
        while (k > 1.0 && ((float)(td + (k - 1.0) * dd) + h >= target)) k -= 1.0;
        while (k < kmax && !((float)(td + k * dd) + h >= target)) k += 1.0;
        t = (float)(td + k * dd);
    }
}

CED_HD float skip_march(float t_last, float target, float step_size, float cone_angle)
{
    if (step_size > 0.0f && cone_angle == 0.0f) return skip_march_const_step(t_last, target, step_size);
    return skip_march_sequential(t_last, target, step_size, cone_angle);
}

CED_HD int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

struct GridSpec {
    const uint8_t *binaries;   // [n_grids, res, res, res] bytes
    const float *aabbs;        // [n_grids, 6]
    int n_grids, res;
    float step_size, cone_angle;
    int limit;                 // <= 0: unlimited
    // Optional conservative early-out (frame renderer only): dilated brick occupancy
    // [n_grids, nb, nb, nb] bytes, nb = ceil(res / kBrick); brick b is set when any cell of the
    // 3x3x3 bricks around b is occupied.  NULL disables it.
    const uint8_t *dilated_bricks;
    int nb;
    // 0: also ask at the start of a segment whether all of it is clear (first iteration of a frame: most
    // rays miss everything); 1: only after a stretch of empty cells (later iterations: a live ray stands in
    // or next to occupied cells, the question at the start would almost always be answered "no")
    int skip_initial_probe;
};

constexpr int kBrick = 8;      // cells per brick side

#ifndef CED_KLOOK
#define CED_KLOOK 8
#endif
constexpr int kLook = CED_KLOOK;   // DDA look-ahead (cells whose occupancy bytes are fetched together)

#if defined(__HIPCC__)
// -DCED_MARCH_PROFILE: per-wave cycle counts of the sections of traverse_ray, summed into g_march_prof
// (diagnostic builds only; tools/debug_march_profile.py reads them)
#ifdef CED_MARCH_PROFILE
static __device__ unsigned long long g_march_prof[16];
#define CED_MP_DECL unsigned long long mp_t = __builtin_readcyclecounter(); unsigned long long mp_acc[8] = {0,0,0,0,0,0,0,0};
#define CED_MP(sec) { const unsigned long long mp_n = __builtin_readcyclecounter(); mp_acc[sec] += mp_n - mp_t; mp_t = mp_n; }
// the lane that was busy longest speaks for the wave (its marks cover the wave's whole run)
#define CED_MP_FLUSH { unsigned long long mp_tot = 0; for (int q = 0; q < 8; ++q) mp_tot += mp_acc[q]; \
        unsigned long long mp_max = mp_tot; \
        for (int off = 32; off > 0; off >>= 1) { const unsigned long long o_ = __shfl_xor(mp_max, off, 64); mp_max = o_ > mp_max ? o_ : mp_max; } \
        const unsigned long long mp_bal = __ballot(mp_tot == mp_max); \
        if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(mp_bal)) { for (int q = 0; q < 8; ++q) atomicAdd(&g_march_prof[q], mp_acc[q]); atomicAdd(&g_march_prof[8], 1ull); } }
#else
#define CED_MP_DECL
#define CED_MP(sec)
#define CED_MP_FLUSH
#endif

// Conservative emptiness test of the ray segment [t_a, t_b] on grid level `lvl`: true only if no cell
// the fine DDA can visit there is occupied.  Points are probed every 6 cells (of the smallest cell
// edge); a cell visited between two probes is < 8 cells (6 + DDA/rounding slop) from the earlier probe
// on every axis, i.e. inside the 3x3x3 bricks around that probe's brick, which the dilated mask covers.
__device__ __forceinline__ bool segment_clear(const GridSpec &G, int lvl, const float (&o)[3], const float (&d)[3],
                                              float t_a, float t_b)
{
    const float *ab = G.aabbs + 6 * lvl;
    const float resf = (float)G.res;
    float inv_ext[3], vmin = 3.4e38f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = ab[3 + a] - ab[a];
        inv_ext[a] = resf / ext;
        vmin = fminf(vmin, ext / resf);
    }
    const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    const float dt = 6.0f * vmin / fmaxf(dn, 1e-20f);
    const uint8_t *mask = G.dilated_bricks + (size_t)lvl * G.nb * G.nb * G.nb;
    if (!(dt > 0.0f) || !(t_b - t_a < 4096.0f * dt)) return false;      // degenerate / absurdly long: do not claim
    // probes are independent: fetch kProbe mask bytes per round trip
    constexpr int kProbe = 4;
    float t = t_a;
    {   // the first probe alone: a ray standing next to occupied cells is answered after one byte
        int b[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) b[a] = clampi((int)((o[a] + d[a] * fminf(t, t_b) - ab[a]) * inv_ext[a]), 0, G.res - 1) / kBrick;
        if (mask[(b[0] * G.nb + b[1]) * G.nb + b[2]]) return false;
    }
    for (;;) {
        uint8_t hit = 0;
#pragma unroll
        for (int p = 0; p < kProbe; ++p) {
            const float tc = fminf(t + (float)p * dt, t_b);
            int b[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int c = clampi((int)((o[a] + d[a] * tc - ab[a]) * inv_ext[a]), 0, G.res - 1);
                b[a] = c / kBrick;
            }
            hit |= mask[(b[0] * G.nb + b[1]) * G.nb + b[2]];
        }
        if (hit) return false;
        t += (float)kProbe * dt;
        // the last probe of this round sat at min(t - dt, t_b): done once it reached the segment end
        if (t - dt >= t_b) break;
    }
    return true;
}

// Traverses one ray; calls emit(i, t_start, t_end) for sample i = 0.. in order.  Returns the
// number of samples; t_term receives the termination plane.
template <class Emit>
__device__ __forceinline__ int traverse_ray(const GridSpec &G, const float (&o)[3], const float (&d)[3], float near,
                                            float far, const float *__restrict__ ts_row,
                                            const int64_t *__restrict__ ti_row, const uint8_t *__restrict__ hit_row,
                                            Emit &&emit, float &t_term
#ifdef CED_MARCH_PROFILE
                                            , unsigned long long (&mp_out)[8], unsigned long long &mp_out_t
#endif
)
{
    CED_MP_DECL
    const float eps = 1e-6f;
    const float inv_d[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    const int n_grids = G.n_grids, res = G.res, limit = G.limit;
    const float step_size = G.step_size, cone_angle = G.cone_angle;
    const float resf = (float)res;
    float t_last = near;
    bool continuous = false;
    int n = 0;
    // A segment that ends in empty cells owes one skip-march to its last empty boundary.  It is deferred to the next
    // processed segment (or to the end of the call): the t_last sequence is the same, but a ray that LEAVES the
    // occupied region -- and will be dead after this call of the frame renderer -- does not pay the closed-form
    // skip (double-precision divisions, executed by the whole wave) for a termination plane nobody reads.
    bool owed = false;
    float owed_to = 0.0f;
    CED_MP(0)                       // [0] kernel prologue: ray loads
    for (int i = 0; i < 2 * n_grids - 1; ++i) {
        // Sample budget used up: later segments change nothing (the cell loop would not run and
        // `continuous` is true right after an emission), so stop before any early-out can touch state.
        if (limit > 0 && n >= limit) break;
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
        float this_tmin = fmaxf(ts_row[i], near);
        float this_tmax = fminf(ts_row[i + 1], far);
        if (this_tmin >= this_tmax) continue;
        CED_MP(1)                   // [1] segment selection
        if (owed) { t_last = skip_march(t_last, owed_to, step_size, cone_angle); owed = false; }
        if (!continuous) t_last = skip_march(t_last, this_tmin, step_size, cone_angle);
        CED_MP(2)                   // [2] skip-march to the segment start
        const float *ab = G.aabbs + 6 * lvl;
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
        const uint8_t *grid = G.binaries + (int64_t)lvl * res * res * res;
        // Early-out (frame renderer): when the rest of this segment provably holds no occupied cell,
        // walking it cell by cell would emit nothing.  Skipping it leaves t_last on an earlier point of
        // the same step lattice, so any later sample is unchanged; only the termination plane of a ray
        // that ends the call short of `limit` -- a ray that is dead afterwards -- is not advanced.
        const bool can_skip = G.dilated_bricks != nullptr;
        CED_MP(3)                   // [3] DDA set-up
        if (can_skip && !G.skip_initial_probe && segment_clear(G, lvl, o, d, this_tmin, this_tmax)) { continuous = false; CED_MP(4) continue; }
        CED_MP(4)                   // [4] brick probes
        int empty_batches = 0;
        // The DDA path does not depend on the occupancy values, so it runs kLook cells ahead and the
        // occupancy bytes of those cells are fetched together.  Runs of empty cells only remember
        // the farthest boundary; the skip-march to it happens once, before the next occupied cell or
        // at the end -- the same t_last sequence as marching cell by cell, because the recurrence
        // t_last += dt does not depend on where the intermediate boundaries are.
        bool dda_done = false, stop = false, has_pending = false;
        float pending = 0.0f;
        int safe_cell = (cur[0] * res + cur[1]) * res + cur[2];
        while (!dda_done && !stop) {
            float tt[kLook];
            int cellv[kLook];
            bool valid[kLook];
            // branch-free look-ahead: predicated selects only, so the kLook occupancy loads below are
            // independent instructions in flight together (a data-dependent branch per cell would
            // serialise them behind s_waitcnt)
#pragma unroll
            for (int b = 0; b < kLook; ++b) {
                const bool live = !dda_done;
                valid[b] = live;
                tt[b] = fminf(fminf(tdist[0], fminf(tdist[1], tdist[2])), this_tmax);
                const int cell = (cur[0] * res + cur[1]) * res + cur[2];
                safe_cell = live ? cell : safe_cell;            // never form an out-of-grid address
                cellv[b] = safe_cell;
                const bool sx = (tdist[0] < tdist[1]) && (tdist[0] < tdist[2]);
                const bool sy = !sx && (tdist[1] < tdist[2]);
                const bool sz = !sx && !sy;
                const float nx = tdist[0] + delta[0], ny = tdist[1] + delta[1], nz = tdist[2] + delta[2];
                tdist[0] = (live && sx) ? nx : tdist[0];
                tdist[1] = (live && sy) ? ny : tdist[1];
                tdist[2] = (live && sz) ? nz : tdist[2];
                cur[0] += (live && sx) ? stp[0] : 0;
                cur[1] += (live && sy) ? stp[1] : 0;
                cur[2] += (live && sz) ? stp[2] : 0;
                const bool over = (sx && cur[0] == ovf[0]) || (sy && cur[1] == ovf[1]) || (sz && cur[2] == ovf[2]);
                dda_done = dda_done || (live && over);
            }
            CED_MP(5)               // [5] look-ahead DDA
            uint8_t occ[kLook];
#pragma unroll
            for (int b = 0; b < kLook; ++b) occ[b] = grid[cellv[b]];
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
                    emit(n, t_last, t_next);
                    n += 1;
                    continuous = true;
                    t_last = t_next;
                    if (t_next >= t_trav) break;
                }
            }
            CED_MP(6)               // [6] occupancy wait + per-cell emission
            if (can_skip && has_pending && !dda_done && !stop) {
                // in empty space: every 4th all-empty stretch, ask whether anything is left ahead
                if ((empty_batches++ & 3) == 0 && segment_clear(G, lvl, o, d, pending, this_tmax)) dda_done = true;
            } else {
                empty_batches = 0;
            }
            CED_MP(4)
        }
        if (has_pending) { owed = true; owed_to = pending; }
        CED_MP(2)
    }
    // frame renderer (the same licence as its brick early-out): the termination plane of a ray that ends the call
    // short of its budget is never read, so the owed skip is dropped; everyone else pays it here
    if (owed && !(G.dilated_bricks != nullptr && limit > 0 && n < limit)) t_last = skip_march(t_last, owed_to, step_size, cone_angle);
    t_term = t_last;
    CED_MP(1)
#ifdef CED_MARCH_PROFILE
    mp_out_t = mp_t;
    for (int q = 0; q < 8; ++q) mp_out[q] = mp_acc[q];
#endif
    return n;
}
#endif  // __HIPCC__

}  // namespace ced
