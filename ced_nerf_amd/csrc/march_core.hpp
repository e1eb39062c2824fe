// Per-ray occupancy-grid traversal shared by the nerfacc-shaped kernels (march.hip) and the frame
// renderer (frame.hip).  Restates nerfacc.traverse_grids (un-vendored CUDA op; call sites
// cednerf/utils.py:241-264 and, through OccGridEstimator.sampling, cednerf/utils.py:115-125).
// Every float operation is a single IEEE op in a fixed order (-ffp-contract=off): sample
// boundaries are bit-exact with the CPU oracle -- do not "simplify" the arithmetic.
#pragma once
#include <cmath>
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
// of the binade, unless r is exactly u/2): the recurrence is an exact arithmetic progression of the
// integer mantissas, t_j = (X + j*INC) * u, there.  The index is estimated by an integer division and
// settled by the exact float predicate fl(t_j + h) >= target; binade crossings, the tie case and tiny
// t take real single steps.  (No double precision: this runs once per ray and iteration on the device.)
CED_HD float skip_march_const_step(float t, float target, float step)
{
    const float h = step * 0.5f;
    if (!(target == target)) return t;                       // NaN target (degenerate ray): nothing to march to
    for (;;) {
        if (t + h >= target) return t;
        const float t1 = t + step;
        if (t1 == t) return t;               // the step no longer moves t (t beyond 2^24 steps): the lattice ends here
        const uint32_t bt = float_to_bits(t), b1 = float_to_bits(t1);
        const uint32_t ex = bt & 0x7f800000u;
        const bool regular = (ex == (b1 & 0x7f800000u)) && (bt >> 31) == 0 && ex > (24u << 23) && ex < (254u << 23);
        if (regular) {
            const float inc = t1 - t;
            const float err = step - inc;
            const float u = bits_to_float(ex - (23u << 23));
            if (inc > 0.0f && fabsf(err) != 0.5f * u) {
                const uint32_t X = (bt & 0x7fffffu) | 0x800000u;
                const uint32_t INC = (uint32_t)(inc / u);
                // J = last index whose term is still inside the binade
                const uint32_t BJ = 0xffffffu - X;
                uint32_t J = (uint32_t)((float)BJ / (float)INC);
                if (J * INC > BJ) --J;
                if ((J + 1u) * INC <= BJ) ++J;
                // estimate of the first j with t_j >= target - h, then the exact predicate settles it
                const float A = target - h;
                uint32_t j = J;
                if (A < bits_to_float(ex + (1u << 23))) {
                    j = 0;
                    if (A > t) {                              // same binade as t
                        const uint32_t BA = ((float_to_bits(A) & 0x7fffffu) | 0x800000u) - X;     // >= 1
                        uint32_t q = (uint32_t)((float)(BA - 1u) / (float)INC);
                        if (q * INC > BA - 1u) --q;
                        if ((q + 1u) * INC <= BA - 1u) ++q;
                        j = q + 1u;                           // ceil(BA / INC)
                        if (j > J) j = J;
                    }
                }
                while (j > 0u && (bits_to_float(ex | ((X + (j - 1u) * INC) & 0x7fffffu)) + h >= target)) --j;
                while (j < J && !(bits_to_float(ex | ((X + j * INC) & 0x7fffffu)) + h >= target)) ++j;
                t = bits_to_float(ex | ((X + j * INC) & 0x7fffffu));
                if (t + h >= target) return t;
                t = t + step;                                 // j == J: the step that leaves the binade, for real
                continue;
            }
        }
        t = t1;
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
    const float *lattice = nullptr;   // frame renderer, cone_angle == 0: first lattice point per binade (march_accel.hpp)
};

constexpr int kBrick = 8;      // cells per brick side

#ifndef CED_KLOOK
#define CED_KLOOK 8
#endif
constexpr int kLook = CED_KLOOK;   // DDA look-ahead (cells whose occupancy bytes are fetched together)

#if defined(__HIPCC__)
// Traverses one ray; calls emit(i, t_start, t_end) for sample i = 0.. in order.  Returns the
// number of samples; t_term receives the termination plane.
template <class Emit>
__device__ __forceinline__ int traverse_ray(const GridSpec &G, const float (&o)[3], const float (&d)[3], float near,
                                            float far, const float *__restrict__ ts_row,
                                            const int64_t *__restrict__ ti_row, const uint8_t *__restrict__ hit_row,
                                            Emit &&emit, float &t_term)
{
    const float eps = 1e-6f;
    const float inv_d[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    const int n_grids = G.n_grids, res = G.res, limit = G.limit;
    const float step_size = G.step_size, cone_angle = G.cone_angle;
    const float resf = (float)res;
    float t_last = near;
    bool continuous = false;
    int n = 0;
    // A segment that ends in empty cells owes one skip-march to its last empty boundary.  It is deferred to the next
    // processed segment (or to the end of the call): the t_last sequence is the same.
    bool owed = false;
    float owed_to = 0.0f;
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
        if (owed) { t_last = skip_march(t_last, owed_to, step_size, cone_angle); owed = false; }
        if (!continuous) t_last = skip_march(t_last, this_tmin, step_size, cone_angle);
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
        }
        if (has_pending) { owed = true; owed_to = pending; }
    }
    if (owed) t_last = skip_march(t_last, owed_to, step_size, cone_angle);
    t_term = t_last;
    return n;
}
#endif  // __HIPCC__

}  // namespace ced
