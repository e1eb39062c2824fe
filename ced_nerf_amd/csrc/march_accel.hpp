// Occupancy-grid traversal of the frame renderer (frame.hip), second generation: the same per-ray sample sets as
// nerfacc.traverse_grids (restated in march_core.hpp / the CPU oracle; call site cednerf/utils.py:241-264), but empty
// space costs O(1) per stretch instead of one DDA step per cell:
//
//   * a per-brick CHEBYSHEV DISTANCE FIELD (8^3-cell bricks; value = brick distance to the nearest occupied brick)
//     is sphere-traced along the ray -- a probe at distance R allows (R-1)*8 - 2 cells of travel on every axis with
//     nothing occupied in reach (conservative: the exact DDA's cells stay within one cell of the ideal line);
//   * where the probes arrive next to occupied bricks, the exact DDA state is RE-ENTERED IN CLOSED FORM: the DDA is a
//     merge of three per-axis sequences T_a(j) = fl(T_a(j-1) + delta_a), and inside one binade such a float recurrence
//     is an exact arithmetic progression of mantissas (the same fact skip_march_const_step uses), so "all events
//     with T < tau" is computed per axis with integer arithmetic -- bit for bit the state the cell-by-cell walk has
//     when it gets there.  Every float the emitted samples depend on (t_last lattice, cell boundaries t_trav) is
//     therefore unchanged; only which empty cells were looked at differs.
//
// Contract of traverse_ray_frame (what the frame loop observes): the emitted (t_start, t_end) pairs and their count
// n <= limit are those of the reference walk; t_term is exact when n == limit (the ray may stay alive and its
// termination plane becomes the next near plane, cednerf/utils.py:301) and unspecified otherwise (a ray that returns
// fewer samples than its budget is dead, utils.py:303-306, and nobody reads its plane).
// Host and device code: the CPU test-suite runs this file against the oracle through ced_host_march_frame.
#pragma once
#include <cmath>

#include "march_core.hpp"

namespace ced {

constexpr int kBrickShift = 3;                 // kBrick == 8
static_assert((1 << kBrickShift) == kBrick, "brick size");

// Emulates   k = 0; while (k < kcap && x < tau) { prev = x; x = x + d; ++k; }   (binary32, round to nearest even)
// in O(#binades).  Returns k; `prev` is the value before the last add (meaningful when k > 0).
CED_HD int count_steps(float &x, float d, float tau, int kcap, float &prev)
{
    int k = 0;
    for (;;) {
        if (k >= kcap || !(x < tau)) break;
        const float x1 = x + d;
        const uint32_t bx = float_to_bits(x), b1 = float_to_bits(x1);
        const uint32_t ex = bx & 0x7f800000u;
        const bool regular = (ex == (b1 & 0x7f800000u)) && (bx >> 31) == 0 && ex > (24u << 23) && ex < (254u << 23);
        if (regular) {
            const float inc = x1 - x;                            // exact: both are multiples of u in one binade
            const float err = d - inc;                           // exact rounding error of the add
            const float u = bits_to_float(ex - (23u << 23));     // ulp of the binade
            if (inc > 0.0f && fabsf(err) != 0.5f * u) {
                // mantissas: x = X * u, inc = INC * u; x_j = (X + j * INC) * u while it stays below 2^24
                const uint32_t X = (bx & 0x7fffffu) | 0x800000u;
                const uint32_t INC = (uint32_t)(inc / u);        // exact (power-of-two scaling)
                uint32_t LIM = 0x1000000u;                       // first mantissa that is not < min(tau, binade top)
                if (tau < bits_to_float(ex + (1u << 23))) LIM = (float_to_bits(tau) & 0x7fffffu) | 0x800000u;   // tau >= x: same binade
                // j_tau = #{ j >= 0 : X + j*INC < LIM } = floor((LIM - X - 1) / INC) + 1      (LIM > X here)
                const uint32_t B = LIM - X - 1u;
                uint32_t q = (uint32_t)((float)B / (float)INC);  // both < 2^24: exact operands, quotient off by <= 1
                if (q * INC > B) --q;
                if ((q + 1u) * INC <= B) ++q;
                uint32_t n = q + 1u;
                const uint32_t room = (uint32_t)(kcap - k);
                if (n > room) n = room;
                // the n-th add must itself stay inside the binade (a crossing add rounds on the next binade's grid)
                if (X + n * INC > 0xffffffu) --n;
                if (n >= 1u) {
                    prev = bits_to_float(ex | ((X + (n - 1u) * INC) & 0x7fffffu));
                    x = bits_to_float(ex | ((X + n * INC) & 0x7fffffu));
                    k += (int)n;
                    continue;
                }
            }
        }
        prev = x;
        x = x1;
        ++k;
    }
    return k;
}

// Brick distance field: dist[lvl][bx][by][bz] = Chebyshev distance (in bricks, capped at 255) from brick b to the
// nearest brick holding an occupied cell; 0 = the brick itself does.  Bricks outside the grid count as empty.
struct AccelSpec {
    const uint8_t *dist;       // [n_grids, nb, nb, nb]; NULL: no acceleration (plain cell-by-cell walk)
    int nb;                    // ceil(res / kBrick)
};

// Sphere-traces the distance field of level `lvl` from t_from towards t_to.  Returns true when nothing occupied can be
// met before t_to; otherwise false with t_stop = a time such that nothing occupied can be met in [t_from, t_stop) and
// the point at t_stop is within one brick of an occupied one (t_stop == t_from: no skip possible).
CED_HD bool coarse_advance(const AccelSpec &S, int lvl, int res, const float *__restrict__ ab, const float (&o)[3],
                           const float (&d)[3], float t_from, float t_to, float &t_stop)
{
    const float resf = (float)res;
    float sc[3], g = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        sc[a] = resf / (ab[3 + a] - ab[a]);                   // cells per unit length on axis a
        g = fmaxf(g, fabsf(d[a]) * sc[a]);                    // cells per unit t, Chebyshev
    }
    t_stop = t_from;
    if (!(g > 0.0f) || !(g < 3.0e38f) || !(t_to - t_from < 3.0e38f)) return false;
    const uint8_t *dist = S.dist + (size_t)lvl * S.nb * S.nb * S.nb;
    const float inv_g = 1.0f / g;
    float t = t_from;
    for (int guard = 0; guard < 4096; ++guard) {
        int b[3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
            b[a] = clampi((int)((o[a] + d[a] * t - ab[a]) * sc[a]), 0, res - 1) >> kBrickShift;
        const int R = dist[(b[0] * S.nb + b[1]) * S.nb + b[2]];
        if (R <= 1) { t_stop = t; return false; }
        // every brick within R-1 of this one is empty: (R-1)*8 cells of travel per axis, minus 2 cells of slop
        // (1 for the floor of the probe's own cell, 1 for the exact DDA's distance from the ideal line)
        t += (float)((R - 1) * kBrick - 2) * inv_g;
        if (t >= t_to) return true;
    }
    t_stop = t_from;            // not reached in practice (each probe advances >= 6 cells)
    return false;
}

// Traverses one ray for the frame renderer; emit(i, t_start, t_end) for sample i = 0..n-1 in order.  See the
// contract at the top of the file.  start_coarse: sphere-trace from the start of every segment (first iteration of
// a frame: most rays miss everything); otherwise only after a stretch of empty cells (later iterations: a live ray
// stands in or next to occupied cells).
template <class Emit>
CED_HD int traverse_ray_frame(const GridSpec &G, const AccelSpec &S, bool start_coarse, const float (&o)[3],
                              const float (&d)[3], float near, float far, const float *__restrict__ ts_row,
                              const int64_t *__restrict__ ti_row, const uint8_t *__restrict__ hit_row, Emit &&emit,
                              float &t_term)
{
    const float eps = 1e-6f;
    const float inv_d[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    const int n_grids = G.n_grids, res = G.res, limit = G.limit;
    const float step_size = G.step_size, cone_angle = G.cone_angle;
    const float resf = (float)res;
    const bool accel = S.dist != nullptr;
    float t_last = near;
    bool continuous = false;
    int n = 0;
    // Skip targets (segment starts, boundaries of empty cells) only ever grow along the ray and the skip recurrence
    // does not depend on intermediate targets, so they are applied lazily: once, right before the next emission.
    // (A target can be marginally smaller than the one before it -- a cell boundary computed a rounding error before
    // the segment start -- and applying both in order equals applying the larger: hence the max.)
    bool has_skip = false;
    float skip_to = 0.0f;
    auto push_skip = [&](float target) {
        skip_to = has_skip ? fmaxf(skip_to, target) : target;
        has_skip = true;
    };
    for (int i = 0; i < 2 * n_grids - 1; ++i) {
        if (n >= limit) break;
        const int64_t ti = ti_row[i];
        const bool entering = ti < n_grids;
        int lvl = (int)(ti % n_grids);
        if (!hit_row[lvl]) continue;
        if (!entering) {
            const int64_t tn = ti_row[i + 1];
            if (tn < n_grids) continue;
            lvl = (int)(tn % n_grids);
            if (!hit_row[lvl]) continue;
        }
        const float this_tmin = fmaxf(ts_row[i], near);
        const float this_tmax = fminf(ts_row[i + 1], far);
        if (this_tmin >= this_tmax) continue;
        if (!continuous) push_skip(this_tmin);
        // DDA set-up: the reference's arithmetic, operation for operation
        const float *ab = G.aabbs + 6 * lvl;
        float tdist[3], delta[3];
        int cur[3], stp[3], ovf[3];
        const float ts = this_tmin + eps, te = this_tmax - eps;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const float ext = ab[3 + a] - ab[a];
            const float vox = ext / resf;
            const float ps = o[a] + d[a] * ts;
            const float pe = o[a] + d[a] * te;
            cur[a] = clampi((int)(((ps - ab[a]) / ext) * resf), 0, res - 1);
            const int fin = clampi((int)(((pe - ab[a]) / ext) * resf), 0, res - 1);
            const int idelta = d[a] > 0.0f ? 1 : 0;
            const float tm = ((ab[a] + (((float)(cur[a] + idelta) * vox) - ps)) * inv_d[a]) + this_tmin;
            const float stepf = (d[a] == 0.0f) ? 0.0f : (d[a] > 0.0f ? 1.0f : -1.0f);
            stp[a] = (int)stepf;
            tdist[a] = (d[a] == 0.0f) ? this_tmax : tm;
            delta[a] = (d[a] == 0.0f) ? this_tmax : (vox * inv_d[a]) * stepf;
            ovf[a] = fin + stp[a];
        }
        const uint8_t *grid = G.binaries + (int64_t)lvl * res * res * res;
        bool coarse = accel && start_coarse;
        float t_c = this_tmin;                 // the time at which the walk stands (entry of the current cell)
        bool dda_done = false;
        int safe_cell = (cur[0] * res + cur[1]) * res + cur[2];
        while (!dda_done) {
            if (coarse) {
                coarse = false;
                float t_stop;
                if (coarse_advance(S, lvl, res, ab, o, d, t_c, this_tmax, t_stop)) {
                    continuous = false;        // the rest of the segment is empty cells
                    break;
                }
                if (t_stop > t_c) {
                    // re-enter the exact DDA at t_stop: per axis, take every boundary crossing with T < t_stop
                    float last_event = t_c;
                    bool any = false;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        // crossings left on this axis before the walk leaves its final cell (none for d == 0: that
                        // axis is never the strict minimum before the others have ended the walk)
                        const int kcap = stp[a] > 0 ? ovf[a] - cur[a] : (stp[a] < 0 ? cur[a] - ovf[a] : 0);
                        if (kcap <= 0) continue;
                        float prev = 0.0f;
                        const int k = count_steps(tdist[a], delta[a], t_stop, kcap, prev);
                        if (k > 0) {
                            cur[a] += k * stp[a];
                            last_event = any ? fmaxf(last_event, prev) : prev;
                            any = true;
                            if (k == kcap) dda_done = true;        // stepped out of the final cell: segment over
                        }
                    }
                    if (any) {
                        continuous = false;                         // the cells stepped over are empty
                        const float t_in = fminf(last_event, this_tmax);       // t_trav of the last of them
                        push_skip(t_in);
                        t_c = t_in;
                    }
                    if (dda_done) break;
                }
            }
            // exact walk, kLook cells ahead: the path does not depend on the occupancy values, so the bytes of
            // those cells are fetched together (branch-free look-ahead, independent loads)
            float tt[kLook];
            int cellv[kLook];
            bool valid[kLook];
#pragma unroll
            for (int b = 0; b < kLook; ++b) {
                const bool live = !dda_done;
                valid[b] = live;
                tt[b] = fminf(fminf(tdist[0], fminf(tdist[1], tdist[2])), this_tmax);
                const int cell = (cur[0] * res + cur[1]) * res + cur[2];
                safe_cell = live ? cell : safe_cell;                // never form an out-of-grid address
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
            bool last_empty = false;
#pragma unroll
            for (int b = 0; b < kLook; ++b) {
                if (!valid[b]) continue;
                const float t_trav = tt[b];
                if (!occ[b]) {
                    push_skip(t_trav);
                    continuous = false;
                    last_empty = true;
                    t_c = t_trav;
                    continue;
                }
                last_empty = false;
                if (has_skip) { t_last = skip_march(t_last, skip_to, step_size, cone_angle); has_skip = false; }
                for (;;) {
                    float t_next;
                    if (step_size <= 0.0f) {
                        t_next = t_trav;
                    } else {
                        const float dt = calc_dt(t_last, cone_angle, step_size, 1e10f);
                        if (t_last + dt * 0.5f >= t_trav) break;
                        t_next = t_last + dt;
                    }
                    emit(n, t_last, t_next);
                    n += 1;
                    continuous = true;
                    t_last = t_next;
                    if (n >= limit) { t_term = t_last; return n; }      // budget used up: the ray stays alive
                    if (t_next >= t_trav) break;
                }
            }
            if (accel && last_empty && !dda_done) {
                // in empty space: if the cell the walk stands in is at least two bricks from anything, trace ahead
                const uint8_t *dist = S.dist + (size_t)lvl * S.nb * S.nb * S.nb;
                const int R = dist[((cur[0] >> kBrickShift) * S.nb + (cur[1] >> kBrickShift)) * S.nb + (cur[2] >> kBrickShift)];
                coarse = R >= 2;
            }
        }
    }
    t_term = t_last;            // n < limit: unspecified by contract (pending skips are not applied)
    return n;
}

}  // namespace ced
