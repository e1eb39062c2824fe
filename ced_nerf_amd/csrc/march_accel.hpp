// Occupancy-grid traversal of the frame renderer (frame.hip), second generation: the same per-ray sample sets as
// nerfacc.traverse_grids (restated in march_core.hpp / the CPU oracle; call site cednerf/utils.py:241-264), but empty
// space costs O(1) per stretch instead of one DDA step per cell:
//
//   * a CHEBYSHEV DISTANCE FIELD over the grid (per cell: a lower bound of the distance in cells to the nearest
//     occupied cell; per 8^3-cell brick when only the cheap brick-level field exists) is sphere-traced along the
//     ray -- a probe with empty radius D allows D - 3 cells of travel on every axis with nothing occupied in reach
//     (conservative: the exact DDA's cells stay within one cell of the ideal line);
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

// Diagnostic build (-DCED_MARCH_DIAG, tools/march_diag.py): every phase of the walk reports how many lanes of the wave
// are in it and how long the wave stays in it.  Nothing in the shipped library.
#if defined(CED_MARCH_DIAG) && defined(__HIP_DEVICE_COMPILE__)
__device__ void ced_diag_tick(int phase);
#define CED_DIAG_TICK(p) ced_diag_tick(p)
#else
#define CED_DIAG_TICK(p) ((void)0)
#endif

namespace ced {

constexpr int kBrickShift = 3;                 // kBrick == 8
#ifndef CED_FRAME_LOOK
#define CED_FRAME_LOOK 4
#endif
constexpr int kFrameLook = CED_FRAME_LOOK;     // look-ahead of the frame renderer's exact walk (cells per batch)
#ifndef CED_MIN_JUMP_CELLS
#define CED_MIN_JUMP_CELLS 3.0f
#endif
#ifndef CED_COARSE_RADIUS
#define CED_COARSE_RADIUS 6
#endif
constexpr float kMinJumpCells = CED_MIN_JUMP_CELLS;   // the closed-form re-entry costs about as much as walking this many cells
constexpr int kCoarseRadius = CED_COARSE_RADIUS;      // the exact walk hands over to the sphere trace at this empty radius
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

// All rays of a frame march on ONE lattice when cone_angle == 0: t_0 = near plane, t_{k+1} = fl(t_k + step)
// (termination planes are lattice points too).  G.lattice[e] = the first lattice point whose biased exponent is e
// (NaN: none), built once per call (build_lattice): a skip to a far target -- the first sample of a ray that enters
// the scene, ten binades from the near plane -- then starts one binade short of the target instead of at t_last.
CED_HD float skip_march_lattice(const GridSpec &G, float t_last, float target)
{
    if (G.lattice && G.step_size > 0.0f && G.cone_angle == 0.0f) {
        const float A = target - G.step_size * 0.5f;
        const int e = (int)((float_to_bits(A) >> 23) & 0xffu);
        // the answer is > A - step; the first lattice point of binade e-1 is below that whenever the binade is
        // much wider than the step (it is at most 2^(e-128) + step)
        if ((float_to_bits(A) >> 31) == 0 && e >= 2 && e <= 254 && bits_to_float((uint32_t)(e - 1) << 23) > 4.0f * G.step_size) {
            const float first = G.lattice[e - 1];
            if (first > t_last && first < A - G.step_size) t_last = first;
        }
    }
    return skip_march(t_last, target, G.step_size, G.cone_angle);
}

// Fills lattice[0..255] for the lattice that starts at `near` (see skip_march_lattice).  One thread.
CED_HD void build_lattice(float near, float step, float *lattice)
{
    for (int e = 0; e < 256; ++e) lattice[e] = bits_to_float(0x7fc00000u);
    if (!(step > 0.0f) || !(near >= 0.0f)) return;
    float t = near;
    for (int guard = 0; guard < 1024; ++guard) {
        const int e = (int)((float_to_bits(t) >> 23) & 0xffu);
        if (e >= 255 || t + step == t) break;
        if (!(lattice[e] == lattice[e])) lattice[e] = t;
        // first lattice point of a later binade: skip to the top of this one (target = 2^(e-126), so that the skip
        // stops at the first t with t + h >= top), then single steps across
        const float top = e == 0 ? bits_to_float(1u << 23) : bits_to_float((uint32_t)(e + 1) << 23);
        if (!(top < 3.0e38f)) break;
        float u = skip_march_const_step(t, top, step);
        int steps = 0;
        while ((int)((float_to_bits(u) >> 23) & 0xffu) == e && steps < 4) { const float v = u + step; if (v == u) return; u = v; ++steps; }
        if ((int)((float_to_bits(u) >> 23) & 0xffu) == e) return;          // the step no longer moves t: the lattice ends
        t = u;
    }
}

// nerfacc.ray_aabb_intersect with near = -inf, far = +inf (cednerf/utils.py:215), one ray and one box
CED_HD bool slab_test(const float (&o)[3], const float (&inv_d)[3], const float *__restrict__ aabb, float &tmin_out,
                      float &tmax_out)
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
    tmin_out = tmin;
    tmax_out = tmax;
    return true;
}

// Distance fields over the occupancy grid (accel.hip builds them):
//   bdist[lvl][bx][by][bz]  Chebyshev distance in BRICKS (8^3 cells) from brick b to the nearest brick holding an
//                           occupied cell, capped; 0 = the brick itself does.  Cheap (nb^3 bytes): built per call when
//                           the caller brings nothing.
//   cdist[lvl][x][y][z]     a lower bound of the Chebyshev distance in CELLS from the cell to the nearest occupied one
//                           (exact up to 15, the brick bound beyond; 0 = occupied).  Optional (res^3 bytes per level):
//                           with it the exact walk is only needed within three cells of an occupied cell.
// Bricks / cells outside the grid count as empty.
struct AccelSpec {
    const uint8_t *bdist;      // [n_grids, nb, nb, nb]; NULL: no acceleration (plain cell-by-cell walk)
    int nb;                    // ceil(res / kBrick)
    const uint8_t *cdist;      // [n_grids, res, res, res] or NULL
};

// lower bound of the Chebyshev cell distance from cell c to the nearest occupied cell of level lvl
CED_HD int empty_radius(const AccelSpec &S, int lvl, int res, int c0, int c1, int c2)
{
    if (S.cdist) return S.cdist[((size_t)lvl * res + c0) * res * res + (size_t)c1 * res + c2];
    const int R = S.bdist[(((size_t)lvl * S.nb + (c0 >> kBrickShift)) * S.nb + (c1 >> kBrickShift)) * S.nb + (c2 >> kBrickShift)];
    return R == 0 ? 0 : (R - 1) * kBrick + 1;
}

// Sphere-traces the distance field of level `lvl` from t_from towards t_to.  Returns true when nothing occupied can be
// met before t_to; otherwise false with t_stop = a time such that nothing occupied can be met in [t_from, t_stop) and
// the point at t_stop is within three cells of an occupied one (t_stop == t_from: no skip possible).
// A probe in cell c with empty radius D: every cell within D-1 of c is empty; travelling s cells per axis reaches cells
// at most floor(s)+1 away from c, and the exact DDA's cell is at most one more from the ideal line: s = D - 3.
CED_HD bool coarse_advance(const AccelSpec &S, int lvl, int res, const float *__restrict__ ab, const float (&o)[3],
                           const float (&d)[3], float t_from, float t_to, float &t_stop, float &cells_skipped)
{
    const float resf = (float)res;
    float sc[3], g = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        sc[a] = resf / (ab[3 + a] - ab[a]);                   // cells per unit length on axis a
        g = fmaxf(g, fabsf(d[a]) * sc[a]);                    // cells per unit t, Chebyshev
    }
    t_stop = t_from;
    cells_skipped = 0.0f;
    if (!(g > 0.0f) || !(g < 3.0e38f) || !(t_to - t_from < 3.0e38f)) return false;
    const float inv_g = 1.0f / g;
    float t = t_from;
    for (int guard = 0; guard < 8192; ++guard) {
        CED_DIAG_TICK(2);
        int c[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) c[a] = clampi((int)((o[a] + d[a] * t - ab[a]) * sc[a]), 0, res - 1);
        const int D = empty_radius(S, lvl, res, c[0], c[1], c[2]);
        if (D < 4) { t_stop = t; cells_skipped = (t - t_from) * g; return false; }
        t += (float)(D - 3) * inv_g;
        if (t >= t_to) return true;
    }
    t_stop = t_from;            // not reached in practice (every probe advances at least one cell)
    return false;
}

// Traverses one ray for the frame renderer; emit(i, t_start, t_end) for sample i = 0..n-1 in order.  See the
// contract at the top of the file.  start_coarse: sphere-trace from the start of every segment (first iteration of
// a frame: most rays miss everything); otherwise only after a stretch of empty cells (later iterations: a live ray
// stands in or next to occupied cells).  LOOK: cells the exact walk looks ahead (their occupancy bytes are fetched
// together).  SINGLE: one grid level -- the ray/box interval is recomputed here with the set-up kernel's arithmetic
// instead of being loaded (ts_row / ti_row / hit_row unused).  PHASED: the exact walk as a looking loop and an emission
// phase (a frame's first iteration: rays travel to their first occupied cell, each for its own number of batches);
// otherwise the emission inline in the batch (rays that stand inside the object and emit in every batch).  Same
// operations per ray in the same order either way.
template <int LOOK, bool SINGLE, bool PHASED, class Idx, class Emit>
CED_HD int traverse_ray_frame(const GridSpec &G, const AccelSpec &S, bool start_coarse, const float (&o)[3],
                              const float (&d)[3], float near, float far, const float *__restrict__ ts_row,
                              const Idx *__restrict__ ti_row, const uint8_t *__restrict__ hit_row, Emit &&emit,
                              float &t_term)
{
    const float eps = 1e-6f;
    const float inv_d[3] = { 1.0f / d[0], 1.0f / d[1], 1.0f / d[2] };
    const int n_grids = SINGLE ? 1 : G.n_grids, res = G.res, limit = G.limit;
    const float step_size = G.step_size, cone_angle = G.cone_angle;
    const float resf = (float)res;
    const bool accel = S.bdist != nullptr;
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
        int lvl = 0;
        float seg_a, seg_b;
        if constexpr (SINGLE) {
            if (!slab_test(o, inv_d, G.aabbs, seg_a, seg_b)) break;
        } else {
            const int ti = (int)ti_row[i];          // event ids are < 2 * n_grids (int64 in nerfacc's layout, bytes in the frame renderer's)
            const bool entering = ti < n_grids;
            lvl = (int)(ti % n_grids);
            if (!hit_row[lvl]) continue;
            if (!entering) {
                const int tn = (int)ti_row[i + 1];
                if (tn < n_grids) continue;
                lvl = (int)(tn % n_grids);
                if (!hit_row[lvl]) continue;
            }
            seg_a = ts_row[i];
            seg_b = ts_row[i + 1];
        }
        const float this_tmin = fmaxf(seg_a, near);
        const float this_tmax = fminf(seg_b, far);
        if (this_tmin >= this_tmax) continue;
        if (!continuous) push_skip(this_tmin);
        const float *ab = G.aabbs + 6 * lvl;
        // A segment that starts with a sphere trace is traced BEFORE the DDA is set up: most segments of a frame's
        // first iteration are empty, and those never need the set-up (twelve divisions).  The trace itself only needs
        // the ray and the level's box; what it returns is used by the first round below, unchanged.
        bool coarse = accel && start_coarse;
        bool traced = false;
        float t_stop0 = 0.0f, cells0 = 0.0f;
        if (coarse) {
            if (coarse_advance(S, lvl, res, ab, o, d, this_tmin, this_tmax, t_stop0, cells0)) {
                continuous = false;            // the whole segment is empty cells
                continue;
            }
            traced = true;
        }
        CED_DIAG_TICK(1);
        // DDA set-up: the reference's arithmetic, operation for operation
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
        float t_c = this_tmin;                 // the time at which the walk stands (entry of the current cell)
        bool dda_done = false;
        int safe_cell = (cur[0] * res + cur[1]) * res + cur[2];
        // Rounds of [sphere-trace] -> [closed-form re-entry] -> [exact walk until it wants to trace again].  The three
        // phases are separate loops on purpose: the lanes of a wave then spend their time in the same phase instead
        // of every lane's phase being issued on every trip of one big loop.
        while (!dda_done) {
            if (coarse) {
                coarse = false;
                float t_stop = t_stop0, cells = cells0;
                if (traced) {
                    traced = false;            // the trace from the segment's start, done above
                } else if (coarse_advance(S, lvl, res, ab, o, d, t_c, this_tmax, t_stop, cells)) {
                    continuous = false;        // the rest of the segment is empty cells
                    break;
                }
                if (cells >= kMinJumpCells) {          // shorter stretches are cheaper walked than jumped
                    CED_DIAG_TICK(3);
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
            if constexpr (PHASED) {
                // exact walk, LOOK cells at a time: the path does not depend on the occupancy values, so the bytes of
                // those cells -- and the empty radius of the cell the walk will stand in afterwards -- are fetched
                // together (branch-free look-ahead, independent loads).  The walk loop only LOOKS: it runs until a batch
                // holds an occupied cell (or the walk ends / wants to trace again) and leaves the batch for the emission
                // phase below.  The lanes of a wave reach their first occupied cell in different batches; with the
                // emission inline every such batch paid a pass through the emission code (the skip-march above all) for
                // a handful of lanes -- three quarters of the instructions of a frame's first iteration.
                float tt[LOOK];
                unsigned valid_mask = 0, occ_mask = 0;
                int radius = 0;
                bool last_empty = false;
                do {
                    CED_DIAG_TICK(4);
                    int cellv[LOOK];
                    valid_mask = 0;
    #pragma unroll
                    for (int b = 0; b < LOOK; ++b) {
                        const bool live = !dda_done;
                        valid_mask |= live ? (1u << b) : 0u;
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
                    uint8_t occ[LOOK];
    #pragma unroll
                    for (int b = 0; b < LOOK; ++b) occ[b] = grid[cellv[b]];
                    radius = 0;
                    if (accel) {
                        const int c0 = dda_done ? 0 : cur[0], c1 = dda_done ? 0 : cur[1], c2 = dda_done ? 0 : cur[2];
                        radius = empty_radius(S, lvl, res, c0, c1, c2);
                    }
                    occ_mask = 0;
    #pragma unroll
                    for (int b = 0; b < LOOK; ++b) occ_mask |= occ[b] ? (1u << b) : 0u;
                    occ_mask &= valid_mask;
                    // the empty cells in front of the batch's first occupied one (all of its cells when it has none)
                    last_empty = false;
    #pragma unroll
                    for (int b = 0; b < LOOK; ++b) {
                        const bool before = (valid_mask >> b & 1u) && (occ_mask & ((2u << b) - 1u)) == 0u;
                        if (before) {
                            push_skip(tt[b]);
                            continuous = false;
                            last_empty = true;
                            t_c = tt[b];
                        }
                    }
                    if (occ_mask) break;
                    // in empty space with room around the cell the walk stands in: trace ahead
                    coarse = last_empty && radius >= kCoarseRadius;
                } while (!dda_done && !coarse);
                if (occ_mask) {
                    // emission phase: the batch's cells from its first occupied one on, in order -- ONE copy of the code,
                    // every lane with its own cursor
                    CED_DIAG_TICK(5);
                    for (int b = __builtin_ctz(occ_mask); b < LOOK; ++b) {
                        if (!(valid_mask >> b & 1u)) break;                 // the walk ended inside the batch
                        float t_trav = tt[0];
    #pragma unroll
                        for (int k = 1; k < LOOK; ++k) t_trav = b == k ? tt[k] : t_trav;
                        if (!(occ_mask >> b & 1u)) {
                            push_skip(t_trav);
                            continuous = false;
                            last_empty = true;
                            t_c = t_trav;
                            continue;
                        }
                        last_empty = false;
                        if (has_skip) { t_last = skip_march_lattice(G, t_last, skip_to); has_skip = false; }
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
                    coarse = last_empty && radius >= kCoarseRadius;
                }
            } else {
                // exact walk, LOOK cells at a time: the path does not depend on the occupancy values, so the bytes of
                // those cells -- and the empty radius of the cell the walk will stand in afterwards -- are fetched
                // together (branch-free look-ahead, independent loads)
                do {
                    CED_DIAG_TICK(4);
                    float tt[LOOK];
                    int cellv[LOOK];
                    bool valid[LOOK];
    #pragma unroll
                    for (int b = 0; b < LOOK; ++b) {
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
                    uint8_t occ[LOOK];
    #pragma unroll
                    for (int b = 0; b < LOOK; ++b) occ[b] = grid[cellv[b]];
                    int radius = 0;
                    if (accel) {
                        const int c0 = dda_done ? 0 : cur[0], c1 = dda_done ? 0 : cur[1], c2 = dda_done ? 0 : cur[2];
                        radius = empty_radius(S, lvl, res, c0, c1, c2);
                    }
                    bool last_empty = false;
    #pragma unroll
                    for (int b = 0; b < LOOK; ++b) {
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
                        CED_DIAG_TICK(5);
                        if (has_skip) { t_last = skip_march_lattice(G, t_last, skip_to); has_skip = false; }
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
                    // in empty space with room around the cell the walk stands in: trace ahead
                    coarse = last_empty && radius >= kCoarseRadius;
                } while (!dda_done && !coarse);
            }
        }
    }
    t_term = t_last;            // n < limit: unspecified by contract (pending skips are not applied)
    return n;
}

}  // namespace ced
