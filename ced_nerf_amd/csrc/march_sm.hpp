// traverse_ray_frame (march_accel.hpp, the form of a frame's first iteration: sphere trace at every segment's start,
// looking loop + emission phase) as a RESUMABLE STATE MACHINE: the ray's whole state lives in a struct, and every phase of
// the walk -- segment selection, one distance-field probe, DDA set-up, closed-form re-entry, one look-ahead batch, the
// emission of a batch -- is one step function.  A wave can then keep 64 rays in flight that are in DIFFERENT phases and
// hand a lane whose ray has finished the next ray of the work list (frame.hip: march_cand_sm_kernel): the lanes of a
// wave no longer wait for the wave's slowest ray, and the trip counts of the phase loops (22 probe passes in a heavy
// wave for ~10 per lane, 12 walk batches for ~4) are filled with other rays' steps.
// Same operations per ray in the same order as traverse_ray_frame<LOOK, SINGLE, true>: bit-identical samples.  Host and
// device code: the CPU test-suite runs one lane of this machine to completion against the oracle.
#pragma once
#include "march_accel.hpp"

namespace ced {

enum : int { SM_IDLE = 0, SM_SEG, SM_TRACE, SM_SETUP, SM_REENTER, SM_WALK, SM_EMIT, SM_FIN, SM_DONE };

constexpr int kSmLook = 4;
static_assert(kSmLook == 4, "RaySM keeps the batch's four boundary times as scalars");

struct RaySM {
    float o[3], d[3], inv_d[3];
    float near, far;
    int phase, limit, seg_i, lvl, n;
    float tmin, tmax;                       // the current segment
    // sphere trace
    float t_tr, t_from, g, inv_g, sc[3], t_stop, cells;
    int probes;
    // exact DDA of the segment
    float tdist[3], delta[3], t_c;
    int cur[3], stp[3], ovf[3], safe_cell;
    int dda_done, setup_done;
    // the look-ahead batch between the walk and the emission
    float tt0, tt1, tt2, tt3;                // (scalars: as an array member the batch times kept the whole struct in scratch)
    unsigned valid_mask, occ_mask;
    int radius;
    int last_empty;
    // sample lattice
    float t_last, skip_to, t_term;
    int continuous, has_skip;
};

CED_HD void sm_push_skip(RaySM &s, float target)
{
    s.skip_to = s.has_skip ? fmaxf(s.skip_to, target) : target;
    s.has_skip = true;
}

CED_HD void sm_begin(RaySM &s, const float (&o)[3], const float (&d)[3], float near, float far, int limit)
{
#pragma unroll
    for (int a = 0; a < 3; ++a) { s.o[a] = o[a]; s.d[a] = d[a]; s.inv_d[a] = 1.0f / d[a]; }
    s.near = near; s.far = far; s.limit = limit;
    s.seg_i = 0; s.n = 0; s.lvl = 0;
    s.t_last = near; s.t_term = near; s.continuous = false; s.has_skip = false; s.skip_to = 0.0f;
    s.phase = SM_SEG;
}

// after a segment is over (or before the first): the next segment the reference visits, its trace set up; SM_FIN when
// the sample budget is used up or no segment is left
template <bool SINGLE, class Idx>
CED_HD void sm_seg(RaySM &s, const GridSpec &G, const float *__restrict__ ts_row, const Idx *__restrict__ ti_row,
                   const uint8_t *__restrict__ hit_row)
{
    const int n_grids = SINGLE ? 1 : G.n_grids;
    if (s.n >= s.limit) { s.t_term = s.t_last; s.phase = SM_FIN; return; }
    for (; s.seg_i < 2 * n_grids - 1; ++s.seg_i) {
        const int i = s.seg_i;
        int lvl = 0;
        float seg_a, seg_b;
        if constexpr (SINGLE) {
            if (!slab_test(s.o, s.inv_d, G.aabbs, seg_a, seg_b)) break;
        } else {
            const int ti = (int)ti_row[i];
            const bool entering = ti < n_grids;
            lvl = ti % n_grids;
            if (!hit_row[lvl]) continue;
            if (!entering) {
                const int tn = (int)ti_row[i + 1];
                if (tn < n_grids) continue;
                lvl = tn % n_grids;
                if (!hit_row[lvl]) continue;
            }
            seg_a = ts_row[i];
            seg_b = ts_row[i + 1];
        }
        const float this_tmin = fmaxf(seg_a, s.near), this_tmax = fminf(seg_b, s.far);
        if (this_tmin >= this_tmax) continue;
        if (!s.continuous) sm_push_skip(s, this_tmin);
        s.lvl = lvl; s.tmin = this_tmin; s.tmax = this_tmax;
        s.seg_i = i + 1;
        s.setup_done = false; s.dda_done = false; s.t_c = this_tmin;
        // coarse_advance's prologue
        const float *ab = G.aabbs + 6 * lvl;
        const float resf = (float)G.res;
        float g = 0.0f;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s.sc[a] = resf / (ab[3 + a] - ab[a]);
            g = fmaxf(g, fabsf(s.d[a]) * s.sc[a]);
        }
        s.g = g;
        s.t_from = this_tmin; s.t_tr = this_tmin; s.t_stop = this_tmin; s.cells = 0.0f; s.probes = 0;
        if (!(g > 0.0f) || !(g < 3.0e38f) || !(this_tmax - this_tmin < 3.0e38f)) { s.phase = SM_SETUP; return; }   // the trace declines
        s.inv_g = 1.0f / g;
        s.phase = SM_TRACE;
        return;
    }
    s.seg_i = 2 * n_grids - 1;
    s.t_term = s.t_last;
    s.phase = SM_FIN;
}

// a trace that starts in the middle of a segment (the walk found room around the cell it stands in)
CED_HD void sm_trace_from_walk(RaySM &s)
{
    s.t_from = s.t_c; s.t_tr = s.t_c; s.t_stop = s.t_c; s.cells = 0.0f; s.probes = 0;
    if (!(s.g > 0.0f) || !(s.g < 3.0e38f) || !(s.tmax - s.t_c < 3.0e38f)) { s.phase = SM_REENTER; return; }
    s.phase = SM_TRACE;
}

// one probe of coarse_advance
CED_HD void sm_probe(RaySM &s, const GridSpec &G, const AccelSpec &S)
{
    const int res = G.res;
    const float *ab = G.aabbs + 6 * s.lvl;
    if (s.probes >= 8192) {                       // coarse_advance's guard: gives up without a skip
        s.t_stop = s.t_from; s.cells = 0.0f;
        s.phase = s.setup_done ? SM_REENTER : SM_SETUP;
        return;
    }
    ++s.probes;
    int c[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) c[a] = clampi((int)((s.o[a] + s.d[a] * s.t_tr - ab[a]) * s.sc[a]), 0, res - 1);
    const int D = empty_radius(S, s.lvl, res, c[0], c[1], c[2]);
    if (D < 4) {
        s.t_stop = s.t_tr; s.cells = (s.t_tr - s.t_from) * s.g;
        s.phase = s.setup_done ? SM_REENTER : SM_SETUP;
        return;
    }
    s.t_tr += (float)(D - 3) * s.inv_g;
    if (s.t_tr >= s.tmax) {                       // the rest of the segment is empty cells
        s.continuous = false;
        s.phase = SM_SEG;
    }
}

// DDA set-up of the segment: the reference's arithmetic, operation for operation
CED_HD void sm_setup(RaySM &s, const GridSpec &G)
{
    const float eps = 1e-6f;
    const int res = G.res;
    const float resf = (float)res;
    const float *ab = G.aabbs + 6 * s.lvl;
    const float ts = s.tmin + eps, te = s.tmax - eps;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float ext = ab[3 + a] - ab[a];
        const float vox = ext / resf;
        const float ps = s.o[a] + s.d[a] * ts;
        const float pe = s.o[a] + s.d[a] * te;
        s.cur[a] = clampi((int)(((ps - ab[a]) / ext) * resf), 0, res - 1);
        const int fin = clampi((int)(((pe - ab[a]) / ext) * resf), 0, res - 1);
        const int idelta = s.d[a] > 0.0f ? 1 : 0;
        const float tm = ((ab[a] + (((float)(s.cur[a] + idelta) * vox) - ps)) * s.inv_d[a]) + s.tmin;
        const float stepf = (s.d[a] == 0.0f) ? 0.0f : (s.d[a] > 0.0f ? 1.0f : -1.0f);
        s.stp[a] = (int)stepf;
        s.tdist[a] = (s.d[a] == 0.0f) ? s.tmax : tm;
        s.delta[a] = (s.d[a] == 0.0f) ? s.tmax : (vox * s.inv_d[a]) * stepf;
        s.ovf[a] = fin + s.stp[a];
    }
    s.safe_cell = (s.cur[0] * res + s.cur[1]) * res + s.cur[2];
    s.setup_done = true;
    s.phase = SM_REENTER;
}

// closed-form re-entry of the exact DDA at t_stop (only for stretches worth it), then the walk
CED_HD void sm_reenter(RaySM &s)
{
    s.phase = SM_WALK;
    if (!(s.cells >= kMinJumpCells)) return;
    float last_event = s.t_c;
    bool any = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int kcap = s.stp[a] > 0 ? s.ovf[a] - s.cur[a] : (s.stp[a] < 0 ? s.cur[a] - s.ovf[a] : 0);
        if (kcap <= 0) continue;
        float prev = 0.0f;
        const int k = count_steps(s.tdist[a], s.delta[a], s.t_stop, kcap, prev);
        if (k > 0) {
            s.cur[a] += k * s.stp[a];
            last_event = any ? fmaxf(last_event, prev) : prev;
            any = true;
            if (k == kcap) s.dda_done = true;
        }
    }
    if (any) {
        s.continuous = false;
        const float t_in = fminf(last_event, s.tmax);
        sm_push_skip(s, t_in);
        s.t_c = t_in;
    }
    if (s.dda_done) s.phase = SM_SEG;
}

// one look-ahead batch of the exact walk
CED_HD void sm_walk(RaySM &s, const GridSpec &G, const AccelSpec &S)
{
    constexpr int LOOK = kSmLook;
    const int res = G.res;
    const uint8_t *grid = G.binaries + (int64_t)s.lvl * res * res * res;
    int cellv[LOOK];
    float tt[LOOK];
    s.valid_mask = 0;
#pragma unroll
    for (int b = 0; b < LOOK; ++b) {
        const bool live = !s.dda_done;
        s.valid_mask |= live ? (1u << b) : 0u;
        tt[b] = fminf(fminf(s.tdist[0], fminf(s.tdist[1], s.tdist[2])), s.tmax);
        const int cell = (s.cur[0] * res + s.cur[1]) * res + s.cur[2];
        s.safe_cell = live ? cell : s.safe_cell;
        cellv[b] = s.safe_cell;
        const bool sx = (s.tdist[0] < s.tdist[1]) && (s.tdist[0] < s.tdist[2]);
        const bool sy = !sx && (s.tdist[1] < s.tdist[2]);
        const bool sz = !sx && !sy;
        const float nx = s.tdist[0] + s.delta[0], ny = s.tdist[1] + s.delta[1], nz = s.tdist[2] + s.delta[2];
        s.tdist[0] = (live && sx) ? nx : s.tdist[0];
        s.tdist[1] = (live && sy) ? ny : s.tdist[1];
        s.tdist[2] = (live && sz) ? nz : s.tdist[2];
        s.cur[0] += (live && sx) ? s.stp[0] : 0;
        s.cur[1] += (live && sy) ? s.stp[1] : 0;
        s.cur[2] += (live && sz) ? s.stp[2] : 0;
        const bool over = (sx && s.cur[0] == s.ovf[0]) || (sy && s.cur[1] == s.ovf[1]) || (sz && s.cur[2] == s.ovf[2]);
        s.dda_done = s.dda_done || (live && over);
    }
    uint8_t occ[LOOK];
#pragma unroll
    for (int b = 0; b < LOOK; ++b) occ[b] = grid[cellv[b]];
    {
        const int c0 = s.dda_done ? 0 : s.cur[0], c1 = s.dda_done ? 0 : s.cur[1], c2 = s.dda_done ? 0 : s.cur[2];
        s.radius = empty_radius(S, s.lvl, res, c0, c1, c2);
    }
    s.occ_mask = 0;
#pragma unroll
    for (int b = 0; b < LOOK; ++b) s.occ_mask |= occ[b] ? (1u << b) : 0u;
    s.occ_mask &= s.valid_mask;
    s.last_empty = false;
#pragma unroll
    for (int b = 0; b < LOOK; ++b) {
        const bool before = (s.valid_mask >> b & 1u) && (s.occ_mask & ((2u << b) - 1u)) == 0u;
        if (before) {
            sm_push_skip(s, tt[b]);
            s.continuous = false;
            s.last_empty = true;
            s.t_c = tt[b];
        }
    }
    s.tt0 = tt[0]; s.tt1 = tt[1]; s.tt2 = tt[2]; s.tt3 = tt[3];
    if (s.occ_mask) { s.phase = SM_EMIT; return; }
    const bool coarse = s.last_empty && s.radius >= kCoarseRadius;
    if (s.dda_done) s.phase = SM_SEG;
    else if (coarse) sm_trace_from_walk(s);
    // else: another batch
}

// the batch's cells from its first occupied one on
template <class Emit>
CED_HD void sm_emit(RaySM &s, const GridSpec &G, Emit &&emit)
{
    constexpr int LOOK = kSmLook;
    const float step_size = G.step_size, cone_angle = G.cone_angle;
    for (int b = __builtin_ctz(s.occ_mask); b < LOOK; ++b) {
        if (!(s.valid_mask >> b & 1u)) break;
        const float t_trav = b == 0 ? s.tt0 : (b == 1 ? s.tt1 : (b == 2 ? s.tt2 : s.tt3));
        if (!(s.occ_mask >> b & 1u)) {
            sm_push_skip(s, t_trav);
            s.continuous = false;
            s.last_empty = true;
            s.t_c = t_trav;
            continue;
        }
        s.last_empty = false;
        if (s.has_skip) { s.t_last = skip_march_lattice(G, s.t_last, s.skip_to); s.has_skip = false; }
        for (;;) {
            float t_next;
            if (step_size <= 0.0f) {
                t_next = t_trav;
            } else {
                const float dt = calc_dt(s.t_last, cone_angle, step_size, 1e10f);
                if (s.t_last + dt * 0.5f >= t_trav) break;
                t_next = s.t_last + dt;
            }
            emit(s.n, s.t_last, t_next);
            s.n += 1;
            s.continuous = true;
            s.t_last = t_next;
            if (s.n >= s.limit) { s.t_term = s.t_last; s.phase = SM_FIN; return; }
            if (t_next >= t_trav) break;
        }
    }
    const bool coarse = s.last_empty && s.radius >= kCoarseRadius;
    if (s.dda_done) s.phase = SM_SEG;
    else if (coarse) sm_trace_from_walk(s);
    else s.phase = SM_WALK;
}

// one lane of the machine run to completion (the CPU test-suite's entry; the device kernel steps 64 of them)
template <bool SINGLE, class Idx, class Emit>
CED_HD int sm_run_ray(const GridSpec &G, const AccelSpec &S, const float (&o)[3], const float (&d)[3], float near,
                      float far, const float *__restrict__ ts_row, const Idx *__restrict__ ti_row,
                      const uint8_t *__restrict__ hit_row, Emit &&emit, float &t_term)
{
    RaySM s;
    sm_begin(s, o, d, near, far, G.limit);
    for (long guard = 0; guard < (1l << 26) && s.phase != SM_FIN; ++guard) {
        switch (s.phase) {
        case SM_SEG: sm_seg<SINGLE>(s, G, ts_row, ti_row, hit_row); break;
        case SM_TRACE: sm_probe(s, G, S); break;
        case SM_SETUP: sm_setup(s, G); break;
        case SM_REENTER: sm_reenter(s); break;
        case SM_WALK: sm_walk(s, G, S); break;
        case SM_EMIT: sm_emit(s, G, emit); break;
        default: s.phase = SM_FIN; break;
        }
    }
    t_term = s.t_term;
    return s.n;
}

}  // namespace ced
