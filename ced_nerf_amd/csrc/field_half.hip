// Half-precision-MFMA variants of the fused dynamic-NGP field (same fusion and same interface as
// field.hip; selected by ced_field_desc.mlp_precision):
//
//   CED_MLP_F16X2 (1): every MLP operand is split into two fp16 numbers, x = hi + lo (22 significant
//       bits), and each product block is three fp16 MFMA blocks (hi*hi + hi*lo + lo*hi) with
//       fp32 accumulation -- fp32-grade results (~1e-6 relative) at a fraction of the fp32-MFMA time.
//   CED_MLP_F16   (2): operands rounded to fp16, fp32 accumulation: the precision class of the
//       reference's tiny-cuda-nn FullyFusedMLP (cednerf/model.py:200-222,280-309) and of
//       BASELINE config 5 ("fp16 hash features + fp16 MFMA MLP").
//
// Outside the GEMMs everything is the same deterministic fp32 code as the exact kernel (field_device.hpp, det_expf,
// IEEE division / square root): encodings (Frequency / SH / time), hash-grid gather and interpolation, position
// normalisation, trunc_exp, tanh, sigmoid.  Together with the CPU model of the matrix instruction
// (oracle/mfma_f16_model.h: two blocks of eight products per v_mfma_f32_16x16x16_f16, each cut below
// 2^(Emax - 24) and rounded once) every output of these kernels is reproducible on the CPU bit for bit.
//
// Geometry: D^T = W * X^T as in field.hip, but K = 32 per product block (mfma_k32, field_half_device.hpp): lane (g = lane>>4, c = lane&15) supplies
// inputs 32ks + 8g + e (e = 0..7, four packed VGPRs) of sample c, and receives accumulator rows 4g + r.
// The host packs the weight rows of every 64-wide hidden layer so that accumulator row 16nb + 4g + r
// holds neuron 32(nb>>1) + 8g + 4(nb&1) + r: the eight values a lane needs as operand of k-step ks are
// then its own registers D[2ks][0..3], D[2ks+1][0..3] -- activations never move between lanes.  The
// inputs are laid out likewise: lane group g computes the eight Frequency features of dimension g,
// gathers hash levels {g, 4+g, 8+g, 12+g} (input columns are permuted on the host to match), and the
// last layer of mlp_base is packed so that a lane's four outputs are its own mlp_head inputs.
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "ced_common.hpp"
#include "field_args.hpp"
#include "field_device.hpp"
#include "field_half_device.hpp"

namespace ced {

// packed blob, in fragments: layer l = [nb][ks] fragments; a second plane of the same shape holds the
// low parts in F16X2 mode
template <bool TE> struct HalfBlob {
    static constexpr int KS_B0 = TE ? 2 : 1;
    static constexpr int M0 = 0;
    static constexpr int M1 = M0 + 4 * 1;
    static constexpr int M2 = M1 + 4 * 2;
    static constexpr int M3 = M2 + 4 * 2;
    static constexpr int B0 = M3 + 1 * 2;
    static constexpr int B1 = B0 + 4 * KS_B0;
    static constexpr int H0 = B1 + 1 * 2;
    static constexpr int H1 = H0 + 4 * 1;
    static constexpr int H2 = H1 + 4 * 2;
    static constexpr int FRAGS = H2 + 1 * 2;       // 42 / 46
};

template <bool TE, bool F16, bool TEMPORAL, bool SPLIT, int NT, int THREADS>
__global__ __launch_bounds__(THREADS) void field_half_kernel(FieldArgs A)
{
    constexpr int WAVES = THREADS / kWave;
    constexpr int TILE = 16 * NT;
    using BL = HalfBlob<TE>;
    constexpr int PLANE = BL::FRAGS * kFragHalves;
    constexpr int WHALVES = PLANE * (SPLIT ? 2 : 1);
    __shared__ __attribute__((aligned(16))) _Float16 lds[WHALVES + 16 * CED_MAX_LEVELS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // the wave's number as a SCALAR: everything derived from it (the wave's tiles, their sample ranges, the bases of the
    // per-sample arrays) then lives in scalar registers and the per-lane part of an index is a 32-bit offset
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;

    int64_t n_eff = A.n;
    if (A.n_dev) {
        const int64_t nd = *A.n_dev;
        n_eff = nd < n_eff ? nd : n_eff;
    }
    // the call's window of persistent per-sample arrays (render_image, frame.hip): ray_idx / t0 / t1 / sigma / rgb
    // entry s of the call is entry sbase + s of the arrays.  (Kept as an index offset: adding it to the pointers of
    // the by-value argument block would make the compiler keep the whole block in scratch.)
    const int64_t sbase = A.base_dev ? *A.base_dev : 0;
    const int64_t n_tiles = (n_eff + TILE - 1) / TILE;
    // a workgroup without a tile leaves before staging anything (field_device.hpp: field_tile_range)
    TileRange tiles;
    if (!field_tile_range(A.spread_tiles, n_tiles, WAVES, wave, tiles)) return;
    if (A.stamp && tid == 0) atomicMin(A.stamp, (unsigned long long)wall_clock64());

    {
        const f4 *src = reinterpret_cast<const f4 *>(A.weights);
        f4 *dst = reinterpret_cast<f4 *>(lds);
        for (int i = tid; i < WHALVES / 8; i += THREADS) dst[i] = src[i];
        if (tid < CED_MAX_LEVELS) {
            uint32_t *lt = reinterpret_cast<uint32_t *>(lds + WHALVES);
            const LevelConst L = make_level(A.scale[tid], A.res[tid], A.offset[tid], A.size[tid], A.hashed[tid],
                                            EntryBytes<F16, TEMPORAL>::value);
            store_level(lt + tid * 8, L);
        }
    }
    __syncthreads();

    const float extent[3] = { A.aabb[3] - A.aabb[0], A.aabb[4] - A.aabb[1], A.aabb[5] - A.aabb[2] };

    for (int64_t tile = tiles.first; tile < tiles.end; tile += tiles.stride) {
        // opaque LDS base per tile: keeps the A-fragment reads inside the loop (see field.hip)
        int lds_off = 0;
        asm volatile("" : "+v"(lds_off));
        const _Float16 *const whi = lds + lds_off;
        const _Float16 *const wlo = whi + PLANE;
        // sample s of the call = tile_base + lane offset; a ragged last tile repeats its last sample (never stored)
        const int64_t tile_base = tile * TILE;
        const int64_t left = n_eff - tile_base;
        const int in_tile = left < TILE ? (int)left : TILE;                  // 1 .. TILE, scalar
        int sofs[NT];
        int64_t ridx[NT];
        float px[NT][3], tq[NT];
        bool any_used = !A.rays_mode;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            sofs[j] = 16 * j + c < in_tile ? 16 * j + c : in_tile - 1;
            const int64_t s = tile_base + sofs[j];
            if (A.rays_mode) {
                // negative ray index = unused sample slot (see field.hip)
                const int64_t r_in = A.ray_idx32 ? (int64_t)A.ray_idx32[sbase + s] : A.ray_idx[sbase + s];
                const bool used = r_in >= 0;
                const int64_t r = used ? r_in : 0;
                any_used = any_used || used;
                ridx[j] = r;
                const float tm2 = used ? A.t0[sbase + s] + A.t1[sbase + s] : 0.0f;
#pragma unroll
                for (int a = 0; a < 3; ++a) px[j][a] = A.rays_o[3 * r + a] + (A.rays_d[3 * r + a] * tm2) / 2.0f;
                tq[j] = A.t_per_ray ? A.timestamps[r] : A.timestamps[0];
            } else {
                ridx[j] = s;
#pragma unroll
                for (int a = 0; a < 3; ++a) px[j][a] = A.pos[3 * s + a];
                tq[j] = A.t[s];
            }
        }

        if (__ballot(any_used) == 0ull) continue;            // a tile of unused slots only (wave-uniform)

        h8 Bh[NT][2], Bl[NT][2];
        f4 D[NT][4];

        // --- tcnn Frequency(4) on (x,y,z,t): lane group g owns dimension g; e = 2*freq + phase ---
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float v = tq[j];
            v = (g == 0) ? px[j][0] : v;
            v = (g == 1) ? px[j][1] : v;
            v = (g == 2) ? px[j][2] : v;
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = det_sinpi_phase(v * (float)(1 << (e >> 1)), e & 1);
            to_half8<SPLIT>(f, Bh[j][0], Bl[j][0]);
        }
        // --- motion MLP 32-64-64-64-(3|6) ---
        mlp_layer_h<1, 4, NT, SPLIT>(whi + BL::M0 * kFragHalves, wlo + BL::M0 * kFragHalves, lane, Bh, Bl, D);
        to_operand_h<NT, SPLIT>(D, Bh, Bl);
        mlp_layer_h<2, 4, NT, SPLIT>(whi + BL::M1 * kFragHalves, wlo + BL::M1 * kFragHalves, lane, Bh, Bl, D);
        to_operand_h<NT, SPLIT>(D, Bh, Bl);
        mlp_layer_h<2, 4, NT, SPLIT>(whi + BL::M2 * kFragHalves, wlo + BL::M2 * kFragHalves, lane, Bh, Bl, D);
        to_operand_h<NT, SPLIT>(D, Bh, Bl);
        mlp_layer_h<2, 1, NT, SPLIT>(whi + BL::M3 * kFragHalves, wlo + BL::M3 * kFragHalves, lane, Bh, Bl, D);

        // --- query_move / normalise / selector (model.py:354-383); motion rows are natural ---
        float xn[NT][3], mnorm[NT];
        bool sel[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float mv[3];
            bool inside = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float off = __shfl(D[j][0][a], c, 64);          // row a: lane group 0, reg a
                float m = off * A.moving_step;
                if (A.use_div) {
                    constexpr int kFineReg[3] = { 3, 0, 1 };          // rows 3,4,5: (g0,r3), (g1,r0), (g1,r1)
                    const float fine = __shfl(D[j][0][kFineReg[a]], (a == 0) ? c : 16 + c, 64);
                    const float e = det_expf(2.0f * fine);
                    const float th = 1.0f - 2.0f / (e + 1.0f);
                    m = m + th * A.moving_step;
                }
                mv[a] = m;
                const float xm = px[j][a] + m;
                const float x = (xm - A.aabb[a]) / extent[a];      // IEEE division: 1 ulp here is amplified by the fine hash levels
                inside = inside && (x > 0.0f && x < 1.0f);
                xn[j][a] = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f);
            }
            sel[j] = inside;
            mnorm[j] = TE ? __builtin_sqrtf((mv[0] * mv[0] + mv[1] * mv[1]) + mv[2] * mv[2]) : 0.0f;
        }

        // --- hash gather: slot i of lane group g is level 4i + g (slot i spans levels 4i..4i+3 across the
        // wave: one index form when they are all dense or all hashed); features land at operand elements
        // 2i, 2i+1.  Level constants are fetched from LDS per use. ---
        float R[NT][8];
        int k_lo[NT];
        float t_frac[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            k_lo[j] = 0;
            t_frac[j] = 0.0f;
            if constexpr (TEMPORAL) temporal_keyframe(tq[j], k_lo[j], t_frac[j]);
        }
        const uint32_t *const ltab = reinterpret_cast<const uint32_t *>(whi + WHALVES);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const LevelConst L = load_level(ltab + (4 * i + g) * 8);
            const int mode = (A.level_mode >> (2 * i)) & 3;
            if (mode == 1) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    hash_level<F16, TEMPORAL, 1>(L, A.table, xn[j], k_lo[j], t_frac[j], R[j][2 * i], R[j][2 * i + 1]);
            } else if (mode == 2) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    hash_level<F16, TEMPORAL, 2>(L, A.table, xn[j], k_lo[j], t_frac[j], R[j][2 * i], R[j][2 * i + 1]);
            } else if (mode == 3) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    hash_level<F16, TEMPORAL, 3>(L, A.table, xn[j], k_lo[j], t_frac[j], R[j][2 * i], R[j][2 * i + 1]);
            } else {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    hash_level<F16, TEMPORAL, 0>(L, A.table, xn[j], k_lo[j], t_frac[j], R[j][2 * i], R[j][2 * i + 1]);
            }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#pragma unroll
            for (int e = 0; e < 8; ++e) R[j][e] = __builtin_amdgcn_fmed3f(R[j][e], -kHalfMax, kHalfMax);
            to_half8<SPLIT>(R[j], Bh[j][0], Bl[j][0]);
            if constexpr (TE) {
                // second k-step: time feature 4e + g at element e < 3 (feature 8 on lane group 0 only)
                float tf[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) tf[e] = (e < 3) ? time_feature(4 * e + g, A.time_mode, tq[j], mnorm[j]) : 0.0f;
                to_half8<SPLIT>(tf, Bh[j][1], Bl[j][1]);
            }
        }

        // --- mlp_base (32|41)-64-16 ---
        mlp_layer_h<BL::KS_B0, 4, NT, SPLIT>(whi + BL::B0 * kFragHalves, wlo + BL::B0 * kFragHalves, lane, Bh, Bl, D);
        to_operand_h<NT, SPLIT>(D, Bh, Bl);
        mlp_layer_h<2, 1, NT, SPLIT>(whi + BL::B1 * kFragHalves, wlo + BL::B1 * kFragHalves, lane, Bh, Bl, D);

        // accumulator row 4g + r: geometry feature 4g + r; row 15 (g = 3, r = 3): raw density
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const bool stored = 16 * j + c < in_tile;
            const int64_t s = tile_base + (16 * j + c);
            float sg = det_expf(D[j][0][3] - 1.0f);            // trunc_exp(raw - 1) * selector
            sg = sel[j] ? sg : 0.0f;
            if (g == 3 && stored) A.sigma[sbase + s] = sg;
            if (A.geo && stored) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * g + r < 15) A.geo[s * 15 + 4 * g + r] = D[j][0][r];
            }
        }

        if (A.want_rgb) {
            // --- head input [SH(4), geo(15)] (model.py:447-459): element 0 = SH_g, 1..4 = this lane's geo ---
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float dv[3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    dv[a] = A.rays_mode ? A.rays_d[3 * ridx[j] + a] : A.dir[3 * (tile_base + sofs[j]) + a];
                const float nrm = __builtin_sqrtf((dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2]);
                // lane group g feeds SH coefficient g: only that direction component is normalised (as field_kernel.hpp)
                const float comp = (g == 1) ? dv[1] : (g == 2) ? dv[2] : dv[0];
                const float u = (comp / nrm + 1.0f) / 2.0f;
                const float vv = u * 2.0f - 1.0f;
                const float coef = (g == 2) ? 0.48860251190291987f : -0.48860251190291987f;
                const float sh = (g == 0) ? 0.28209479177387814f : coef * vv;
                float hin[8];
                hin[0] = sh;
#pragma unroll
                for (int r = 0; r < 4; ++r) hin[1 + r] = __builtin_amdgcn_fmed3f(D[j][0][r], -kHalfMax, kHalfMax);
                hin[4] = (g == 3) ? 0.0f : hin[4];
                hin[5] = hin[6] = hin[7] = 0.0f;
                to_half8<SPLIT>(hin, Bh[j][0], Bl[j][0]);
            }
            mlp_layer_h<1, 4, NT, SPLIT>(whi + BL::H0 * kFragHalves, wlo + BL::H0 * kFragHalves, lane, Bh, Bl, D);
            to_operand_h<NT, SPLIT>(D, Bh, Bl);
            mlp_layer_h<2, 4, NT, SPLIT>(whi + BL::H1 * kFragHalves, wlo + BL::H1 * kFragHalves, lane, Bh, Bl, D);
            to_operand_h<NT, SPLIT>(D, Bh, Bl);
            mlp_layer_h<2, 1, NT, SPLIT>(whi + BL::H2 * kFragHalves, wlo + BL::H2 * kFragHalves, lane, Bh, Bl, D);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t s = tile_base + (16 * j + c);
                // the packer put colour channel a on accumulator row 4a = (lane group a, register 0): one sigmoid per lane
                const float o1 = 1.0f / (1.0f + det_expf(-D[j][0][0]));
                if (g < 3 && 16 * j + c < in_tile) A.rgb[3 * (sbase + s) + g] = o1;
            }
        }
    }
    // tracing only: every wave stamps its own end (waves of a workgroup finish up to a tile apart; a barrier here
    // would hold the early ones' registers and cost 3 % of throughput)
    if (A.stamp && lane == 0) atomicMax(A.stamp + 1, (unsigned long long)wall_clock64());
}

// Host: one layer into 16x16x32 A-fragment order.  Element (accumulator row p, operand position k) goes to
// fragment [frag + (p/16)*ks + k/32], lane 16*((k%32)/8) + p%16, half k%8; row_map says which neuron row p
// computes, col_map which input of the layer sits at position k (header comment of this file).
void pack_half_layer(const float *w, int n_out, int n_in, int nb, int ks, int frag, int row_map, int col_map,
                     _Float16 *hi, _Float16 *lo)
{
    for (int p = 0; p < nb * 16; ++p) {
        int neuron = p;
        if (row_map == HALF_ROW_HIDDEN) neuron = half_hidden_neuron(p);
        else if (row_map == HALF_ROW_BASE_OUT) neuron = half_base_out_neuron(p);
        else if (row_map == HALF_ROW_RGB) neuron = (p % 4 == 0) ? p / 4 : n_out;
        if (neuron >= n_out) continue;
        for (int k = 0; k < ks * 32; ++k) {
            const int g = (k % 32) / 8, e = k % 8;
            int in = k;
            if (col_map == HALF_COL_HASH) {
                if (k < 32) in = 2 * (4 * (e >> 1) + g) + (e & 1);          // level 4i + g, feature f at e = 2i + f
                else in = (e < 3 && 4 * e + g <= 8) ? 32 + 4 * e + g : -1;   // time feature 4e + g
            } else if (col_map == HALF_COL_HEAD) {
                if (e == 0) in = g;                                          // SH component g
                else if (e <= 4 && 4 * g + e - 1 < 15) in = 4 + 4 * g + e - 1;   // geometry feature 4g + e - 1
                else in = -1;
            }
            if (in < 0 || in >= n_in) continue;
            const float v = w[(int64_t)neuron * n_in + in];
            const _Float16 h = (_Float16)v;
            const int64_t idx = ((int64_t)(frag + (p / 16) * ks + k / 32) * 64 + 16 * g + (p % 16)) * 8 + e;
            hi[idx] = h;
            if (lo) lo[idx] = (_Float16)(v - (float)h);
        }
    }
}

static std::atomic<int> g_half_variant{ [] { const char *e = getenv("CED_HALF_VARIANT"); return e ? atoi(e) : 0; }() };
void set_half_variant(int v) { g_half_variant = v; }

int launch_field_half(FieldArgs &A, int time_mode, int precision, void *stream)
{
    auto launch = [&](auto kernel, int nt, int threads) {
        const int64_t n_tiles = (A.n + 16 * nt - 1) / (16 * nt);
        const int waves = threads / 64;
        int64_t blocks = A.spread_tiles ? (n_tiles + 3) / 4 : (n_tiles + waves - 1) / waves;
        const int cap = A.max_blocks > 0 ? A.max_blocks : kFieldBlocksDefault;
        if (blocks > cap) blocks = cap;                                   // one resident workgroup per CU, persistent over tiles
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(threads), 0, (hipStream_t)stream, A);
    };
    const int sel = (time_mode ? 1 : 0) | (A.table_dtype ? 2 : 0) | (A.temporal ? 4 : 0);
#define CED_HALF_CASE(SP_, NT_, TH_)                                                                            \
    switch (sel) {                                                                                              \
    case 0: launch(field_half_kernel<false, false, false, SP_, NT_, TH_>, NT_, TH_); break;                     \
    case 1: launch(field_half_kernel<true, false, false, SP_, NT_, TH_>, NT_, TH_); break;                      \
    case 2: launch(field_half_kernel<false, true, false, SP_, NT_, TH_>, NT_, TH_); break;                      \
    case 3: launch(field_half_kernel<true, true, false, SP_, NT_, TH_>, NT_, TH_); break;                       \
    case 4: launch(field_half_kernel<false, false, true, SP_, NT_, TH_>, NT_, TH_); break;                      \
    case 5: launch(field_half_kernel<true, false, true, SP_, NT_, TH_>, NT_, TH_); break;                       \
    case 6: launch(field_half_kernel<false, true, true, SP_, NT_, TH_>, NT_, TH_); break;                       \
    default: launch(field_half_kernel<true, true, true, SP_, NT_, TH_>, NT_, TH_); break;                       \
    }
    // Launch geometry ("half_variant": 0 = automatic, 1 = 2 x 512, 2 = 2 x 1024, 3 = 2 x 768 threads; env CED_HALF_AUTO_*
    // override the automatic picks for A/B runs).  Which kernels fit which register cap without scratch is printed by
    // tools/kernel_schedule.py; the automatic rule follows the round-4 measurements (profiles/r04_ab_half_geometry.txt).
    static const int auto_plain_split = [] { const char *e = getenv("CED_HALF_AUTO_F16X2"); return e ? atoi(e) : 2; }();
    static const int auto_plain = [] { const char *e = getenv("CED_HALF_AUTO_F16"); return e ? atoi(e) : 3; }();
    static const int auto_te = [] { const char *e = getenv("CED_HALF_AUTO_TE"); return e ? atoi(e) : 3; }();
    const int requested_variant = g_half_variant.load(std::memory_order_relaxed);
    int half_variant = requested_variant;
    if (half_variant == 0) {
        if (A.temporal) half_variant = 1;                    // temporal tables: 228-248 registers, scratch above 512 threads
        else if (time_mode) half_variant = auto_te;
        else half_variant = precision == CED_MLP_F16X2 ? auto_plain_split : auto_plain;
    }
    if (precision == CED_MLP_F16X2) {
        switch (half_variant) {
        case 1: CED_HALF_CASE(true, 2, 512) break;
        case 2: CED_HALF_CASE(true, 2, 1024) break;
#ifdef CED_AB_HALF_NT1
        case 4: CED_HALF_CASE(true, 1, 1024) break;
#endif
        default: CED_HALF_CASE(true, 2, 768) break;
        }
    } else {
        switch (half_variant) {
        case 1: CED_HALF_CASE(false, 2, 512) break;
        case 2: CED_HALF_CASE(false, 2, 1024) break;
#ifdef CED_AB_HALF_NT1
        case 4: CED_HALF_CASE(false, 1, 1024) break;
#endif
        default: CED_HALF_CASE(false, 2, 768) break;
        }
    }
#undef CED_HALF_CASE
    return check_launch("field_forward (half-precision MLP)");
}

}  // namespace ced

extern "C" int64_t ced_packed_weight_words(int use_div_offsets, int time_mode, int mlp_precision)
{
    if (mlp_precision == CED_MLP_F32 || mlp_precision == CED_MLP_F32_HEAD16X2) return ced_packed_weight_floats(use_div_offsets, time_mode);
    if (mlp_precision != CED_MLP_F16X2 && mlp_precision != CED_MLP_F16) return -1;
    const int64_t frags = time_mode ? ced::HalfBlob<true>::FRAGS : ced::HalfBlob<false>::FRAGS;
    return frags * (ced::kFragHalves / 2) * (mlp_precision == CED_MLP_F16X2 ? 2 : 1);
}

// Host-side reorder into 16x16x32 A-fragment order: element (accumulator row p, operand position k) of a
// layer goes to fragment [nb = p/16][ks = k/32], lane 16*((k%32)/8) + p%16, half k%8.  Which neuron row p
// computes and which input sits at position k are the layer's placements (header comment of this file).
extern "C" int ced_pack_field_weights_half(int use_div_offsets, int time_mode, int mlp_precision, const float *m_w0,
                                           const float *m_w1, const float *m_w2, const float *m_w3, const float *b_w0,
                                           const float *b_w1, const float *h_w0, const float *h_w1, const float *h_w2,
                                           void *out)
{
    CED_REQUIRE(m_w0 && m_w1 && m_w2 && m_w3 && b_w0 && b_w1 && h_w0 && h_w1 && h_w2 && out,
                "pack_field_weights_half: null pointer");
    CED_REQUIRE(time_mode >= 0 && time_mode <= 2, "pack_field_weights_half: time_mode=%d", time_mode);
    CED_REQUIRE(mlp_precision == CED_MLP_F16X2 || mlp_precision == CED_MLP_F16,
                "pack_field_weights_half: mlp_precision=%d (1 = f16x2, 2 = f16)", mlp_precision);
    const bool te = time_mode != 0, split = mlp_precision == CED_MLP_F16X2;
    const int64_t words = ced_packed_weight_words(use_div_offsets, time_mode, mlp_precision);
    memset(out, 0, (size_t)words * 4);
    _Float16 *hi = reinterpret_cast<_Float16 *>(out);
    _Float16 *lo = hi + (te ? ced::HalfBlob<true>::FRAGS : ced::HalfBlob<false>::FRAGS) * ced::kFragHalves;
    using namespace ced;
    struct L { const float *w; int n_out, n_in, nb, ks, frag, row, col; };
    const int base_in = te ? 41 : 32, n_mo = use_div_offsets ? 6 : 3, ksb0 = te ? 2 : 1;
    int fr[9];
    if (te) {
        using B = HalfBlob<true>;
        const int o[9] = { B::M0, B::M1, B::M2, B::M3, B::B0, B::B1, B::H0, B::H1, B::H2 };
        for (int i = 0; i < 9; ++i) fr[i] = o[i];
    } else {
        using B = HalfBlob<false>;
        const int o[9] = { B::M0, B::M1, B::M2, B::M3, B::B0, B::B1, B::H0, B::H1, B::H2 };
        for (int i = 0; i < 9; ++i) fr[i] = o[i];
    }
    const L layers[9] = {
        { m_w0, 64, 32, 4, 1, fr[0], HALF_ROW_HIDDEN, HALF_COL_NATURAL },   { m_w1, 64, 64, 4, 2, fr[1], HALF_ROW_HIDDEN, HALF_COL_NATURAL },
        { m_w2, 64, 64, 4, 2, fr[2], HALF_ROW_HIDDEN, HALF_COL_NATURAL },   { m_w3, n_mo, 64, 1, 2, fr[3], HALF_ROW_NATURAL, HALF_COL_NATURAL },
        { b_w0, 64, base_in, 4, ksb0, fr[4], HALF_ROW_HIDDEN, HALF_COL_HASH }, { b_w1, 16, 64, 1, 2, fr[5], HALF_ROW_BASE_OUT, HALF_COL_NATURAL },
        { h_w0, 64, 19, 4, 1, fr[6], HALF_ROW_HIDDEN, HALF_COL_HEAD },      { h_w1, 64, 64, 4, 2, fr[7], HALF_ROW_HIDDEN, HALF_COL_NATURAL },
        { h_w2, 3, 64, 1, 2, fr[8], HALF_ROW_RGB, HALF_COL_NATURAL },
    };
    for (const L &l : layers) pack_half_layer(l.w, l.n_out, l.n_in, l.nb, l.ks, l.frag, l.row, l.col, hi, split ? lo : nullptr);
    return CED_OK;
}
