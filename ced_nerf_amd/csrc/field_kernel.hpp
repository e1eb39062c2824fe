// The fused field kernel (template) and its packed-weight layout: included by field.hip (exact fp32 head) and
// field_mixed.hip (exact sigma chain, split-fp16 colour head).  See field.hip for the description of the kernel.
#pragma once
#include "ced_common.hpp"
#include "field_args.hpp"
#include "field_device.hpp"
#include "field_half_device.hpp"

namespace ced {

// ---- packed weight blob: layer l stored as [nb][ks4][lane 64][4] floats ----------------------
struct LayerShape { int nb; int ks; };
__host__ __device__ constexpr int ks4_of(int ks) { return (ks + 3) / 4; }
__host__ __device__ constexpr int layer_floats(int nb, int ks) { return nb * ks4_of(ks) * 256; }

// Output-row placement of mlp_base's last layer: accumulator row p = 4g + r (lane group g, register r)
// holds output neuron base_out_neuron(p).  Neuron 0 is the raw density, neuron n >= 1 is geometry feature
// n - 1 = input 3 + n of mlp_head (model.py:455), which must sit where k-step r, lane group g reads it.
__host__ __device__ constexpr int base_out_neuron(int p)
{
    const int g = p >> 2, r = p & 3;
    return r > 0 ? (4 * r + g - 3) : (g < 3 ? 13 + g : 0);
}

template <bool TE> struct Blob {
    static constexpr int KS_B0 = TE ? 11 : 8;
    static constexpr int M0 = 0;
    static constexpr int M1 = M0 + layer_floats(4, 8);
    static constexpr int M2 = M1 + layer_floats(4, 16);
    static constexpr int M3 = M2 + layer_floats(4, 16);
    static constexpr int B0 = M3 + layer_floats(1, 16);
    static constexpr int B1 = B0 + layer_floats(4, KS_B0);
    static constexpr int H0 = B1 + layer_floats(1, 16);
    static constexpr int H1 = H0 + layer_floats(4, 5);
    static constexpr int H2 = H1 + layer_floats(4, 16);
    static constexpr int TOTAL = H2 + layer_floats(1, 16);
    // HEAD16 kernels: the colour head's three layers sit in the same region as fp16 K = 32 fragments, a plane of the
    // high parts and a plane of the remainders (14 fragments each: H0 4, H1 8, H2 2) -- 28 KB, the size of the fp32 form
    static constexpr int HEAD_FRAGS = 14;
    static constexpr int HF_H0 = 0, HF_H1 = 4, HF_H2 = 12;
    static_assert(2 * HEAD_FRAGS * kFragHalves * 2 == (TOTAL - H0) * 4, "the fp16 head must fill the fp32 head's region");
};
constexpr int kMaxBlobFloats = Blob<true>::TOTAL;


// D[j][nb] (16 neurons x 16 samples, neurons 16nb+4g+r on lane group g reg r) =
//     sum_k W[neuron][k] * B[j][k/4] (k = 4S+g on lane group g), ascending k.
template <int KS, int NB, int NT>
__device__ __forceinline__ void mlp_layer(const float *__restrict__ wl, int lane, const float (&B)[NT][16],
                                          f4 (&D)[NT][4])
{
    constexpr int KS4 = ks4_of(KS);
    constexpr int G = NB * KS4;
    // the A fragment of group g + 1 (four k-steps of one 16-neuron block) is fetched from LDS before the MFMAs of group g
    // issue: fetched where it is used, every group began with the LDS latency exposed (as in field_half_device.hpp)
    f4 a = *reinterpret_cast<const f4 *>(wl + lane * 4);
    f4 acc[NT];
#pragma unroll
    for (int grp = 0; grp < G; ++grp) {
        const int nb = grp / KS4, q = grp % KS4;
        if (q == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
        f4 a_next = a;
        if (grp + 1 < G) a_next = *reinterpret_cast<const f4 *>(wl + ((grp + 1) * 64 + lane) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (4 * q + s < KS) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], B[j][4 * q + s], acc[j], 0, 0, 0);
            }
        }
        if (q == KS4 - 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j) D[j][nb] = acc[j];
        }
        a = a_next;
    }
}

// ReLU (optional) on the accumulator blocks, which then serve as the next layer's B operand.
template <int NB, bool RELU, int NT>
__device__ __forceinline__ void to_operand(const f4 (&D)[NT][4], float (&B)[NT][16])
{
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float r[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = D[j][nb][q];
#ifdef CED_FIELD_SKELETON
                if constexpr (false) {
#else
                if constexpr (RELU) {
#endif
                    // ReLU as ONE integer max on the float's bits: non-negative floats order like their bit
                    // patterns, every negative float (and -0) has the sign bit set, i.e. a negative int.
                    // Same result as (v > 0 ? v : 0) for every non-NaN v; the float forms (fmax, compare +
                    // select, med3) all lower to two v_max_f32, the first one only canonicalising.
                    const int bits = __float_as_int(v);
                    v = __int_as_float(bits > 0 ? bits : 0);
                }
                r[q] = v;
            }
            // no lane movement: the host's row placement makes register r of block nb the operand of
            // k-step 4nb + r (see ced_pack_field_weights)
#pragma unroll
            for (int s = 0; s < 4; ++s) B[j][4 * nb + s] = r[s];
        }
    }
}

// NT: 16-sample MFMA column tiles per wave iteration; THREADS: workgroup size (one workgroup per CU).
// HEAD16 (CED_MLP_F32_HEAD16X2): Frequency -> motion MLP -> hash -> mlp_base -> exp -- everything a sample COUNT, an
// opacity or a depth depends on -- stays the exact fp32 chain, bit for bit; only mlp_head (SH + geometry features ->
// colour, cednerf/model.py:447-466), a third of the kernel's MFMA time, runs on split-fp16 operands (hi + lo, three fp16
// MFMA blocks per product block, fp32 accumulation), whose error reaches rgb alone and stays far below the 1e-4 allowed
// there.  mlp_base's output rows are then placed for the K = 32 head operand (row 4g + r = geometry feature 4g + r,
// row 15 = the raw density; half_base_out_neuron).
template <bool TE, bool F16, bool TEMPORAL, int NT, int THREADS, bool HEAD16 = false>
__global__ __launch_bounds__(THREADS) void field_kernel(FieldArgs A)
{
    constexpr int FIELD_THREADS = THREADS;
    constexpr int FIELD_WAVES = THREADS / kWave;
    constexpr int TILE = 16 * NT;
    using BL = Blob<TE>;
    __shared__ __attribute__((aligned(16))) float lds[BL::TOTAL + 8 * CED_MAX_LEVELS];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
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
    // a workgroup without a tile leaves before staging anything (launches are sized by a host-side upper bound of
    // the sample count; the exact count comes from device memory); tile -> wave mapping: field_device.hpp
    TileRange tiles;
    if (!field_tile_range(A.spread_tiles, n_tiles, FIELD_WAVES, threadIdx.x >> 6, tiles)) return;
    if (A.stamp && tid == 0) atomicMin(A.stamp, (unsigned long long)wall_clock64());

    // stage weights + level tables into LDS
    {
        const f4 *src = reinterpret_cast<const f4 *>(A.weights);
        f4 *dst = reinterpret_cast<f4 *>(lds);
        for (int i = tid; i < BL::TOTAL / 4; i += FIELD_THREADS) dst[i] = src[i];
        if (tid < CED_MAX_LEVELS) {
            uint32_t *lt = reinterpret_cast<uint32_t *>(lds + BL::TOTAL);
            const LevelConst L = make_level(A.scale[tid], A.res[tid], A.offset[tid], A.size[tid], A.hashed[tid],
                                            EntryBytes<F16, TEMPORAL>::value);
            store_level(lt + tid * 8, L);
        }
    }
    __syncthreads();

    if (A.stagger > 0) {
        // Waves w, w+4, w+8 of a workgroup share a SIMD and run the same program: offset their phases so
        // that their MFMA-dense and VALU-dense stretches interleave instead of colliding.
        const int slot = __builtin_amdgcn_readfirstlane(wave >> 2);
        for (int k = 0; k < slot * A.stagger; ++k) __builtin_amdgcn_s_sleep(127);
    }
    const float extent[3] = { A.aabb[3] - A.aabb[0], A.aabb[4] - A.aabb[1], A.aabb[5] - A.aabb[2] };
    // In eval frames every sample carries the same timestamp (cednerf/utils.py:186-193): the two
    // Frequency features of t that this lane feeds to the motion MLP are computed once.
    const bool shared_time = A.rays_mode && !A.t_per_ray;
    float t_feat[2] = { 0.0f, 0.0f };
    if (shared_time) {
        const float t_all = A.timestamps[0];
#pragma unroll
        for (int S = 6; S < 8; ++S) t_feat[S - 6] = det_sinpi_phase(t_all * (float)(1 << (2 * (S & 1) + (g >> 1))), g & 1);
    }

    for (int64_t tile = tiles.first; tile < tiles.end; tile += tiles.stride) {
        // Re-derive the LDS weight base every tile through an opaque register: the A fragments sit at
        // tile-invariant addresses and the compiler would otherwise hoist all ~80 ds_read_b128 out of
        // the persistent loop and park them in scratch.
        int lds_off = 0;
        asm volatile("" : "+v"(lds_off));
        const float *const lw = lds + lds_off;
        int64_t sidx[NT];
        int64_t ridx[NT];
        float px[NT][3], tq[NT];
        bool any_used = !A.rays_mode;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            int64_t s = tile * TILE + 16 * j + c;
            s = s < n_eff ? s : n_eff - 1;
            sidx[j] = s;
            if (A.rays_mode) {
                // a negative ray index marks an unused sample slot (the frame renderer's slot-major sample layout):
                // it is evaluated on ray 0 at t = 0 and its outputs land in its own, never-read slot
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

        float B[NT][16];
        f4 D[NT][4];

        // --- tcnn Frequency(4) on (x,y,z,t): feature k = 8*dim + 2*freq + phase, k = 4S+g, i.e. k-step S of lane group g
        // is dimension S>>1, frequency 2(S&1) + (g>>1), phase g&1.  Lane groups g and g^1 need the same terms in the two
        // phases, and one evaluation yields both (det_sinpi_both): the even group evaluates one k-step of a pair, the
        // odd group the other, and a single v_permlane16_swap hands each its partner's half -- swap(p0, p1) leaves the
        // even group's k-step in the first register and the odd group's in the second, each lane seeing its own phase.
        // Pairs: (S0 | S2) and (S1 | S3) = x | y at the two frequencies, (S4 | S5) = z at both, (S6 | S7) = t at both. ---
#pragma unroll
        for (int j = 0; j < NT; ++j) {
#ifdef CED_FIELD_SKELETON
#pragma unroll
            for (int S = 0; S < 8; ++S) B[j][S] = (S < 6 ? px[j][S >> 1] : tq[j]) * (float)(1 << (2 * (S & 1) + (g >> 1)));
#elif defined(CED_AB_NO_FREQ_SPLIT)
#pragma unroll
            for (int S = 0; S < 8; ++S) {
                if ((S >> 1) == 3 && shared_time) { B[j][S] = t_feat[S - 6]; continue; }
                const float v = ((S >> 1) < 3) ? px[j][(S >> 1) < 3 ? (S >> 1) : 0] : tq[j];
                B[j][S] = det_sinpi_phase(v * (float)(1 << (2 * (S & 1) + (g >> 1))), g & 1);
            }
#else
            const bool odd = (g & 1) != 0;
            const float sc0 = (float)(1 << (g >> 1)), sc1 = 4.0f * sc0;           // 2^f for the pair's two frequencies
            const float vxy = odd ? px[j][1] : px[j][0];
            float p0, p1;
            det_sinpi_both(vxy * sc0, p0, p1);
            auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
            B[j][0] = __uint_as_float(sw[0]); B[j][2] = __uint_as_float(sw[1]);
            det_sinpi_both(vxy * sc1, p0, p1);
            sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
            B[j][1] = __uint_as_float(sw[0]); B[j][3] = __uint_as_float(sw[1]);
            const float scz = odd ? sc1 : sc0;
            det_sinpi_both(px[j][2] * scz, p0, p1);
            sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
            B[j][4] = __uint_as_float(sw[0]); B[j][5] = __uint_as_float(sw[1]);
            if (shared_time) {                           // eval frames: one timestamp for every sample
                B[j][6] = t_feat[0];
                B[j][7] = t_feat[1];
            } else {
                det_sinpi_both(tq[j] * scz, p0, p1);
                sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(p0), __float_as_uint(p1), false, false);
                B[j][6] = __uint_as_float(sw[0]); B[j][7] = __uint_as_float(sw[1]);
            }
#endif
        }
        // --- motion MLP 32-64-64-64-(3|6) ---
        mlp_layer<8, 4, NT>(lw + BL::M0, lane, B, D);
        to_operand<4, true, NT>(D, B);
        mlp_layer<16, 4, NT>(lw + BL::M1, lane, B, D);
        to_operand<4, true, NT>(D, B);
        mlp_layer<16, 4, NT>(lw + BL::M2, lane, B, D);
        to_operand<4, true, NT>(D, B);
        mlp_layer<16, 1, NT>(lw + BL::M3, lane, B, D);

        // --- query_move / normalise / selector (model.py:354-383) ---
        float xn[NT][3], mnorm[NT];
        bool sel[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float mv[3];
            bool inside = true;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float off = __shfl(D[j][0][a], c, 64);          // row a lives on lane group 0, reg a
                float m = off * A.moving_step;
                if (A.use_div) {
                    // rows 3,4,5: (g0,r3), (g1,r0), (g1,r1)
                    constexpr int kFineReg[3] = { 3, 0, 1 };
                    const float fine = __shfl(D[j][0][kFineReg[a]], (a == 0) ? c : 16 + c, 64);
                    const float e = det_expf(2.0f * fine);
                    const float th = 1.0f - 2.0f / (e + 1.0f);
                    m = m + th * A.moving_step;
                }
                mv[a] = m;
                const float xm = px[j][a] + m;
                const float x = (xm - A.aabb[a]) / extent[a];
                inside = inside && (x > 0.0f && x < 1.0f);
                xn[j][a] = __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f);
            }
            sel[j] = inside;
            mnorm[j] = TE ? __builtin_sqrtf((mv[0] * mv[0] + mv[1] * mv[1]) + mv[2] * mv[2]) : 0.0f;
        }

        // --- hash gather: this lane's 4 levels for each of its samples, then into operand order ---
        float R[NT][8];
        int k_lo[NT];
        float t_frac[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            k_lo[j] = 0;
            t_frac[j] = 0.0f;
            if constexpr (TEMPORAL) temporal_keyframe(tq[j], k_lo[j], t_frac[j]);
        }
        const uint32_t *const ltab = reinterpret_cast<const uint32_t *>(lw + BL::TOTAL);
#ifdef CED_FIELD_SKELETON   // diagnostic build: MFMA skeleton only (results are meaningless)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) R[j][i] = xn[j][i % 3] + (float)i;
#else
        // gather slot i of lane group g is level 4i + 2(g&1) + (g>>1): slot i spans levels 4i..4i+3 across the
        // wave, and when those are all dense or all hashed (wave-uniform, decided on the host) only that index
        // form is computed.  The level's constants come from LDS here rather than living in registers.
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const LevelConst L = load_level(ltab + (4 * i + 2 * (g & 1) + (g >> 1)) * 8);
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
#endif
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // Slot i holds (f0, f1) of level 4i + h on the even lane group 2h (k-step 2i) and of level
            // 4i + 2 + h on the odd group 2h+1 (k-step 2i+1).  Operand element (k-step S, group g) is
            // feature g&1 of level 2S + (g>>1): swapping the odd rows of the f0 register with the even rows
            // of the f1 register leaves k-step 2i in the first and k-step 2i+1 in the second.
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#ifndef CED_FIELD_SKELETON
                auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(R[j][2 * i]), __float_as_uint(R[j][2 * i + 1]), false, false);
                B[j][2 * i] = __uint_as_float(sw[0]);
                B[j][2 * i + 1] = __uint_as_float(sw[1]);
#else
                B[j][2 * i] = R[j][2 * i];
                B[j][2 * i + 1] = R[j][2 * i + 1];
#endif
            }
            if (TE) {
#pragma unroll
                for (int S = 8; S < 11; ++S) B[j][S] = time_feature(4 * (S - 8) + g, A.time_mode, tq[j], mnorm[j]);
            }
        }

        // --- mlp_base (32|41)-64-16; output row placement: see base_out_neuron() ---
        mlp_layer<BL::KS_B0, 4, NT>(lw + BL::B0, lane, B, D);
        to_operand<4, true, NT>(D, B);
        mlp_layer<16, 1, NT>(lw + BL::B1, lane, B, D);

#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int64_t s = tile * TILE + 16 * j + c;
            // density = trunc_exp(raw - 1) * selector; raw: lane group 3, register 0 (HEAD16: register 3)
            float sg = det_expf(D[j][0][HEAD16 ? 3 : 0] - 1.0f);
            sg = sel[j] ? sg : 0.0f;
            if (g == 3 && s < n_eff) A.sigma[sbase + s] = sg;
            if (A.geo && s < n_eff) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int nidx = HEAD16 ? half_base_out_neuron(4 * g + r) : base_out_neuron(4 * g + r);
                    if (nidx >= 1) A.geo[s * 15 + nidx - 1] = D[j][0][r];
                }
            }
        }

        if constexpr (HEAD16) {
            if (A.want_rgb) {
                // --- colour head on split-fp16 MFMAs (field_half.hip's head, SPLIT): operand element 0 = SH_g, 1..4 = this
                // lane's four mlp_base outputs (geometry features 4g + r; the raw density of group 3 is masked) ---
                const _Float16 *const hw = reinterpret_cast<const _Float16 *>(lw + BL::H0);
                const _Float16 *const hwl = hw + BL::HEAD_FRAGS * kFragHalves;
                h8 Bh[NT][2], Bl[NT][2];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    float dv[3];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
                        dv[a] = A.rays_mode ? A.rays_d[3 * ridx[j] + a] : A.dir[3 * sidx[j] + a];
                    const float nrm = __builtin_sqrtf((dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2]);
                    const float comp = (g == 1) ? dv[1] : (g == 2) ? dv[2] : dv[0];
                    const float u = (comp / nrm + 1.0f) / 2.0f;
                    const float vv = u * 2.0f - 1.0f;
                    const float coef = (g == 2) ? 0.48860251190291987f : -0.48860251190291987f;
                    float hin[8];
                    hin[0] = (g == 0) ? 0.28209479177387814f : coef * vv;
#pragma unroll
                    for (int r = 0; r < 4; ++r) hin[1 + r] = __builtin_amdgcn_fmed3f(D[j][0][r], -kHalfMax, kHalfMax);
                    hin[4] = (g == 3) ? 0.0f : hin[4];
                    hin[5] = hin[6] = hin[7] = 0.0f;
                    to_half8<true>(hin, Bh[j][0], Bl[j][0]);
                }
                mlp_layer_h<1, 4, NT, true>(hw + BL::HF_H0 * kFragHalves, hwl + BL::HF_H0 * kFragHalves, lane, Bh, Bl, D);
                to_operand_h<NT, true>(D, Bh, Bl);
                mlp_layer_h<2, 4, NT, true>(hw + BL::HF_H1 * kFragHalves, hwl + BL::HF_H1 * kFragHalves, lane, Bh, Bl, D);
                to_operand_h<NT, true>(D, Bh, Bl);
                mlp_layer_h<2, 1, NT, true>(hw + BL::HF_H2 * kFragHalves, hwl + BL::HF_H2 * kFragHalves, lane, Bh, Bl, D);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    // the packer put colour channel a on accumulator row 4a = (lane group a, register 0)
                    const float o1 = 1.0f / (1.0f + det_expf(-D[j][0][0]));
                    int lane_now = (int)threadIdx.x;
                    asm volatile("" : "+v"(lane_now));
                    const int g_now = (lane_now >> 4) & 3;
                    const int64_t s_now = tile * TILE + 16 * j + (lane_now & 15);
                    if (g_now < 3 && s_now < n_eff) A.rgb[3 * (sbase + s_now) + g_now] = o1;
                }
            }
            continue;
        }

        if (A.want_rgb) {
            // --- head input: [SH(4), geo(15)] (model.py:447-459); k = 4S+g.  The base layer's output rows
            // were placed so that register r of lane group g is head input 4r + g (r = 1..3) and
            // 16 + g (r = 0, g < 3); row (g = 3, r = 0) is the raw density, masked out here. ---
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const float geo_tail = (g == 3) ? 0.0f : D[j][0][0];
                float dv[3];
#pragma unroll
                for (int a = 0; a < 3; ++a)
                    dv[a] = A.rays_mode ? A.rays_d[3 * ridx[j] + a] : A.dir[3 * sidx[j] + a];
                const float nrm = __builtin_sqrtf((dv[0] * dv[0] + dv[1] * dv[1]) + dv[2] * dv[2]);
                // lane group g feeds SH coefficient g: only that one direction component is normalised here
                // (Y00 const, Y1-1 ~ -y, Y10 ~ z, Y11 ~ -x; tcnn maps the unit vector to [0,1] and back)
                const float comp = (g == 1) ? dv[1] : (g == 2) ? dv[2] : dv[0];
                const float u = (comp / nrm + 1.0f) / 2.0f;
                const float vv = u * 2.0f - 1.0f;
                const float coef = (g == 2) ? 0.48860251190291987f : -0.48860251190291987f;
                B[j][0] = (g == 0) ? 0.28209479177387814f : coef * vv;
                B[j][1] = D[j][0][1];
                B[j][2] = D[j][0][2];
                B[j][3] = D[j][0][3];
                B[j][4] = geo_tail;
            }
            mlp_layer<5, 4, NT>(lw + BL::H0, lane, B, D);
            to_operand<4, true, NT>(D, B);
            mlp_layer<16, 4, NT>(lw + BL::H1, lane, B, D);
            to_operand<4, true, NT>(D, B);
            mlp_layer<16, 1, NT>(lw + BL::H2, lane, B, D);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int64_t s = tile * TILE + 16 * j + c;
                // the packer put colour channel a on accumulator row 4a = (lane group a, register 0): every lane
                // evaluates ONE sigmoid (its group's channel) instead of three of which only group 0's were kept
                const float o1 = 1.0f / (1.0f + det_expf(-D[j][0][0]));
                // (re-derived from the lane id here rather than kept live across the tile: the kernel sits at the
                //  168-register limit of three waves per SIMD)
                int lane_now = (int)threadIdx.x;
                asm volatile("" : "+v"(lane_now));
                const int g_now = (lane_now >> 4) & 3;
                const int64_t s_now = tile * TILE + 16 * j + (lane_now & 15);
                if (g_now < 3 && s_now < n_eff) A.rgb[3 * (sbase + s_now) + g_now] = o1;
            }
        }
    }
    // tracing only: every wave stamps its own end (waves of a workgroup finish up to a tile apart; a barrier here
    // would hold the early ones' registers and cost 3 % of throughput)
    if (A.stamp && lane == 0) atomicMax(A.stamp + 1, (unsigned long long)wall_clock64());
}

}  // namespace ced
