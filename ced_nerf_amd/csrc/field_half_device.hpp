// Device-side pieces of the half-precision-MFMA MLP layers (fp16 operands, fp32 accumulate) of field_half.hip:
// operand packing, the layer loop with its operand guard, the accumulator -> operand step, and the
// host/device row maps.
#pragma once
#include <cstdint>

#include "ced_common.hpp"
#include "field_device.hpp"

namespace ced {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int kFragHalves = 512;      // one (nb, ks) A fragment: 64 lanes x 8 halves

// accumulator row -> neuron of a 64-wide hidden layer (see the header comment)
__host__ __device__ constexpr int half_hidden_neuron(int p)
{
    const int nb = p >> 4, g = (p >> 2) & 3, r = p & 3;
    return 32 * (nb >> 1) + 8 * g + 4 * (nb & 1) + r;
}
// mlp_base output: row p < 15 is geometry feature p (neuron 1 + p), row 15 the raw density (neuron 0)
__host__ __device__ constexpr int half_base_out_neuron(int p) { return p < 15 ? p + 1 : 0; }

constexpr float kHalfMax = 65504.0f;

#ifndef CED_HALF_MFMA_GUARD
#define CED_HALF_MFMA_GUARD 3
#endif

// Elementwise math of the half-precision kernels: since round 4 the SAME deterministic forms as the exact kernel
// (det_expf, IEEE division and square root).  With the matrix instruction's own summation restated on the CPU
// (oracle/mfma_f16_model.h) that makes these modes bit-comparable with the oracle's fp16-operand modes: sigma,
// hence every sample count, opacity and depth -- and rgb.  (Rounds 1-3 used the hardware's 1-ulp exp2 / reciprocal /
// rsqrt here: 2-3 % faster, not reproducible on a CPU.)

// eight fp32 values -> packed fp16 operand (and the fp16 remainder in F16X2 mode).  Two values at a time: ONE packed
// conversion gives both high parts (v_cvt_pk_f16_f32, round to nearest even like the scalar form).  The remainder
// lo = fp16(x - fp32(hi)) is ONE instruction per value, v_fma_mixlo_f16 / v_fma_mixhi_f16 computing fp16(hi * -1.0 + x)
// straight from the packed high part: x - hi is exact in fp32 (hi is x rounded to 11 bits), so rounding the fused
// result once to fp16 is rounding the same real number as the two-step form the oracle states
// (tools/probes/split_mix_exhaustive.hip: all 2^32 patterns agree).  Per value 1.5 vector instructions for the split
// (+ 1 for the ReLU / saturation of to_operand_h); rounds 1-3: 4, round 4 before this: 2.5.
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));
template <bool SPLIT> __device__ __forceinline__ void to_half8(const float (&v)[8], h8 &hi, h8 &lo)
{
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        // The values are made opaque first: left visible, hipcc folds a producing multiply / fma into the conversion
        // (v_fma_mixlo_f16 with a multiplier other than 1), which rounds ONCE to fp16 instead of to fp32 and then to
        // fp16 -- a different result whenever the fp32 value is an fp16 tie (found in round 4: the SH inputs of the f16
        // kernels, 1 value in 8192).  tools/isa_lint.py allows the mix forms only in the shape of the remainder below.
        float x0 = v[e], x1 = v[e + 1];
        asm("" : "+v"(x0));
        asm("" : "+v"(x1));
        const h2 h = __builtin_convertvector(f2v{ x0, x1 }, h2);
        hi[e] = h[0];
        hi[e + 1] = h[1];
        if constexpr (SPLIT) {
            // fptrunc(fma(fpext(hi), m1, x)) is the pattern hipcc selects v_fma_mix{lo,hi}_f16 for; m1 = -1.0 is kept
            // opaque in a scalar register, or the middle end turns the fma into a subtraction (three instructions again).
            // NOT inline assembly: the hazard recogniser does not see through it, and an MFMA that reads the remainder
            // less than two wait states after the write takes the stale register (seen with the fp16-table kernels).
            float m1 = -1.0f;
            asm("" : "+s"(m1));
            lo[e] = (_Float16)__builtin_fmaf((float)h[0], m1, x0);
            lo[e + 1] = (_Float16)__builtin_fmaf((float)h[1], m1, x1);
        }
    }
}

typedef _Float16 h4 __attribute__((ext_vector_type(4)));

// One K = 32 product block of D^T = W * X^T.  Default: TWO v_mfma_f32_16x16x16_f16 over the low / high four halves
// of each lane's operands.  -DCED_HALF_MFMA_K32 (build-time opt-in, CED_HALF_MFMA_K32=1 in the environment of
// _lib.build): gfx950's single v_mfma_f32_16x16x32_f16 (f16x2: 4.2 instead of 3.8 Gsamples/s; f16: 5.2 either way).
//
// Why the pair is the default (hazard found in round 2): while a wave of a SIMD executes
// v_mfma_f32_16x16x32_f16, a packed-fp32 VALU instruction of ANOTHER wave of the SIMD whose op_sel takes the HIGH
// half of src1 for the low result lane (v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[0,1], v_pk_fma_f32 ... op_sel:[0,1,0])
// reads that operand as zero, about once in 1e4 executions (tools/probes/pk_opsel_mfma.hip isolates it: never without
// the MFMA, never beside v_mfma_f32_16x16x16_f16, never for un-swizzled, op_sel_hi or src0/src2 swizzles).  This
// library is built with -fno-slp-vectorize (hipcc's SLP vectoriser emits that form) and tools/isa_lint.py rejects any
// kernel of it that contains the form -- but kernels of OTHER code objects (torch element-wise / gather kernels, RCCL)
// run on other streams beside the field kernels of a pipelined renderer, are compiled with SLP, and cannot be
// linted here: they would be corrupted silently.  The K = 32 form is therefore only for processes in which nothing
// foreign can be co-resident with a half-precision field kernel.
__device__ __forceinline__ f4 mfma_k32(const h8 &a, const h8 &b, f4 c)
{
#ifdef CED_HALF_MFMA_K32
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#else
    const h4 a0 = { a[0], a[1], a[2], a[3] }, a1 = { a[4], a[5], a[6], a[7] };
    const h4 b0 = { b[0], b[1], b[2], b[3] }, b1 = { b[4], b[5], b[6], b[7] };
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, c, 0, 0, 0);
#endif
}

// One layer: D^T[nb] = sum over k-steps of W[nb][ks] * X^T[ks], NB * KS groups of (SPLIT ? 3 : 1) * NT product blocks.
// The A fragments of group g + 1 are fetched from LDS BEFORE the MFMAs of group g issue (two fragment pairs alive: 8
// registers more): fetched where they are used -- rounds 1-4 -- every group began with ~100 cycles of exposed LDS latency,
// once per 12 MFMAs (192 cycles of the matrix pipe).
template <int KS, int NB, int NT, bool SPLIT>
__device__ __forceinline__ void mlp_layer_h(const _Float16 *__restrict__ whi, const _Float16 *__restrict__ wlo, int lane,
                                            const h8 (&Bh)[NT][2], const h8 (&Bl)[NT][2], f4 (&D)[NT][4])
{
    constexpr int G = NB * KS;
    h8 ah = *reinterpret_cast<const h8 *>(whi + lane * 8);
    h8 al = ah;
    if constexpr (SPLIT) al = *reinterpret_cast<const h8 *>(wlo + lane * 8);
    f4 acc[NT];
#pragma unroll
    for (int grp = 0; grp < G; ++grp) {
        const int nb = grp / KS, ks = grp % KS;
        if (ks == 0) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
        }
        h8 ah_next = ah, al_next = al;
        if (grp + 1 < G) {
            ah_next = *reinterpret_cast<const h8 *>(whi + ((grp + 1) * 64 + lane) * 8);
            if constexpr (SPLIT) al_next = *reinterpret_cast<const h8 *>(wlo + ((grp + 1) * 64 + lane) * 8);
            else al_next = ah_next;
        }
        if constexpr (SPLIT) {
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = mfma_k32(al, Bh[j][ks], acc[j]);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = mfma_k32(ah, Bl[j][ks], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = mfma_k32(ah, Bh[j][ks], acc[j]);
        // Group fence: the MFMAs of a group issue back to back; it keeps the conversions of the next operands
        // out of the MFMA stream (the operand pins are left-overs of round 1's hunt for the hazard described at
        // mfma_k32 and cost nothing measurable).
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::"v"(ah));
        if constexpr (SPLIT) asm volatile("" ::"v"(al));
        __builtin_amdgcn_sched_barrier(0);
        if (ks == KS - 1) {
#pragma unroll
            for (int j = 0; j < NT; ++j) D[j][nb] = acc[j];
        }
        ah = ah_next;
        al = al_next;
    }
    // the same for the B operands, whose last readers are the MFMAs of the last group
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            asm volatile("" ::"v"(Bh[j][ks]));
            if constexpr (SPLIT) asm volatile("" ::"v"(Bl[j][ks]));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ReLU + saturation to the fp16 range (one v_med3_f32), then the accumulator registers become the
// next layer's operand: k-step ks takes D[2ks][0..3], D[2ks+1][0..3].
template <int NT, bool SPLIT>
__device__ __forceinline__ void to_operand_h(const f4 (&D)[NT][4], h8 (&Bh)[NT][2], h8 (&Bl)[NT][2])
{
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = __builtin_amdgcn_fmed3f(D[j][2 * ks + (e >> 2)][e & 3], 0.0f, kHalfMax);
            to_half8<SPLIT>(v, Bh[j][ks], Bl[j][ks]);
        }
    }
}

// host: one layer W[n_out][n_in] into 16x16x32 A-fragment order at fragment `frag` (see field_half.hip)
// HALF_ROW_RGB: colour channel a on accumulator row 4a = (lane group a, register 0): one sigmoid per lane
enum HalfRowMap { HALF_ROW_NATURAL, HALF_ROW_HIDDEN, HALF_ROW_BASE_OUT, HALF_ROW_RGB };
enum HalfColMap { HALF_COL_NATURAL, HALF_COL_HASH, HALF_COL_HEAD };
void pack_half_layer(const float *w, int n_out, int n_in, int nb, int ks, int frag, int row_map, int col_map,
                     _Float16 *hi, _Float16 *lo);

}  // namespace ced
