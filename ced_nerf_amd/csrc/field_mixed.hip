// CED_MLP_F32_HEAD16X2: the fused field kernel with the sigma chain exact and the colour head on split-fp16 MFMAs.
//
// What a sample COUNT, an opacity or a depth depends on is sigma and t alone: Frequency encoding -> motion MLP ->
// normalise / selector -> hash gather -> [time encoding] -> mlp_base -> trunc_exp (cednerf/model.py:354-445).  Those
// stay the ascending-k fp32 FMA chains of field.hip (v_mfma_f32_16x16x4_f32), bit-identical to the CPU oracle -- so the
// image-global N_samples schedule of render_image_test, every ray's sample set, its termination plane, opacity and
// depth are those of the exact mode, bit for bit.  mlp_head (cednerf/model.py:447-466: [SH(4), geometry(15)] -> 64 -> 64
// -> 3 -> sigmoid) only feeds rgb, where the north-star allows 1e-4: its 100 of the kernel's 324 fp32 MFMAs per
// 16-sample tile (32 cycles each) become 42 fp16 K = 32 blocks (16 cycles each) on operands split into two fp16 numbers
// (hi + lo, 22 significant bits; hi*hi + hi*lo + lo*hi, fp32 accumulation) -- field_half.hip's head.
//
// Same template as the exact kernel (field_kernel.hpp, HEAD16); the packed blob keeps its size: the three head layers
// sit in their region as fp16 fragments (a plane of high parts, a plane of remainders).
#include <atomic>
#include <cstdlib>
#include <cstring>

#include "ced_common.hpp"
#include "field_args.hpp"
#include "field_device.hpp"
#include "field_kernel.hpp"

namespace ced {

// launch geometry (ced_set_option("mixed_variant")): 0 = 768 threads, 512 for the temporal-table kernels (which spill at
// three waves per SIMD); 1 = 512; 2 = 768
static std::atomic<int> g_mixed_variant{ [] { const char *e = getenv("CED_MIXED_VARIANT"); return e ? atoi(e) : 0; }() };
void set_mixed_variant(int v) { g_mixed_variant = v; }

int launch_field_mixed(FieldArgs &A, int time_mode, void *stream)
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
    // 768 threads (three waves per SIMD); the temporal-table kernels run 512 (see field.hip)
    const int mv = g_mixed_variant.load(std::memory_order_relaxed);
    const bool small = mv == 1 || (mv == 0 && A.temporal);
#define CED_MIXED_CASE(TH_)                                                                                    \
    switch (sel) {                                                                                              \
    case 0: launch(field_kernel<false, false, false, 2, TH_, true>, 2, TH_); break;                             \
    case 1: launch(field_kernel<true, false, false, 2, TH_, true>, 2, TH_); break;                              \
    case 2: launch(field_kernel<false, true, false, 2, TH_, true>, 2, TH_); break;                              \
    case 3: launch(field_kernel<true, true, false, 2, TH_, true>, 2, TH_); break;                               \
    case 4: launch(field_kernel<false, false, true, 2, TH_, true>, 2, TH_); break;                              \
    case 5: launch(field_kernel<true, false, true, 2, TH_, true>, 2, TH_); break;                               \
    case 6: launch(field_kernel<false, true, true, 2, TH_, true>, 2, TH_); break;                               \
    default: launch(field_kernel<true, true, true, 2, TH_, true>, 2, TH_); break;                               \
    }
    if (small) { CED_MIXED_CASE(512) } else { CED_MIXED_CASE(768) }
#undef CED_MIXED_CASE
    return check_launch("field_forward (fp32 sigma chain, split-fp16 colour head)");
}

}  // namespace ced

// Host packer of the CED_MLP_F32_HEAD16X2 blob: the six layers of the sigma chain in fp32 MFMA A-fragment order exactly
// as ced_pack_field_weights lays them out -- except that mlp_base's 16 output rows are placed for the K = 32 head
// operand (row p < 15 = geometry feature p, row 15 = raw density) -- then the three head layers as fp16 K = 32
// fragments, high parts and remainders.  Same size as the fp32 blob (ced_packed_weight_floats).
extern "C" int ced_pack_field_weights_mixed(int use_div_offsets, int time_mode, const float *m_w0, const float *m_w1,
                                            const float *m_w2, const float *m_w3, const float *b_w0, const float *b_w1,
                                            const float *h_w0, const float *h_w1, const float *h_w2, float *out)
{
    using namespace ced;
    CED_REQUIRE(m_w0 && m_w1 && m_w2 && m_w3 && b_w0 && b_w1 && h_w0 && h_w1 && h_w2 && out,
                "pack_field_weights_mixed: null pointer");
    CED_REQUIRE(time_mode >= 0 && time_mode <= 2, "pack_field_weights_mixed: time_mode=%d", time_mode);
    const bool te = time_mode != 0;
    // the fp32 part: the exact packer, then mlp_base's last layer again with the other row placement
    int rc = ced_pack_field_weights(use_div_offsets, time_mode, m_w0, m_w1, m_w2, m_w3, b_w0, b_w1, h_w0, h_w1, h_w2, out);
    if (rc) return rc;
    const int off_b1 = te ? Blob<true>::B1 : Blob<false>::B1;
    const int off_h0 = te ? Blob<true>::H0 : Blob<false>::H0;
    const int total = te ? Blob<true>::TOTAL : Blob<false>::TOTAL;
    for (int i = off_b1; i < total; ++i) out[i] = 0.0f;
    for (int p = 0; p < 16; ++p) {
        const int neuron = half_base_out_neuron(p);
        for (int k = 0; k < 64; ++k) {
            const int S = k / 4, kk = k % 4;
            const int lane = kk * 16 + p;
            out[off_b1 + ((int64_t)(S / 4) * 64 + lane) * 4 + (S % 4)] = b_w1[(int64_t)neuron * 64 + k];
        }
    }
    _Float16 *hi = reinterpret_cast<_Float16 *>(out + off_h0);
    _Float16 *lo = hi + Blob<false>::HEAD_FRAGS * kFragHalves;
    pack_half_layer(h_w0, 64, 19, 4, 1, Blob<false>::HF_H0, HALF_ROW_HIDDEN, HALF_COL_HEAD, hi, lo);
    pack_half_layer(h_w1, 64, 64, 4, 2, Blob<false>::HF_H1, HALF_ROW_HIDDEN, HALF_COL_NATURAL, hi, lo);
    pack_half_layer(h_w2, 3, 64, 1, 2, Blob<false>::HF_H2, HALF_ROW_RGB, HALF_COL_NATURAL, hi, lo);
    return CED_OK;
}
