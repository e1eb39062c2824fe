// A whole bias-free ReLU MLP of the training path in ONE launch per direction (the fused forward /
// backward of the tiny-cuda-nn FullyFusedMLP networks of cednerf/model.py:200-222,280-344 that the reference trains
// through, train_real.py:339-420):
//   forward    a_0 = x;  a_l = act(a_{l-1} W_l^T), l = 1..L   (ReLU between layers, the last layer linear)
//              every a_l is written once (the backward needs it); it is never read back
//   backward   g_L = dy;  g_{l-1} = (g_l W_l) * [a_{l-1} > 0]   (the input gradient g_0 without a mask)
//              every g_l is written once (the weight gradient dW_l = g_l^T a_{l-1} is ced_weight_grad's)
// Layer-at-a-time launches (ced_linear) re-read every activation; here a wave keeps the 32 samples of its tile in
// registers from the first layer to the last.  That works without any data movement because ced_linear's fragment
// geometry closes on itself: lane (g, c) ends a layer holding outputs 16 nb + 4 g + r of sample c, and as B operand of
// the next layer it must supply inputs 16 q + 4 g + s -- the same elements (nb = q, r = s).  Same MFMA order per
// output as ced_linear => the same bits.  Widths <= 64, <= 6 layers; all layers' fragments live in LDS (<= 96 KB).
#include <cstdlib>

#include "ced_common.hpp"

namespace ced {

typedef float mf4 __attribute__((ext_vector_type(4)));

constexpr int kMlpMaxLayers = 6;

struct MlpArgs {
    int64_t n;
    int n_layers, backward;
    const float *x;                       // forward: input [n, width[0]]; backward: dy [n, width[L]]
    int width[kMlpMaxLayers + 1];         // width[0] = input width, width[l] = output width of layer l (1-based)
    const float *w[kMlpMaxLayers];        // W_l [width[l], width[l-1]] row-major (layer l = index l-1)
    float *out[kMlpMaxLayers];            // forward: a_l [n, width[l]] (index l-1); backward: g_{l-1} [n, width[l-1]] (index l-1), may be NULL
    const float *mask[kMlpMaxLayers];     // backward: a_{l-1} for l >= 2 (index l-1); NULL = no mask
    int relu_last;
};

// fragment order of one layer's matrix M [N, K] (forward: M = W; backward: M = W^T): [nb][q][lane = kk*16 + row][s]
// = M[16nb + row][16q + 4kk + s], zero beyond N / K  (ced_linear's, linear.hip)
__device__ __forceinline__ void stage_layer(float *frag, const float *m, int64_t so, int64_t si, int N, int K, int tid, int threads)
{
    const int NB = (N + 15) / 16, KQ = (K + 15) / 16;
    for (int e = tid; e < NB * KQ * 256; e += threads) {
        const int s = e & 3, ln = (e >> 2) & 63, q = (e >> 8) % KQ, nb = (e >> 8) / KQ;
        const int row = 16 * nb + (ln & 15), k = 16 * q + 4 * (ln >> 4) + s;
        frag[e] = (row < N && k < K) ? m[row * so + k * si] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void mlp_chain_kernel(MlpArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float frag[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int L = A.n_layers;
    // layer order of the walk: forward 1..L, backward L..1; step t handles layer lay(t)
    int off[kMlpMaxLayers + 1];
    off[0] = 0;
#pragma unroll
    for (int t = 0; t < kMlpMaxLayers; ++t) {
        if (t < L) {
            const int l = A.backward ? L - 1 - t : t;                    // 0-based layer
            const int N = A.backward ? A.width[l] : A.width[l + 1], K = A.backward ? A.width[l + 1] : A.width[l];
            if (A.backward) stage_layer(frag + off[t], A.w[l], 1, A.width[l], N, K, tid, 256);      // M = W^T: M[o][i] = W[i][o]
            else stage_layer(frag + off[t], A.w[l], A.width[l], 1, N, K, tid, 256);
            off[t + 1] = off[t] + ((N + 15) / 16) * ((K + 15) / 16) * 256;
        } else {
            off[t + 1] = off[t];
        }
    }
    __syncthreads();
    const int K0 = A.backward ? A.width[L] : A.width[0];
    const bool vec_in = (K0 & 3) == 0 && (reinterpret_cast<uintptr_t>(A.x) & 15) == 0;
    const int64_t n_tiles = (A.n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        int64_t srow[2];
        bool live[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t s = tile * 32 + 16 * j + c;
            live[j] = s < A.n;
            srow[j] = live[j] ? s : A.n - 1;
        }
        mf4 b[2][4];                                   // the tile's current activations: inputs 16q + 4g + s of sample c
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k0 = 16 * q + 4 * g;
                b[j][q] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
                if (k0 < K0) {
                    const float *p = A.x + srow[j] * K0 + k0;
                    if (vec_in && k0 + 3 < K0) {
                        b[j][q] = *reinterpret_cast<const mf4 *>(p);
                    } else {
#pragma unroll
                        for (int s = 0; s < 4; ++s) b[j][q][s] = (k0 + s < K0) ? p[s] : 0.0f;
                    }
                }
            }
#pragma unroll
        for (int t = 0; t < kMlpMaxLayers; ++t) {
            if (t >= L) break;
            const int l = A.backward ? L - 1 - t : t;
            const int N = A.backward ? A.width[l] : A.width[l + 1], K = A.backward ? A.width[l + 1] : A.width[l];
            const int NB = (N + 15) / 16, KQ = (K + 15) / 16;
            const float *const fl = frag + off[t];
            mf4 acc[2][4];
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                acc[0][nb] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
                acc[1][nb] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
                if (nb < NB) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q < KQ) {
                            const mf4 a = *reinterpret_cast<const mf4 *>(fl + ((nb * KQ + q) * 64 + lane) * 4);
#pragma unroll
                            for (int s = 0; s < 4; ++s)
#pragma unroll
                                for (int j = 0; j < 2; ++j)
                                    acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[j][q][s], acc[j][nb], 0, 0, 0);
                        }
                    }
                }
            }
            // activation / mask, store, and hand the registers to the next layer
            const bool relu = !A.backward && (t < L - 1 || A.relu_last);
            const float *const mk = A.backward ? A.mask[l] : nullptr;
            float *const dst = A.out[l];
            const bool vec_out = (N & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0 &&
                                 (!mk || (reinterpret_cast<uintptr_t>(mk) & 15) == 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    mf4 v = acc[j][nb];
                    const int o0 = 16 * nb + 4 * g;
                    if (nb < NB && o0 < N) {
                        if (relu) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
                        }
                        if (mk) {
                            if (vec_out && o0 + 3 < N) {
                                const mf4 m4 = *reinterpret_cast<const mf4 *>(mk + srow[j] * N + o0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) v[r] = m4[r] > 0.0f ? v[r] : 0.0f;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (o0 + r < N) v[r] = mk[srow[j] * N + o0 + r] > 0.0f ? v[r] : 0.0f;
                            }
                        }
                        if (dst && live[j]) {
                            float *out = dst + srow[j] * N + o0;
                            if (vec_out && o0 + 3 < N) {
                                *reinterpret_cast<mf4 *>(out) = v;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                    if (o0 + r < N) out[r] = v[r];
                            }
                        }
                    } else {
                        v = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
                    }
                    b[j][nb] = v;                      // outputs 16nb + 4g + r == the next layer's inputs 16q + 4g + s
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Backward of a whole MLP WITH its weight gradients (round 3): the walk of mlp_chain_kernel's backward direction, and at
// every layer, from the registers the walk already holds, dW_l += dz_l^T a_l -- instead of writing every dz_l for
// ced_weight_grad to read back together with a_l (512 B per sample and 64 x 64 layer, 8 layers per step).
// The walk holds a tile as [sample c][features 16q + 4g + s] per lane (g, c); dW's MFMA contracts over SAMPLES, so both
// operands are transposed through a per-wave LDS buffer ([16 samples][80 floats]: the 16-float pad makes the operand
// reads conflict-free): lane (g', c') of k-step ks supplies dz[sample 4ks + g'][neuron 16ob + c'] and
// a[sample 4ks + g'][input 16ib + c'].  Every wave keeps all of the network's dW tiles in accumulators over its share of
// the samples and writes them once; a fixed-order reduction sums the waves' partials (no float atomics, reproducible).
// Shapes: input width <= 16 KB0, H hidden layers of 64, output width <= 16 NBL (the model's networks; others take the
// layer-wise path).
struct MlpDwArgs {
    int64_t n;
    int k0, nl;                           // input width, output width
    const float *dy;                      // [n, nl]
    const float *act[kMlpMaxLayers];      // act[0] = x [n, k0]; act[l] = a_l [n, 64], l = 1..H
    const float *w[kMlpMaxLayers];        // W_l, l = 0..H
    float *g0;                            // [n, k0] or NULL
    float *partial;                       // [waves, stride]
    int stride;
    int layer_off[kMlpMaxLayers];
};

constexpr int kDwPad = 80;                // floats per sample row of the transposition buffers

__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// dw[ob][ib] += dz^T a over the 16 samples of one sub-tile.  dz: OB feature blocks, a: IB feature blocks, both as the
// walk holds them (lane (g, c): features 16q + 4g + s of sample c).
template <int OB, int IB>
__device__ __forceinline__ void dw_accumulate(const mf4 (&dz)[4], const mf4 (&a)[4], mf4 (&dw)[OB][IB], float *ta, float *tb,
                                              int g, int c)
{
    lds_fence();                                   // earlier reads of the buffers are done (in-order LDS) and not moved below
#pragma unroll
    for (int q = 0; q < OB; ++q) *reinterpret_cast<mf4 *>(ta + c * kDwPad + 16 * q + 4 * g) = dz[q];
#pragma unroll
    for (int q = 0; q < IB; ++q) *reinterpret_cast<mf4 *>(tb + c * kDwPad + 16 * q + 4 * g) = a[q];
    lds_fence();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        float av[OB], bv[IB];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob) av[ob] = ta[(4 * ks + g) * kDwPad + 16 * ob + c];
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) bv[ib] = tb[(4 * ks + g) * kDwPad + 16 * ib + c];
#pragma unroll
        for (int ob = 0; ob < OB; ++ob)
#pragma unroll
            for (int ib = 0; ib < IB; ++ib) dw[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ob], bv[ib], dw[ob][ib], 0, 0, 0);
    }
}

// loads features 16q + 4g + s (q < Q) of the tile's two 16-sample halves from a [n, width] array
template <int Q>
__device__ __forceinline__ void load_rows(const float *p, int width, const int64_t (&srow)[2], int g, mf4 (&v)[2][4])
{
    const bool vec = (width & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[j][q] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
            const int k0 = 16 * q + 4 * g;
            if (q < Q && k0 < width) {
                const float *r = p + srow[j] * width + k0;
                if (vec && k0 + 3 < width) {
                    v[j][q] = *reinterpret_cast<const mf4 *>(r);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) v[j][q][s] = (k0 + s < width) ? r[s] : 0.0f;
                }
            }
        }
}

// g = dz M with M = W^T staged as [nb][q][lane][4] (stage_layer): NB output blocks, KQ input blocks
template <int NB, int KQ>
__device__ __forceinline__ void chain_matmul(const float *fl, const mf4 (&b)[2][4], mf4 (&acc)[2][4], int lane)
{
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        acc[0][nb] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
        acc[1][nb] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
        if (nb < NB) {
#pragma unroll
            for (int q = 0; q < KQ; ++q) {
                const mf4 a = *reinterpret_cast<const mf4 *>(fl + ((nb * KQ + q) * 64 + lane) * 4);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[j][q][s], acc[j][nb], 0, 0, 0);
            }
        }
    }
}

template <int OB, int IB>
__device__ __forceinline__ void store_dw(const mf4 (&dw)[OB][IB], float *dst, int n_out, int n_in, int g, int c)
{
#pragma unroll
    for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int ib = 0; ib < IB; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 16 * ob + 4 * g + r, i = 16 * ib + c;
                if (o < n_out && i < n_in) dst[o * n_in + i] = dw[ob][ib][r];
            }
}

template <int KB0, int H, int NBL>
__global__ __launch_bounds__(256, 1) void mlp_bwd_dw_kernel(MlpDwArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float frag[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    constexpr int L = H + 1;
    // W_l^T fragments: layer H first (the walk's order)
    int off[L + 1];
    off[0] = 0;
#pragma unroll
    for (int t = 0; t < L; ++t) {
        const int l = H - t;
        const int n_out = l == H ? A.nl : 64, n_in = l == 0 ? A.k0 : 64;          // W_l [n_out, n_in]; M = W_l^T [n_in, n_out]
        stage_layer(frag + off[t], A.w[l], 1, n_in, n_in, n_out, tid, 256);
        off[t + 1] = off[t] + ((n_in + 15) / 16) * ((n_out + 15) / 16) * 256;
    }
    float *const ta = frag + off[L] + wave * 2 * 16 * kDwPad, *const tb = ta + 16 * kDwPad;
    __syncthreads();
    mf4 dw0[4][KB0], dwl[NBL][4];
    mf4 dwh[H > 1 ? H - 1 : 1][4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int k = 0; k < KB0; ++k) dw0[a][k] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int a = 0; a < NBL; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k) dwl[a][k] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
#pragma unroll
    for (int h = 0; h < (H > 1 ? H - 1 : 1); ++h)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int k = 0; k < 4; ++k) dwh[h][a][k] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };

    const int64_t n_tiles = (A.n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        int64_t srow[2];
        bool live[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t s = tile * 32 + 16 * j + c;
            live[j] = s < A.n;
            srow[j] = live[j] ? s : A.n - 1;
        }
        mf4 b[2][4], m[2][4], acc[2][4];
        load_rows<NBL>(A.dy, A.nl, srow, g, b);
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (!live[j]) {                         // samples beyond n contribute nothing anywhere
#pragma unroll
                for (int q = 0; q < 4; ++q) b[j][q] = mf4{ 0.0f, 0.0f, 0.0f, 0.0f };
            }
        // layer H (the linear output layer): dW_H += dy^T a_H; dz_{H-1} = (dy W_H) * [a_H > 0]
        load_rows<4>(A.act[H], 64, srow, g, m);
#pragma unroll
        for (int j = 0; j < 2; ++j) dw_accumulate<NBL, 4>(b[j], m[j], dwl, ta, tb, g, c);
        chain_matmul<4, NBL>(frag + off[0], b, acc, lane);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) b[j][q][r] = m[j][q][r] > 0.0f ? acc[j][q][r] : 0.0f;
        // hidden layers H-1 .. 1 (64 -> 64)
#pragma unroll
        for (int l = H - 1; l >= 1; --l) {
            load_rows<4>(A.act[l], 64, srow, g, m);
#pragma unroll
            for (int j = 0; j < 2; ++j) dw_accumulate<4, 4>(b[j], m[j], dwh[l - 1], ta, tb, g, c);
            chain_matmul<4, 4>(frag + off[H - l], b, acc, lane);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int r = 0; r < 4; ++r) b[j][q][r] = m[j][q][r] > 0.0f ? acc[j][q][r] : 0.0f;
        }
        // layer 0: dW_0 += dz_0^T x; the input gradient (no mask) when asked for
        load_rows<KB0>(A.act[0], A.k0, srow, g, m);
#pragma unroll
        for (int j = 0; j < 2; ++j) dw_accumulate<4, KB0>(b[j], m[j], dw0, ta, tb, g, c);
        if (A.g0) {
            chain_matmul<KB0, 4>(frag + off[H], b, acc, lane);
            const bool vec = (A.k0 & 3) == 0 && (reinterpret_cast<uintptr_t>(A.g0) & 15) == 0;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < KB0; ++q) {
                    const int o0 = 16 * q + 4 * g;
                    if (live[j] && o0 < A.k0) {
                        float *out = A.g0 + srow[j] * A.k0 + o0;
                        if (vec && o0 + 3 < A.k0) {
                            *reinterpret_cast<mf4 *>(out) = acc[j][q];
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                if (o0 + r < A.k0) out[r] = acc[j][q][r];
                        }
                    }
                }
        }
    }
    // this wave's partial tiles, row-major [n_out][n_in] per layer
    float *const mine = A.partial + (size_t)(blockIdx.x * 4 + wave) * A.stride;
    store_dw<4, KB0>(dw0, mine + A.layer_off[0], 64, A.k0, g, c);
#pragma unroll
    for (int l = 1; l < H; ++l) store_dw<4, 4>(dwh[l - 1], mine + A.layer_off[l], 64, 64, g, c);
    store_dw<NBL, 4>(dwl, mine + A.layer_off[H], A.nl, 64, g, c);
}

// out[e] = sum over the waves' partials, 16 interleaved chains per element combined in a fixed order
__global__ __launch_bounds__(1024) void mlp_dw_reduce_kernel(const float *partial, int n_partials, int stride, int total, float *out)
{
    __shared__ float red[16][64];
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + col;
    float sum = 0.0f;
    if (e < total)
        for (int b = q; b < n_partials; b += 16) sum += partial[(size_t)b * stride + e];
    red[q][col] = sum;
    __syncthreads();
    if (q == 0 && e < total) {
        float t = red[0][col];
#pragma unroll
        for (int k = 1; k < 16; ++k) t += red[k][col];
        out[e] = t;
    }
}

}  // namespace ced

static int dw_grid(int64_t n)
{
    const int64_t n_tiles = (n + 31) / 32;
    int64_t blocks = (n_tiles + 3) / 4;
    if (blocks > 256) blocks = 256;                 // one 4-wave workgroup per CU (the accumulators fill the registers)
    return (int)(blocks < 1 ? 1 : blocks);
}

extern "C" int64_t ced_mlp_backward_dw_workspace_bytes(int64_t n, int32_t n_layers, const int32_t *widths)
{
    if (!widths || n_layers < 2 || n_layers > ced::kMlpMaxLayers) return 0;
    int64_t total = 0;
    for (int l = 0; l < n_layers; ++l) total += (int64_t)widths[l] * widths[l + 1];
    return (int64_t)dw_grid(n) * 4 * total * (int64_t)sizeof(float);
}

// widths (host) [n_layers + 1]; acts (host array of device pointers) [n_layers]: acts[0] = the network input x,
// acts[l] = the forward's (post-ReLU) output of layer l-1; dws (device, [sum widths[l] * widths[l+1]] floats): the
// weight gradients back to back, dW_l row-major [widths[l+1], widths[l]]; g0 [n, widths[0]] or NULL.
// Returns CED_E_INVALID for shapes outside {input <= 48, hidden layers all 64 wide, 1..3 of them, output <= 32}.
extern "C" int ced_mlp_backward_dw(int64_t n, int32_t n_layers, const float *dy, const int32_t *widths,
                                   const float *const *weights, const float *const *acts, float *g0, float *dws,
                                   void *workspace, int64_t workspace_bytes, void *stream)
{
    CED_REQUIRE(n >= 0 && n_layers >= 2 && n_layers <= 4, "mlp_backward_dw: 2..4 layers");
    CED_REQUIRE(widths && weights && acts && dws, "mlp_backward_dw: null pointer");
    const int H = n_layers - 1, k0 = widths[0], nl = widths[n_layers];
    CED_REQUIRE(k0 >= 1 && k0 <= 48 && nl >= 1 && nl <= 32, "mlp_backward_dw: input width %d (1..48), output width %d (1..32)", k0, nl);
    for (int l = 1; l <= H; ++l) CED_REQUIRE(widths[l] == 64, "mlp_backward_dw: hidden width %d (64)", widths[l]);
    ced::MlpDwArgs A{};
    A.n = n; A.k0 = k0; A.nl = nl; A.dy = dy; A.g0 = g0; A.partial = (float *)workspace;
    int total = 0;
    size_t floats = 0;
    for (int l = 0; l < n_layers; ++l) {
        CED_REQUIRE(weights[l] && acts[l], "mlp_backward_dw: null weight / activation %d", l);
        A.w[l] = weights[l]; A.act[l] = acts[l];
        A.layer_off[l] = total;
        total += widths[l] * widths[l + 1];
        floats += (size_t)((widths[l] + 15) / 16) * ((widths[l + 1] + 15) / 16) * 256;
    }
    A.stride = total;
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(dws, 0, (size_t)total * 4, st) != hipSuccess) return ced::check_launch("mlp_backward_dw (memset)");
        return CED_OK;
    }
    CED_REQUIRE(dy != nullptr && workspace != nullptr, "mlp_backward_dw: null pointer");
    const int blocks = dw_grid(n);
    CED_REQUIRE(workspace_bytes >= (int64_t)blocks * 4 * total * 4, "mlp_backward_dw: workspace too small");
    const size_t lds = (floats + 4 * 2 * 16 * ced::kDwPad) * sizeof(float);
    CED_REQUIRE(lds <= 128 * 1024, "mlp_backward_dw: %zu bytes of LDS", lds);
    const int KB0 = (k0 + 15) / 16, NBL = (nl + 15) / 16;
    const void *fn = nullptr;
#define CED_DW_CASE(kb, h, nb)                                                                                      \
    if (KB0 == kb && H == h && NBL == nb) {                                                                         \
        fn = reinterpret_cast<const void *>(ced::mlp_bwd_dw_kernel<kb, h, nb>);                                     \
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);                      \
        hipLaunchKernelGGL((ced::mlp_bwd_dw_kernel<kb, h, nb>), dim3((unsigned)blocks), dim3(256), lds, st, A);     \
    }
    // every shape the interface admits (input blocks 1..3, hidden layers 1..3, output blocks 1..2)
    CED_DW_CASE(1, 1, 1) CED_DW_CASE(1, 2, 1) CED_DW_CASE(1, 3, 1) CED_DW_CASE(1, 1, 2) CED_DW_CASE(1, 2, 2) CED_DW_CASE(1, 3, 2)
    CED_DW_CASE(2, 1, 1) CED_DW_CASE(2, 2, 1) CED_DW_CASE(2, 3, 1) CED_DW_CASE(2, 1, 2) CED_DW_CASE(2, 2, 2) CED_DW_CASE(2, 3, 2)
    CED_DW_CASE(3, 1, 1) CED_DW_CASE(3, 2, 1) CED_DW_CASE(3, 3, 1) CED_DW_CASE(3, 1, 2) CED_DW_CASE(3, 2, 2) CED_DW_CASE(3, 3, 2)
#undef CED_DW_CASE
    CED_REQUIRE(fn != nullptr, "mlp_backward_dw: no kernel for input blocks %d, hidden layers %d, output blocks %d", KB0, H, NBL);
    int rc = ced::check_launch("mlp_backward_dw");
    if (rc) return rc;
    hipLaunchKernelGGL(ced::mlp_dw_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, st, (const float *)workspace,
                       blocks * 4, total, total, dws);
    return ced::check_launch("mlp_backward_dw (reduce)");
}

extern "C" int ced_mlp_chain(int64_t n, int32_t n_layers, int32_t backward, const float *x, const int32_t *widths,
                             const float *const *weights, float *const *outs, const float *const *masks, int32_t relu_last,
                             void *stream)
{
    CED_REQUIRE(n >= 0 && n_layers >= 1 && n_layers <= ced::kMlpMaxLayers, "mlp_chain: 1..%d layers", ced::kMlpMaxLayers);
    CED_REQUIRE(widths && weights && outs, "mlp_chain: null pointer");
    ced::MlpArgs A{};
    A.n = n; A.n_layers = n_layers; A.backward = backward ? 1 : 0; A.x = x; A.relu_last = relu_last ? 1 : 0;
    size_t floats = 0;
    for (int l = 0; l <= n_layers; ++l) {
        CED_REQUIRE(widths[l] >= 1 && widths[l] <= 64, "mlp_chain: width[%d] = %d (1..64)", l, widths[l]);
        A.width[l] = widths[l];
    }
    for (int l = 0; l < n_layers; ++l) {
        CED_REQUIRE(weights[l] != nullptr, "mlp_chain: null weight %d", l);
        A.w[l] = weights[l];
        A.out[l] = outs[l];
        A.mask[l] = (backward && masks) ? masks[l] : nullptr;
        floats += (size_t)((widths[l] + 15) / 16) * ((widths[l + 1] + 15) / 16) * 256;
    }
    if (n == 0) return CED_OK;
    CED_REQUIRE(x != nullptr, "mlp_chain: null input");
    const size_t lds = floats * sizeof(float);
    CED_REQUIRE(lds <= 128 * 1024, "mlp_chain: %zu bytes of layer fragments do not fit in LDS", lds);
    constexpr int kMaxDevices = 64;
    static bool attr_set[kMaxDevices] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return ced::check_launch("mlp_chain (device)");
    if (!attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(ced::mlp_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                128 * 1024) != hipSuccess)
            return ced::check_launch("mlp_chain (LDS attribute)");
        attr_set[dev] = true;
    }
    const int64_t n_tiles = (n + 31) / 32;
    int64_t blocks = (n_tiles + 3) / 4;
    // persistent over tiles: no more workgroups than are resident at once (4 per CU by registers, fewer when the layers'
    // fragments fill the LDS: a late second round of workgroups would only lengthen the launch)
    int per_cu = (int)((160 * 1024) / (lds > 0 ? lds : 1));
    if (per_cu > 4) per_cu = 4;
    if (per_cu < 1) per_cu = 1;
    static const int blocks_env = [] { const char *e = getenv("CED_MLP_MAX_BLOCKS"); return e ? atoi(e) : 0; }();
    const int64_t cap = blocks_env > 0 ? blocks_env : 256 * per_cu;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(ced::mlp_chain_kernel, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, A);
    return ced::check_launch("mlp_chain");
}
