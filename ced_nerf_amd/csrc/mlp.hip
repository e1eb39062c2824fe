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

}  // namespace ced

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
