// Weight gradient of a bias-free dense layer over a long sample stream (training path):
//     dW[o][i] = sum_s dY[s][o] * X[s][i],     S ~ 1e5..1e7 samples, n_out, n_in <= 64.
// Replaces the weight-gradient GEMM inside tiny-cuda-nn's Network backward (the tcnn modules built at
// cednerf/model.py:200-222,280-309, run by `loss.backward()` at train_real.py:414-419).  A library GEMM sees a
// 64x64 output with a million-deep reduction and runs it on a handful of workgroups (~1 ms per layer on MI355X);
// here the sample stream is split over the whole chip and every wave keeps the full n_out x n_in tile in
// v_mfma_f32_16x16x4_f32 accumulators, so the kernel streams X and dY once at HBM rate.
//
// Layout trick: accumulator block (nb, ib) row r / column c stands for neuron o = NBO*r + nb and input
// i = NBI*c + ib, so a lane's NBO (NBI) operands of one sample are contiguous in memory (one 16-byte load when the
// layer is 64 wide) and a wave reads four whole rows of dY and of X per MFMA step.
// Two deterministic stages: per-workgroup partial tiles (fixed sample -> wave assignment, fixed wave order), then a
// fixed-order sum of the partials -- the result is reproducible run to run (no float atomics).
#include "ced_common.hpp"

namespace ced {

typedef float wf4 __attribute__((ext_vector_type(4)));

struct WgradArgs {
    int64_t n;
    const float *x;     // [n, n_in]
    const float *dy;    // [n, n_out]
    int n_in, n_out;
    float *partial;     // [gridDim.x, n_out, n_in]
};

constexpr int kWgradWaves = 4;
constexpr int kWgradMaxBlocks = 768;          // three 4-wave workgroups per CU: all resident at once

template <int NB>
__device__ __forceinline__ void load_operands(const float *row, int first, int dim, bool valid, float (&v)[NB])
{
    // NB consecutive values row[first .. first+NB); a whole 64- or 32-wide row is vector-loadable
    if constexpr (NB == 4) {
        if (dim == 64) {
            const wf4 q = valid ? *reinterpret_cast<const wf4 *>(row + first) : wf4{0.f, 0.f, 0.f, 0.f};
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            return;
        }
    }
    if constexpr (NB == 2) {
        if (dim == 32) {
            typedef float wf2 __attribute__((ext_vector_type(2)));
            const wf2 q = valid ? *reinterpret_cast<const wf2 *>(row + first) : wf2{0.f, 0.f};
            v[0] = q.x; v[1] = q.y;
            return;
        }
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) v[k] = (valid && first + k < dim) ? row[first + k] : 0.0f;
}

template <int NBO, int NBI>
__global__ __launch_bounds__(kWgradWaves * 64) void wgrad_partial_kernel(WgradArgs A)
{
    __shared__ float red[kWgradWaves - 1][NBO * NBI * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, kk = lane >> 4;
    wf4 acc[NBO][NBI];
#pragma unroll
    for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
        for (int ib = 0; ib < NBI; ++ib) acc[nb][ib] = wf4{0.f, 0.f, 0.f, 0.f};

    // one MFMA step = 4 samples (k = lane >> 4).  A workgroup owns a contiguous range of steps and its four waves
    // interleave inside it, so the workgroup streams two contiguous spans (x and dy) instead of striding by megabytes.
    const int64_t n_steps = (A.n + 3) / 4;
    const int64_t per_block = (n_steps + gridDim.x - 1) / gridDim.x;
    const int64_t block_end = min(n_steps, (int64_t)(blockIdx.x + 1) * per_block);
    constexpr int64_t stride = kWgradWaves;
    constexpr int UNROLL = 4;
    for (int64_t step0 = (int64_t)blockIdx.x * per_block + wave; step0 < block_end; step0 += stride * UNROLL) {
        float a[UNROLL][NBO], b[UNROLL][NBI];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int64_t s = (step0 + u * stride) * 4 + kk;
            const bool valid = step0 + u * stride < block_end && s < A.n;
            const int64_t sc = valid ? s : 0;
            load_operands<NBO>(A.dy + sc * A.n_out, NBO * c, A.n_out, valid, a[u]);
            load_operands<NBI>(A.x + sc * A.n_in, NBI * c, A.n_in, valid, b[u]);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u)
#pragma unroll
            for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
                for (int ib = 0; ib < NBI; ++ib)
                    acc[nb][ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][nb], b[u][ib], acc[nb][ib], 0, 0, 0);
    }

    // waves 1..3 park their tiles in LDS; wave 0 adds them in wave order and writes the workgroup's partial
    if (wave > 0) {
#pragma unroll
        for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
            for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
                for (int v = 0; v < 4; ++v) red[wave - 1][((nb * NBI + ib) * 4 + v) * 64 + lane] = acc[nb][ib][v];
    }
    __syncthreads();
    if (wave == 0) {
        float *out = A.partial + (int64_t)blockIdx.x * A.n_out * A.n_in;
#pragma unroll
        for (int nb = 0; nb < NBO; ++nb)
#pragma unroll
            for (int ib = 0; ib < NBI; ++ib)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float sum = acc[nb][ib][v];
#pragma unroll
                    for (int w = 0; w < kWgradWaves - 1; ++w) sum += red[w][((nb * NBI + ib) * 4 + v) * 64 + lane];
                    const int o = NBO * (4 * kk + v) + nb, i = NBI * c + ib;
                    if (o < A.n_out && i < A.n_in) out[o * A.n_in + i] = sum;
                }
    }
}

// dW[e] = sum_b partial[b][e]: 16 interleaved chains per element (chain q takes b = q, q+16, ... in ascending order,
// loads issued 8 at a time), combined in ascending q -- a fixed order, so the sum is reproducible.
constexpr int kReduceChains = 16;
__global__ __launch_bounds__(64 * kReduceChains) void wgrad_reduce_kernel(const float *partial, int n_blocks, int n_elems, float *dw)
{
    __shared__ float red[kReduceChains][64];
    const int col = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + col;
    float sum = 0.0f;
    if (e < n_elems) {
        int b = q;
        for (; b + 7 * kReduceChains < n_blocks; b += 8 * kReduceChains) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * kReduceChains) * n_elems + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; b < n_blocks; b += kReduceChains) sum += partial[(int64_t)b * n_elems + e];
    }
    red[q][col] = sum;
    __syncthreads();
    if (q == 0 && e < n_elems) {
        float total = red[0][col];
#pragma unroll
        for (int k = 1; k < kReduceChains; ++k) total += red[k][col];
        dw[e] = total;
    }
}

static int wgrad_blocks(int64_t n)
{
    const int64_t steps = (n + 3) / 4;
    int64_t blocks = (steps + kWgradWaves * 4 - 1) / (kWgradWaves * 4);      // >= 4 MFMA steps per wave
    if (blocks > kWgradMaxBlocks) blocks = kWgradMaxBlocks;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

template <int NBO>
static void launch_partial(int nbi, int blocks, hipStream_t st, const WgradArgs &A)
{
    switch (nbi) {
        case 1: hipLaunchKernelGGL((wgrad_partial_kernel<NBO, 1>), dim3(blocks), dim3(kWgradWaves * 64), 0, st, A); break;
        case 2: hipLaunchKernelGGL((wgrad_partial_kernel<NBO, 2>), dim3(blocks), dim3(kWgradWaves * 64), 0, st, A); break;
        case 3: hipLaunchKernelGGL((wgrad_partial_kernel<NBO, 3>), dim3(blocks), dim3(kWgradWaves * 64), 0, st, A); break;
        default: hipLaunchKernelGGL((wgrad_partial_kernel<NBO, 4>), dim3(blocks), dim3(kWgradWaves * 64), 0, st, A); break;
    }
}

}  // namespace ced

extern "C" int64_t ced_weight_grad_workspace_bytes(int64_t n, int32_t n_out, int32_t n_in)
{
    if (n <= 0 || n_out <= 0 || n_in <= 0) return 0;
    return (int64_t)ced::wgrad_blocks(n) * n_out * n_in * (int64_t)sizeof(float);
}

extern "C" int ced_weight_grad(int64_t n, const float *x, int32_t n_in, const float *dy, int32_t n_out, float *dw,
                               void *workspace, int64_t workspace_bytes, void *stream)
{
    CED_REQUIRE(n >= 0, "weight_grad: n < 0");
    CED_REQUIRE(n_in >= 1 && n_in <= 64 && n_out >= 1 && n_out <= 64, "weight_grad: layer widths must be in 1..64 (got %d x %d)",
                (int)n_out, (int)n_in);
    CED_REQUIRE(dw, "weight_grad: null output");
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * n_out * n_in, st);
        if (e != hipSuccess) {
            ced::set_error("weight_grad: %s", hipGetErrorString(e));
            return CED_E_LAUNCH;
        }
        return CED_OK;
    }
    CED_REQUIRE(x && dy, "weight_grad: null input");
    CED_REQUIRE(workspace && workspace_bytes >= ced_weight_grad_workspace_bytes(n, n_out, n_in),
                "weight_grad: workspace too small (need %lld bytes)", (long long)ced_weight_grad_workspace_bytes(n, n_out, n_in));
    CED_REQUIRE((((uintptr_t)x | (uintptr_t)dy) & 15) == 0, "weight_grad: x and dy must be 16-byte aligned");
    const int blocks = ced::wgrad_blocks(n);
    ced::WgradArgs A{n, x, dy, (int)n_in, (int)n_out, (float *)workspace};
    const int nbo = (n_out + 15) / 16, nbi = (n_in + 15) / 16;
    switch (nbo) {
        case 1: ced::launch_partial<1>(nbi, blocks, st, A); break;
        case 2: ced::launch_partial<2>(nbi, blocks, st, A); break;
        case 3: ced::launch_partial<3>(nbi, blocks, st, A); break;
        default: ced::launch_partial<4>(nbi, blocks, st, A); break;
    }
    int rc = ced::check_launch("weight_grad (partial)");
    if (rc != CED_OK) return rc;
    const int n_elems = n_out * n_in;
    hipLaunchKernelGGL(ced::wgrad_reduce_kernel, dim3((n_elems + 63) / 64), dim3(64 * ced::kReduceChains), 0, st, (const float *)workspace, blocks,
                       n_elems, dw);
    return ced::check_launch("weight_grad (reduce)");
}
