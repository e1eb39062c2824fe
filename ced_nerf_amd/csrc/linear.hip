// Bias-free dense layers of the training path, hand-written for gfx950 in place of library GEMMs:
//   forward         y[s][o]  = act( sum_i x[s][i] * W[o][i] )                       (tcnn FullyFusedMLP layers of
//   input gradient  dx[s][i] = ( sum_o dz[s][o] * W[o][i] ) * [ mask[s][i] > 0 ]     cednerf/model.py:200-222,280-344)
// are the same tall-skinny product out = in * M^T with M = W (forward) or W^T (backward); widths <= 64, millions of
// samples.  (The weight gradient dW = dz^T y is ced_weight_grad, wgrad.hip.)  Replaces the forward / backward of the
// tiny-cuda-nn Networks the reference trains through (loss.backward(), train_real.py:412-420).
//
// Execution shape: one 256-thread workgroup keeps the (<= 64 x 64) matrix in LDS in MFMA A-fragment order; a wave
// takes 32 samples at a time: D^T = M * X^T on v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate -- gradients get
// the inference kernels' arithmetic), samples on lanes.  The k index of the MFMAs is PERMUTED so that lane group g
// feeds the 16-byte chunk g of every 64-byte piece of a sample's row: a row is read once, in whole 16-byte loads,
// and every lane stores 16 contiguous bytes of the output row.  Memory-bound by design: 4*(K + N) bytes per sample.
#include "ced_common.hpp"

namespace ced {

typedef float lf4 __attribute__((ext_vector_type(4)));

struct LinearArgs {
    int64_t n;
    const float *x;        // [n, K]
    const float *m;        // M[o][i] = m[o * m_so + i * m_si], o < N, i < K
    int64_t m_so, m_si;
    const float *mask;     // [n, N] or NULL: out *= (mask > 0)
    float *y;              // [n, N]
    int K, N, relu;
};

// KQ = ceil(K / 16) quads of k-steps, NB = ceil(N / 16) output blocks
template <int KQ, int NB>
__global__ __launch_bounds__(256) void linear_kernel(LinearArgs A)
{
    __shared__ __attribute__((aligned(16))) float frag[NB * KQ * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    // A-fragment order: [nb][q][lane = kk*16 + row][s] holds M[16nb + row][k(4q + s, kk)], k(S, kk) = 16*(S/4) + 4*kk + S%4
    for (int e = tid; e < NB * KQ * 256; e += 256) {
        const int s = e & 3, ln = (e >> 2) & 63, q = (e >> 8) % KQ, nb = (e >> 8) / KQ;
        const int row = 16 * nb + (ln & 15), k = 16 * q + 4 * (ln >> 4) + s;
        frag[e] = (row < A.N && k < A.K) ? A.m[row * A.m_so + k * A.m_si] : 0.0f;
    }
    __syncthreads();
    const bool vec_in = (A.K & 3) == 0 && (reinterpret_cast<uintptr_t>(A.x) & 15) == 0;
    const bool vec_out = (A.N & 3) == 0 && (reinterpret_cast<uintptr_t>(A.y) & 15) == 0 &&
                         (!A.mask || (reinterpret_cast<uintptr_t>(A.mask) & 15) == 0);
    const int64_t n_tiles = (A.n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tiles; tile += (int64_t)gridDim.x * 4) {
        lf4 acc[2][NB];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[j][nb] = lf4{ 0.0f, 0.0f, 0.0f, 0.0f };
        int64_t srow[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t s = tile * 32 + 16 * j + c;
            srow[j] = s < A.n ? s : A.n - 1;
        }
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            lf4 b[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k0 = 16 * q + 4 * g;
                const float *p = A.x + srow[j] * A.K + k0;
                if (vec_in && k0 + 3 < A.K) {
                    b[j] = *reinterpret_cast<const lf4 *>(p);
                } else {
#pragma unroll
                    for (int s = 0; s < 4; ++s) b[j][s] = (k0 + s < A.K) ? p[s] : 0.0f;
                }
            }
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const lf4 a = *reinterpret_cast<const lf4 *>(frag + ((nb * KQ + q) * 64 + lane) * 4);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[j][s], acc[j][nb], 0, 0, 0);
            }
        }
        // lane (g, c) holds outputs 16nb + 4g + r of sample c
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t s = tile * 32 + 16 * j + c;
            if (s >= A.n) continue;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int o0 = 16 * nb + 4 * g;
                if (o0 >= A.N) continue;
                lf4 v = acc[j][nb];
                if (A.relu) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.0f ? v[r] : 0.0f;
                }
                float *out = A.y + s * A.N + o0;
                if (vec_out && o0 + 3 < A.N) {
                    if (A.mask) {
                        const lf4 mk = *reinterpret_cast<const lf4 *>(A.mask + s * A.N + o0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = mk[r] > 0.0f ? v[r] : 0.0f;
                    }
                    *reinterpret_cast<lf4 *>(out) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (o0 + r < A.N) {
                            float w = v[r];
                            if (A.mask) w = A.mask[s * A.N + o0 + r] > 0.0f ? w : 0.0f;
                            out[r] = w;
                        }
                    }
                }
            }
        }
    }
}

template <int KQ>
static void launch_linear_nb(const LinearArgs &A, int nb, dim3 grid, hipStream_t stream)
{
    switch (nb) {
    case 1: hipLaunchKernelGGL((linear_kernel<KQ, 1>), grid, dim3(256), 0, stream, A); break;
    case 2: hipLaunchKernelGGL((linear_kernel<KQ, 2>), grid, dim3(256), 0, stream, A); break;
    case 3: hipLaunchKernelGGL((linear_kernel<KQ, 3>), grid, dim3(256), 0, stream, A); break;
    default: hipLaunchKernelGGL((linear_kernel<KQ, 4>), grid, dim3(256), 0, stream, A); break;
    }
}

}  // namespace ced

extern "C" int ced_linear(int64_t n, const float *x, int32_t n_in, const float *w, int32_t w_rows, int32_t w_cols,
                          int32_t transpose_w, int32_t n_out, int32_t relu, const float *mask, float *y, void *stream)
{
    CED_REQUIRE(n >= 0 && n_in >= 1 && n_in <= 64 && n_out >= 1 && n_out <= 64, "linear: widths must be 1..64 (n_in=%d n_out=%d)",
                n_in, n_out);
    // forward: y = x W^T with W [n_out, n_in]; transpose_w: y = x W with W [n_in, n_out] (the input gradient dz W)
    CED_REQUIRE(transpose_w ? (w_rows == n_in && w_cols == n_out) : (w_rows == n_out && w_cols == n_in),
                "linear: weight shape [%d, %d] does not match n_in=%d n_out=%d transpose_w=%d", w_rows, w_cols, n_in, n_out,
                transpose_w);
    if (n == 0) return CED_OK;
    CED_REQUIRE(x && w && y, "linear: null pointer");
    ced::LinearArgs A{};
    A.n = n; A.x = x; A.m = w; A.mask = mask; A.y = y; A.K = n_in; A.N = n_out; A.relu = relu ? 1 : 0;
    if (transpose_w) { A.m_so = 1; A.m_si = w_cols; } else { A.m_so = w_cols; A.m_si = 1; }
    const int kq = (n_in + 15) / 16, nb = (n_out + 15) / 16;
    const int64_t n_tiles = (n + 31) / 32;
    int64_t blocks = (n_tiles + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks);
    switch (kq) {
    case 1: ced::launch_linear_nb<1>(A, nb, grid, (hipStream_t)stream); break;
    case 2: ced::launch_linear_nb<2>(A, nb, grid, (hipStream_t)stream); break;
    case 3: ced::launch_linear_nb<3>(A, nb, grid, (hipStream_t)stream); break;
    default: ced::launch_linear_nb<4>(A, nb, grid, (hipStream_t)stream); break;
    }
    return ced::check_launch("linear");
}
