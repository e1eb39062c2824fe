// Probe: does v_mfma_f32_16x16x32_f16 honour fp16 subnormal inputs, and how does it accumulate?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(const float *a_in, const float *b_in, float *out)
{
    const int lane = threadIdx.x;
    h8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (_Float16)a_in[(lane & 15) * 32 + 8 * (lane >> 4) + e];      // A[row][k]
        b[e] = (_Float16)b_in[(8 * (lane >> 4) + e) * 16 + (lane & 15)];    // B[k][col]
    }
    f4 c = { 0, 0, 0, 0 };
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[(4 * (lane >> 4) + r) * 16 + (lane & 15)] = c[r];
}
int main()
{
    float ha[16 * 32], hb[32 * 16], ho[256];
    float *da, *db, *dout;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dout, sizeof ho);
    // test 1: A = 2^-20 (fp16 subnormal) in k=0 only, B = 2^10 -> expect 2^-10 if honoured, 0 if flushed
    for (int i = 0; i < 512; ++i) { ha[i] = 0; hb[i] = 0; }
    for (int r = 0; r < 16; ++r) ha[r * 32 + 0] = ldexpf(1.0f, -20);
    for (int c = 0; c < 16; ++c) hb[0 * 16 + c] = 1024.0f;
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    printf("subnormal A (2^-20) * 2^10 = %g (expect %g if subnormals are honoured)\n", ho[0], ldexpf(1.0f, -10));
    // test 2: accumulation accuracy: random values, compare with double and with sequential fp32 fma
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.0f - 0.5f; };
    for (int i = 0; i < 512; ++i) { ha[i] = (float)(_Float16)rnd(); hb[i] = (float)(_Float16)(rnd() * 8.0f); }
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout);
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    double max_rel_d = 0; int n_eq_seq = 0, n_eq_dbl = 0;
    for (int r = 0; r < 16; ++r)
        for (int c = 0; c < 16; ++c) {
            double d = 0; float f = 0;
            for (int k = 0; k < 32; ++k) { d += (double)ha[r * 32 + k] * hb[k * 16 + c]; f = fmaf(ha[r * 32 + k], hb[k * 16 + c], f); }
            max_rel_d = fmax(max_rel_d, fabs(ho[r * 16 + c] - d) / fmax(fabs(d), 1e-3));
            n_eq_seq += (ho[r * 16 + c] == f);
            n_eq_dbl += (ho[r * 16 + c] == (float)d);
        }
    printf("K=32 dot: max rel err vs double %.3e; equal to sequential fp32 fma: %d/256; equal to rounded double: %d/256\n",
           max_rel_d, n_eq_seq, n_eq_dbl);
    return 0;
}
