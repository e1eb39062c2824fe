// Probe: issue cost of the two fp16 MFMA forms the half-precision field kernels can be built on (gfx950).
// One wave per SIMD (256-thread workgroups, one per CU), 4 independent accumulators, back-to-back issue; cycles from
// s_memtime (constant 100 MHz) and the shader clock (clock64).  Prints cycles per instruction per SIMD.
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_f16_rate tools/probes/mfma_f16_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int FORM> __global__ __launch_bounds__(256) void rate(float *out, long long *cycles, int iters)
{
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    f4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f4{ (float)k, 1, 2, 3 };
    const h4 a0 = { a[0], a[1], a[2], a[3] }, b0 = { b[0], b[1], b[2], b[3] };
    typedef float f16v __attribute__((ext_vector_type(16)));
    f16v big[4];
    for (int k = 0; k < 4; ++k) for (int e = 0; e < 16; ++e) big[k][e] = (float)(k + e);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        // eight independent accumulation chains, issued back to back (inline asm: the compiler neither reorders nor pads)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (FORM == 0) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a0), "v"(b0));
            else if (FORM == 1) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
            else if (FORM == 2) asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(big[k & 3]) : "v"(a0), "v"(b0));
            else if (FORM == 3) asm volatile("v_mfma_f32_4x4x4_16b_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a0), "v"(b0));
            else if (FORM == 4) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[0]) : "v"(a0), "v"(b0));      // one dependent chain
            else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[k >> 1 & 1]) : "v"(a0), "v"(b0));         // pairs: a a b b a a b b
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    for (int k = 0; k < 4; ++k) s += big[k][0] + big[k][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main()
{
    float *out; long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int form = 0; form < 6; ++form) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (form == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else if (form == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else if (form == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else if (form == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else if (form == 4) hipLaunchKernelGGL(rate<4>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(rate<5>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[256]; hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        const double n = 8.0 * iters;
        printf("%s: %.3f ms for %d MFMAs per wave, one wave per SIMD -> %.2f ns per MFMA per SIMD; clock64 ticks per MFMA %.2f\n",
               form == 0 ? "v_mfma_f32_16x16x16_f16" : form == 1 ? "v_mfma_f32_16x16x32_f16" : form == 2 ? "v_mfma_f32_32x32x8_f16" : form == 3 ? "v_mfma_f32_4x4x4_16b_f16" : form == 4 ? "16x16x16_f16, ONE dependent chain" : "16x16x16_f16, dependent pairs (a a b b)", ms, (int)n, ms * 1e6 / n, (double)h[0] / n);
    }
    return 0;
}
