// Probe: do packed-fp32 VALU instructions with an op_sel swizzle return wrong results while OTHER waves of the same SIMD
// run v_mfma_f32_16x16x32_f16?  (Background: DESIGN 4.1b -- the half-precision field kernels built on that MFMA were
// irreproducible exactly when the compiler's SLP vectoriser had turned the hash-coordinate arithmetic into
// `v_pk_mul_f32 ... op_sel:[0,1]` / `v_pk_mov_b32 ... op_sel:[1,0]`.)
//
// One workgroup of 768 threads per CU = three waves per SIMD (wave w runs on SIMD w % 4).  Waves 0..3 loop MFMAs
// (form chosen by `mfma_mode`: 0 none, 1 16x16x32_f16, 2 two 16x16x16_f16) and compare every product with their first
// one; waves 4..11 loop one packed instruction (form chosen by `pk_mode`) on changing operands and compare each
// result with the same arithmetic done by unpacked instructions.  Both kinds of mismatch are counted.
//   build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o tools/probes/pk_opsel_mfma tools/probes/pk_opsel_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f2 pk_op(int mode, f2 a, f2 b, f2 &want)
{
    f2 d;
    switch (mode) {
    case 0:   // lo = a.lo * b.hi, hi = a.hi * b.hi
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[0] * b[1], a[1] * b[1] };
        break;
    case 1:   // lo = a.hi, hi = b.lo
        asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[1], b[0] };
        break;
    case 2:   // no swizzle
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[0] * b[0], a[1] * b[1] };
        break;
    case 3:
        asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[0] - b[0], a[1] - b[1] };
        break;
    case 4:   // lo = a.lo * b.lo, hi = a.lo * b.hi  (op_sel_hi:[0,1])
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[0] * b[0], a[0] * b[1] };
        break;
    case 6:   // lo = a.hi * b.lo, hi = a.hi * b.hi
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[1] * b[0], a[1] * b[1] };
        break;
    case 7:   // lo = a.hi * b.hi, hi = a.hi * b.hi
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[1] * b[1], a[1] * b[1] };
        break;
    case 8:   // lo = a.lo + b.hi, hi = a.hi + b.hi
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[0] + b[1], a[1] + b[1] };
        break;
    case 9:   // lo = fma(a.lo, b.hi, a.lo), hi = fma(a.hi, b.hi, a.hi)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,1,0]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ __builtin_fmaf(a[0], b[1], a[0]), __builtin_fmaf(a[1], b[1], a[1]) };
        break;
    case 10:  // lo = fma(a.lo, b.lo, a.hi), hi = fma(a.hi, b.hi, a.hi)
        asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel:[0,0,1]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ __builtin_fmaf(a[0], b[0], a[1]), __builtin_fmaf(a[1], b[1], a[1]) };
        break;
    case 11:  // the broadcast form the kernels do contain: lo = a.lo * b.lo, hi = a.hi * b.lo
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(d) : "v"(a), "v"(b));
        want = f2{ a[0] * b[0], a[1] * b[0] };
        break;
    case 5:   // the dependent pair the kernels contained: packed subtract, then the swizzled move of its result
        {
            f2 t;
            asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(a), "v"(b));
            asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(d) : "v"(t), "v"(b));
            want = f2{ a[1] - b[1], b[0] };
        }
        break;
    default:
        d = want = a;
        break;
    }
    return d;
}

__global__ __launch_bounds__(768) void probe(int mfma_mode, int pk_mode, int iters, unsigned long long *errs, float *sink, float *example)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long bad = 0;
    if (wave < 4) {
        if (mfma_mode == 0) return;
        h8 a, b[4];
        for (int e = 0; e < 8; ++e) {
            a[e] = (_Float16)(0.03125f * (float)((lane * 7 + e * 3 + blockIdx.x) % 61 - 30));
            for (int j = 0; j < 4; ++j) b[j][e] = (_Float16)(0.0625f * (float)((lane * 5 + e * 11 + j * 13) % 53 - 26));
        }
        f4 first[4];
        for (int it = 0; it < iters; ++it) {
            f4 acc[4];
            f16v acc32[4];
            for (int j = 0; j < 4; ++j) {
                acc[j] = f4{ 0.0f, 0.0f, 0.0f, 0.0f };
                for (int r = 0; r < 16; ++r) acc32[j][r] = 0.0f;
            }
#pragma unroll
            for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (mfma_mode == 1) {
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b[j], acc[j], 0, 0, 0);
                    } else if (mfma_mode == 3) {
                        // the other full-rate fp16 form of gfx950 (round 4): 32x32x16, 16 accumulator registers
                        acc32[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[j], acc32[j], 0, 0, 0);
                        acc[j] = f4{ acc32[j][0], acc32[j][5], acc32[j][10], acc32[j][15] };
                    } else {
                        const h4 a0 = { a[0], a[1], a[2], a[3] }, a1 = { a[4], a[5], a[6], a[7] };
                        const h4 b0 = { b[j][0], b[j][1], b[j][2], b[j][3] }, b1 = { b[j][4], b[j][5], b[j][6], b[j][7] };
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a0, b0, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a1, b1, acc[j], 0, 0, 0);
                    }
                }
            }
            for (int j = 0; j < 4; ++j) {
                if (it == 0) first[j] = acc[j];
                for (int r = 0; r < 4; ++r) bad += (__float_as_uint(acc[j][r]) != __float_as_uint(first[j][r]));
            }
            asm volatile("" : "+v"(a));     // keeps the loop body from being hoisted
        }
        sink[blockIdx.x * 768 + threadIdx.x] = first[0][0] + first[3][3];
        if (bad) atomicAdd(&errs[0], bad);
    } else {
        unsigned s = 1234567u * (threadIdx.x + 1) + blockIdx.x;
        f2 keep = { 0.0f, 0.0f };
        for (int it = 0; it < iters * 16; ++it) {
            s = s * 1664525u + 1013904223u;
            f2 a = { (float)(s & 0xffffu) * 0.001f + 0.5f, (float)((s >> 8) & 0xffffu) * 0.003f + 1.5f };
            f2 b = { (float)((s >> 4) & 0xfffu) * 0.01f + 0.25f, (float)((s >> 12) & 0xfffu) * 0.02f + 0.75f };
            f2 want;
            const f2 d = pk_op(pk_mode, a, b, want);
            const int wrong = (d[0] != want[0]) + (d[1] != want[1]);
            if (wrong && bad == 0 && atomicAdd(&errs[2], 1ull) == 0) {    // first wrong result of the launch: keep it
                example[0] = a[0]; example[1] = a[1]; example[2] = b[0]; example[3] = b[1];
                example[4] = d[0]; example[5] = d[1]; example[6] = want[0]; example[7] = want[1];
                example[8] = (float)lane; example[9] = (float)wave;
            }
            bad += wrong;
            keep += d;
        }
        sink[blockIdx.x * 768 + threadIdx.x] = keep[0] + keep[1];
        if (bad) atomicAdd(&errs[1], bad);
    }
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    unsigned long long *errs;
    float *sink;
    hipMalloc(&errs, 32);
    float *example;
    hipMalloc(&example, 64);
    hipMalloc(&sink, 256 * 768 * sizeof(float));
    const char *mf[] = { "no MFMA", "16x16x32_f16", "2 x 16x16x16_f16", "32x32x16_f16" };
    const char *pk[] = { "v_pk_mul_f32 op_sel:[0,1]", "v_pk_mov_b32 op_sel:[1,0]", "v_pk_mul_f32", "v_pk_add_f32 neg",
                         "v_pk_mul_f32 op_sel_hi:[0,1]", "v_pk_add_f32 neg -> v_pk_mov_b32 op_sel:[1,0]", "v_pk_mul_f32 op_sel:[1,0]",
                         "v_pk_mul_f32 op_sel:[1,1]", "v_pk_add_f32 op_sel:[0,1]", "v_pk_fma_f32 op_sel:[0,1,0]",
                         "v_pk_fma_f32 op_sel:[0,0,1]", "v_pk_mul_f32 op_sel_hi:[1,0]" };
    const int order[] = { 0, 6, 7, 8, 9, 10, 1, 5, 2, 3, 4, 11 };
    for (int m = 0; m < 4; ++m)
        for (int q = 0; q < 12; ++q) {
            const int p = order[q];
            hipMemset(errs, 0, 32);
            hipMemset(example, 0, 64);
            hipLaunchKernelGGL(probe, dim3(256), dim3(768), 0, 0, m, p, iters, errs, sink, example);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
            unsigned long long h[2];
            hipMemcpy(h, errs, 16, hipMemcpyDeviceToHost);
            printf("%-18s | %-46s | MFMA mismatches %llu, packed-op mismatches %llu\n", mf[m], pk[p], h[0], h[1]);
            if (h[1]) {
                float e[16];
                hipMemcpy(e, example, 64, hipMemcpyDeviceToHost);
                printf("      e.g. lane %g wave %g: a = (%.9g, %.9g) b = (%.9g, %.9g) got (%.9g, %.9g) want (%.9g, %.9g); a.lo*b.lo = %.9g\n",
                       e[8], e[9], e[0], e[1], e[2], e[3], e[4], e[5], e[6], e[7], e[0] * e[2]);
            }
        }
    return 0;
}
