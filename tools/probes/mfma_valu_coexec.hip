// Probe: how much vector work fits beside v_mfma_f32_16x16x16_f16 on one SIMD of gfx950?
// Workgroups of 512 threads = 8 waves, two per SIMD (wave w and w + 4 share SIMD w % 4).  The first four waves run a
// stream of MFMAs (8 independent accumulators, back to back), the other four a stream of one kind of vector
// instruction.  Three launches per kind: MFMA waves alone, vector waves alone, both together; time per instruction of
// each stream from the wall clock of the launch (the streams are sized to last about equally long).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/mfma_valu_coexec tools/probes/mfma_valu_coexec.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// KIND: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_cvt_pk_f16_f32, 3 v_fma_mixlo_f16, 4 v_med3_f32, 5 v_mul_lo_u32, 6 v_xor_b32,
//       7 a SECOND MFMA stream (two MFMA waves on the SIMD)
template <int KIND> __global__ __launch_bounds__(512) void coexec(float *out, int mfma_iters, int valu_iters, long long *ticks)
{
    const int wave = threadIdx.x >> 6;
    const bool mfma_role = wave < 4;
    float s = 0;
    const long long t0 = wall_clock64();
    if (mfma_role || KIND == 7) {
        const int iters = mfma_role ? mfma_iters : valu_iters;
        h4 a, b;
        for (int e = 0; e < 4; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
        f4 acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = f4{ (float)k, 1, 2, 3 };
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
        for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    } else {
        float x[8], y = 1.0001f + threadIdx.x * 1e-7f, z = 0.5f;
        f2 p[8];
        uint32_t u[8];
        for (int k = 0; k < 8; ++k) { x[k] = (float)k + threadIdx.x; p[k] = f2{ x[k], x[k] + 1 }; u[k] = threadIdx.x * 2654435761u + k; }
        const f2 y2 = { y, y }, z2 = { z, z };
        for (int i = 0; i < valu_iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
                else if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(y2), "v"(z2));
                else if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u[k]) : "v"(x[k]), "v"(y));
                else if (KIND == 3) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(y), "v"(z));
                else if (KIND == 4) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(z), "v"(y));
                else if (KIND == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
                else asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
            }
        }
        for (int k = 0; k < 8; ++k) s += x[k] + p[k][0] + p[k][1] + (float)u[k];
    }
    const long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 8 + wave] = t1 - t0;      // 100 MHz ticks
}

template <int KIND> static void run(const char *name, float *out, long long *ticks, int valu_per_mfma)
{
    const int mi = 20000, vi = mi * valu_per_mfma;
    printf("%-18s", name);
    for (int mode = 0; mode < 3; ++mode) {          // 0: MFMA alone, 1: vector alone, 2: both
        const int m = mode == 1 ? 0 : mi, v = mode == 0 ? 0 : vi;
        hipLaunchKernelGGL(coexec<KIND>, dim3(256), dim3(512), 0, 0, out, m, v, ticks);
        (void)hipDeviceSynchronize();
        long long h[8];
        (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
        // ns per instruction of each stream, from the role's own wall-clock ticks (10 ns each)
        const double ns_m = m ? h[0] * 10.0 / (8.0 * m) : 0, ns_v = v ? h[4] * 10.0 / (8.0 * v) : 0;
        printf("  %s: mfma %6.2f ns  vec %6.2f ns |", mode == 0 ? "mfma alone" : mode == 1 ? "vec alone " : "together  ", ns_m, ns_v);
    }
    printf("\n");
}

// second experiment: SLOTS waves per SIMD (workgroup of 256 * SLOTS threads); slot s runs MFMAs when bit s of mfma_mask is set,
// v_fma_f32 when bit s of valu_mask is set, nothing otherwise.  Reports the aggregate rate per SIMD of each kind.
__global__ __launch_bounds__(1024) void slots(float *out, int mfma_mask, int valu_mask, int iters, long long *ticks)
{
    const int slot = threadIdx.x >> 8;       // waves 4s .. 4s+3 are slot s (one per SIMD)
    float s = 0;
    const long long t0 = wall_clock64();
    if ((mfma_mask >> slot) & 1) {
        h4 a, b;
        for (int e = 0; e < 4; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
        f4 acc[8];
        for (int k = 0; k < 8; ++k) acc[k] = f4{ (float)k, 1, 2, 3 };
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n s_nop 15" ::: "memory");
        for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    } else if ((valu_mask >> slot) & 1) {
        float x[8], y = 1.0001f + threadIdx.x * 1e-7f, z = 0.5f;
        for (int k = 0; k < 8; ++k) x[k] = (float)k + threadIdx.x;
        for (int i = 0; i < 3 * iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
        }
        for (int k = 0; k < 8; ++k) s += x[k];
    }
    const long long t1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// third experiment: ONE wave per SIMD issuing groups of one MFMA followed by NV independent v_fma_f32
template <int NV> __global__ __launch_bounds__(256) void interleaved(float *out, int iters, long long *ticks)
{
    h4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (threadIdx.x - e)); }
    f4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f4{ (float)k, 1, 2, 3 };
    float x[8], y = 1.0001f + threadIdx.x * 1e-7f, z = 0.5f;
    for (int k = 0; k < 8; ++k) x[k] = (float)k + threadIdx.x;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < NV; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(k + v) & 7]) : "v"(y), "v"(z));
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    const long long t1 = wall_clock64();
    float s = 0;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3] + x[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int NV> static void run_interleaved(float *out, long long *ticks)
{
    const int iters = 20000;
    hipLaunchKernelGGL(interleaved<NV>, dim3(256), dim3(256), 0, 0, out, iters, ticks);
    (void)hipDeviceSynchronize();
    long long h;
    (void)hipMemcpy(&h, ticks, sizeof h, hipMemcpyDeviceToHost);
    printf("one wave per SIMD, groups of 1 MFMA + %d v_fma_f32: %.2f ns per group\n", NV, h * 10.0 / (8.0 * iters));
}

static void run_slots(float *out, long long *ticks, int mfma_mask, int valu_mask)
{
    const int iters = 20000;
    hipLaunchKernelGGL(slots, dim3(256), dim3(1024), 0, 0, out, mfma_mask, valu_mask, iters, ticks);
    (void)hipDeviceSynchronize();
    long long h[16];
    (void)hipMemcpy(h, ticks, sizeof h, hipMemcpyDeviceToHost);
    double rate_m = 0, rate_v = 0;        // instructions per ns per SIMD
    int nm = 0, nv = 0;
    for (int sl = 0; sl < 4; ++sl) {
        const double ns = h[4 * sl] * 10.0;
        if ((mfma_mask >> sl) & 1) { rate_m += 8.0 * iters / ns; ++nm; }
        else if ((valu_mask >> sl) & 1) { rate_v += 24.0 * iters / ns; ++nv; }
    }
    printf("%d MFMA wave(s) + %d v_fma_f32 wave(s) per SIMD: ", nm, nv);
    if (nm) printf("one MFMA per %.2f ns per SIMD   ", 1.0 / rate_m);
    if (nv) printf("one v_fma_f32 per %.2f ns per SIMD", 1.0 / rate_v);
    printf("\n");
}

int main()
{
    float *out; long long *ticks;
    (void)hipMalloc(&out, 256 * 1024 * 4); (void)hipMalloc(&ticks, 256 * 16 * 8);
    // warm up (clocks)
    hipLaunchKernelGGL(coexec<0>, dim3(256), dim3(512), 0, 0, out, 200000, 200000, ticks); (void)hipDeviceSynchronize();
    printf("ns per instruction per wave; one MFMA wave and one vector wave per SIMD (vector stream = N instructions per MFMA)\n");
    run<0>("v_fma_f32 x3", out, ticks, 3);
    run<0>("v_fma_f32 x4", out, ticks, 4);
    run<1>("v_pk_fma_f32 x3", out, ticks, 3);
    run<2>("v_cvt_pk_f16 x3", out, ticks, 3);
    run<3>("v_fma_mixlo x3", out, ticks, 3);
    run<4>("v_med3_f32 x3", out, ticks, 3);
    run<5>("v_mul_lo_u32 x1", out, ticks, 1);
    run<6>("v_xor_b32 x3", out, ticks, 3);
    run<7>("second mfma x1", out, ticks, 1);
    printf("\naggregate rates, up to four waves per SIMD\n");
    const int combos[][2] = { {1, 0}, {3, 0}, {7, 0}, {15, 0}, {0, 1}, {0, 3}, {0, 7}, {0, 15}, {1, 2}, {1, 6}, {1, 14}, {3, 4}, {3, 12}, {7, 8} };
    for (auto &c : combos) run_slots(out, ticks, c[0], c[1]);
    printf("\nMFMA and vector instructions interleaved in ONE wave\n");
    run_interleaved<0>(out, ticks);
    run_interleaved<1>(out, ticks);
    run_interleaved<2>(out, ticks);
    run_interleaved<3>(out, ticks);
    run_interleaved<4>(out, ticks);
    run_interleaved<6>(out, ticks);
    return 0;
}
