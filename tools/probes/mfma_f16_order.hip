// Probe: WHAT does an fp16-operand / fp32-accumulate MFMA of gfx950 compute, bit for bit?
//
// Background: the half-precision field kernels (field_half.hip) run their layers on
// v_mfma_f32_16x16x16_f16 pairs (or one v_mfma_f32_16x16x32_f16); their sample counts differ from the oracle's
// fp16-operand mode by a few in 4.67 M because the hardware's summation inside one instruction is not the oracle's
// sequential fmaf chain.  This program only RECORDS the hardware: tiles of operands come from a file written by
// mfma_f16_order.py (random, sparse, cancelling, subnormal ... families), every tile is pushed through
//   v0: one v_mfma_f32_16x16x16_f16 over k = 0..15           (lane group g = lane>>4 holds k = 4g..4g+3)
//   v1: two chained v_mfma_f32_16x16x16_f16, k = 0..15 then k = 16..31
//   v2: one v_mfma_f32_16x16x32_f16 over k = 0..31           (lane group g holds k = 8g..8g+7)
//   v3: the pair form of field_half_device.hpp::mfma_k32: halves 0..3 of each lane's eight, then halves 4..7
//       (so the first instruction sees k = 8g..8g+3 as ITS k = 4g..4g+3)
// and the 256 results of each are written back.  The model fitting happens offline, on the CPU
// (mfma_f16_order.py fit): candidate summation orders / internal widths / roundings are evaluated in exact integer
// arithmetic against these records.
//
//   build: hipcc --offload-arch=gfx950 -O2 -o tools/probes/mfma_f16_order tools/probes/mfma_f16_order.hip
//   run:   tools/probes/mfma_f16_order in.bin out.bin
// in.bin : int32 n_tiles, then per tile  A u16[16][32] (row, k), B u16[32][16] (k, col), C f32[16][16] (row, col)
// out.bin: int32 n_tiles, then per tile  D f32[4][16][16] (variant, row, col)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct TileIn { uint16_t a[16][32]; uint16_t b[32][16]; float c[16][16]; };
struct TileOut { float d[4][16][16]; };

__device__ __forceinline__ _Float16 bits2h(uint16_t u) { return __builtin_bit_cast(_Float16, u); }

__global__ __launch_bounds__(64) void probe(const TileIn *__restrict__ in, TileOut *__restrict__ out, int n)
{
    const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
    for (int t = blockIdx.x; t < n; t += gridDim.x) {
        const TileIn &T = in[t];
        f4 acc;
        for (int r = 0; r < 4; ++r) acc[r] = T.c[4 * g + r][c];
        // operands in the x16 layout, k-block kb: lane holds k = 16kb + 4g + e
        h4 a16[2], b16[2];
        for (int kb = 0; kb < 2; ++kb)
            for (int e = 0; e < 4; ++e) {
                a16[kb][e] = bits2h(T.a[c][16 * kb + 4 * g + e]);
                b16[kb][e] = bits2h(T.b[16 * kb + 4 * g + e][c]);
            }
        // operands in the x32 layout: lane holds k = 8g + e
        h8 a32, b32;
        for (int e = 0; e < 8; ++e) {
            a32[e] = bits2h(T.a[c][8 * g + e]);
            b32[e] = bits2h(T.b[8 * g + e][c]);
        }
        f4 d0 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16[0], b16[0], acc, 0, 0, 0);
        f4 d1 = __builtin_amdgcn_mfma_f32_16x16x16f16(a16[1], b16[1], d0, 0, 0, 0);
        f4 d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a32, b32, acc, 0, 0, 0);
        const h4 al = { a32[0], a32[1], a32[2], a32[3] }, ah = { a32[4], a32[5], a32[6], a32[7] };
        const h4 bl = { b32[0], b32[1], b32[2], b32[3] }, bh = { b32[4], b32[5], b32[6], b32[7] };
        f4 d3 = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bl, acc, 0, 0, 0);
        d3 = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, d3, 0, 0, 0);
        for (int r = 0; r < 4; ++r) {
            out[t].d[0][4 * g + r][c] = d0[r];
            out[t].d[1][4 * g + r][c] = d1[r];
            out[t].d[2][4 * g + r][c] = d2[r];
            out[t].d[3][4 * g + r][c] = d3[r];
        }
    }
}

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 1; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    int32_t n = 0;
    if (fread(&n, 4, 1, f) != 1 || n <= 0 || n > (1 << 20)) { fprintf(stderr, "bad tile count\n"); return 1; }
    std::vector<TileIn> hin((size_t)n);
    if (fread(hin.data(), sizeof(TileIn), (size_t)n, f) != (size_t)n) { fprintf(stderr, "short input\n"); return 1; }
    fclose(f);
    TileIn *din = nullptr;
    TileOut *dout = nullptr;
    HIP_OK(hipMalloc(&din, sizeof(TileIn) * (size_t)n));
    HIP_OK(hipMalloc(&dout, sizeof(TileOut) * (size_t)n));
    HIP_OK(hipMemcpy(din, hin.data(), sizeof(TileIn) * (size_t)n, hipMemcpyHostToDevice));
    const int blocks = n < 4096 ? n : 4096;
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, din, dout, n);
    HIP_OK(hipGetLastError());
    HIP_OK(hipDeviceSynchronize());
    std::vector<TileOut> hout((size_t)n);
    HIP_OK(hipMemcpy(hout.data(), dout, sizeof(TileOut) * (size_t)n, hipMemcpyDeviceToHost));
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 1; }
    fwrite(&n, 4, 1, f);
    fwrite(hout.data(), sizeof(TileOut), (size_t)n, f);
    fclose(f);
    printf("mfma_f16_order: %d tiles -> %s\n", n, argv[2]);
    return 0;
}
