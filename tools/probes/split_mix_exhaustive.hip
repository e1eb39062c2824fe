// Exhaustive check (all 2^32 fp32 bit patterns) that the fp16 remainder of the operand split,
//     lo = fp16( x - fp32( fp16(x) ) )                      (two roundings: the subtraction is exact in fp32)
// equals ONE v_fma_mixlo_f16 / v_fma_mixhi_f16 computing  fp16( hi * -1.0 + x )  straight from the packed high part.
// The subtraction x - hi is exact (hi is x rounded to 11 bits, so the difference fits 24), hence rounding it once to
// fp16 is rounding the same real number: the two forms must agree wherever fp16(x) is finite.  This probe is the proof
// by enumeration, including fp16-subnormal high parts and remainders.
// build: hipcc --offload-arch=gfx950 -O3 -o split_mix_exhaustive split_mix_exhaustive.hip ; run: ./split_mix_exhaustive
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>

typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2v __attribute__((ext_vector_type(2)));

__global__ void probe(unsigned long long *counts, uint32_t *first_bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad_lo = 0, bad_hi = 0, n_finite = 0, bad_nonfinite = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i;
        float x = __uint_as_float(bits);
        // a second value for the other half of the pair: the bit pattern rotated (any value will do)
        float y = __uint_as_float((bits << 7) | (bits >> 25));
        asm volatile("" : "+v"(x));
        asm volatile("" : "+v"(y));
        const h2 h = __builtin_convertvector(f2v{ x, y }, h2);
        const f2v back = __builtin_convertvector(h, f2v);
        const h2 ref = __builtin_convertvector(f2v{ x, y } - back, h2);
        uint32_t hp, got;
        __builtin_memcpy(&hp, &h, 4);
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(got) : "v"(hp), "v"(x));
        asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(got) : "v"(hp), "v"(y));
        uint32_t want;
        __builtin_memcpy(&want, &ref, 4);
        const bool fin_x = (hp & 0x7c00u) != 0x7c00u, fin_y = ((hp >> 16) & 0x7c00u) != 0x7c00u;
        if (fin_x) {
            ++n_finite;
            if ((want & 0xffffu) != (got & 0xffffu)) {
                if (bad_lo == 0) atomicCAS(first_bad, 0u, bits);
                ++bad_lo;
            }
        } else if ((want & 0xffffu) != (got & 0xffffu)) ++bad_nonfinite;
        if (fin_y && (want >> 16) != (got >> 16)) ++bad_hi;
    }
    atomicAdd(&counts[0], n_finite);
    atomicAdd(&counts[1], bad_lo);
    atomicAdd(&counts[2], bad_hi);
    atomicAdd(&counts[3], bad_nonfinite);
}

int main()
{
    unsigned long long *d, h[4];
    uint32_t *fb, hfb;
    hipMalloc(&d, sizeof h);
    hipMalloc(&fb, 4);
    hipMemset(d, 0, sizeof h);
    hipMemset(fb, 0, 4);
    probe<<<4096, 256>>>(d, fb);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(&hfb, fb, 4, hipMemcpyDeviceToHost);
    printf("fp32 patterns whose fp16 high part is finite: %llu of 4294967296\n", h[0]);
    printf("remainder differs, low half  (v_fma_mixlo_f16): %llu\n", h[1]);
    printf("remainder differs, high half (v_fma_mixhi_f16): %llu\n", h[2]);
    printf("differs where the high part is inf / nan (never an operand: inputs are clamped to 65504): %llu\n", h[3]);
    if (h[1]) printf("first differing pattern: 0x%08x\n", hfb);
    return (h[1] || h[2]) ? 1 : 0;
}
