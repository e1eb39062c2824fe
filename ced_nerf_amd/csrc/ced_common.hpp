// Shared host/device helpers for libcednerf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/cednerf_hip.h"

namespace ced {

void set_error(const char *fmt, ...);

inline int check_launch(const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return CED_E_LAUNCH;
    }
    return CED_OK;
}

#define CED_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ced::set_error(__VA_ARGS__);       \
            return CED_E_INVALID;              \
        }                                      \
    } while (0)

constexpr int kWave = 64;

// ----------------------------------------------------------------------------------------------
// Deterministic scalar math (arithmetic contract).  Compiled with -ffp-contract=off:
// a fused multiply-add happens only where __builtin_fmaf is written.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float pow2i(int n) { return __uint_as_float((uint32_t)(n + 127) << 23); }

// exp(x): n = rint(x*log2e), two-step Cody-Waite reduction, degree-6 Taylor, two-step 2^n scaling.
__device__ __forceinline__ float det_expf(float x)
{
    float n = __builtin_rintf(x * 1.44269502162933349609375f);
    float r = __builtin_fmaf(-n, 0.693145751953125f, x);
    r = __builtin_fmaf(-n, 1.428606765330187045e-06f, r);
    float p = 1.38888892e-3f;
    p = __builtin_fmaf(p, r, 8.33333377e-3f);
    p = __builtin_fmaf(p, r, 4.16666679e-2f);
    p = __builtin_fmaf(p, r, 1.66666672e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    int ni = (int)n;
    int n1 = ni / 2, n2 = ni - n1;
    float v = (p * pow2i(n1)) * pow2i(n2);
    v = (x > 88.72283935546875f) ? __builtin_inff() : v;
    v = (x < -103.97283935546875f) ? 0.0f : v;
    return (x != x) ? x : v;
}

__device__ __forceinline__ float sin_kernel(float x)
{
    float x2 = x * x;
    float p = 2.75573192e-6f;
    p = __builtin_fmaf(p, x2, -1.98412701e-4f);
    p = __builtin_fmaf(p, x2, 8.33333377e-3f);
    p = __builtin_fmaf(p, x2, -1.66666672e-1f);
    return __builtin_fmaf(x * x2, p, x);
}
__device__ __forceinline__ float cos_kernel(float x)
{
    float x2 = x * x;
    float p = -2.75573188e-7f;
    p = __builtin_fmaf(p, x2, 2.48015876e-5f);
    p = __builtin_fmaf(p, x2, -1.38888892e-3f);
    p = __builtin_fmaf(p, x2, 4.16666679e-2f);
    p = __builtin_fmaf(p, x2, -0.5f);
    return __builtin_fmaf(p, x2, 1.0f);
}
// q mod 4: 0 sin, 1 cos, 2 -sin, 3 -cos
__device__ __forceinline__ float quadrant_select(int q, float x)
{
    float s = sin_kernel(x), c = cos_kernel(x);
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
// sin(pi*y + phase*pi/2) with exact reduction (tcnn Frequency encoding terms)
__device__ __forceinline__ float det_sinpi_phase(float y, int phase)
{
    float n = __builtin_rintf(y + y);
    float r = y - 0.5f * n;
    float x = 3.14159274101257324f * r;
    return quadrant_select((((int)n) & 3) + phase, x);
}
// Both phases of one Frequency term at once: p0 = sin(pi*y), p1 = sin(pi*y + pi/2) -- the same reduction, the same two
// polynomials and the same selection as det_sinpi_phase(y, 0) and det_sinpi_phase(y, 1), evaluated once.
__device__ __forceinline__ void det_sinpi_both(float y, float &p0, float &p1)
{
    float n = __builtin_rintf(y + y);
    float r = y - 0.5f * n;
    float x = 3.14159274101257324f * r;
    const float s = sin_kernel(x), c = cos_kernel(x);
    const int q = ((int)n) & 3;
    const float v0 = (q & 1) ? c : s;
    p0 = (q & 2) ? -v0 : v0;
    const int q1 = q + 1;
    const float v1 = (q1 & 1) ? c : s;
    p1 = (q1 & 2) ? -v1 : v1;
}

// sin(x), two-step Cody-Waite reduction by pi/2
__device__ __forceinline__ float det_sinf(float x)
{
    float n = __builtin_rintf(x * 0.636619746685028076f);
    float r = __builtin_fmaf(-n, 1.57079625129699707f, x);
    r = __builtin_fmaf(-n, 7.54978941586159635e-08f, r);
    return quadrant_select(((int)n) & 3, r);
}

__device__ __forceinline__ float half_bits_to_float(uint16_t h)
{
    return (float)__builtin_bit_cast(_Float16, h);
}

}  // namespace ced
