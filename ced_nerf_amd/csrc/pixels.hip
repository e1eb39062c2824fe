// Consumer side of a rendered frame: the 8-bit images the reference's video step builds on the host,
//   rgb_frames.append(np.flip(rgb.cpu().numpy() * 255, axis=1).astype(np.uint8))          train_real.py:556
//   depth = (depth - depth.min()) / (depth.max() - depth.min()); (depth * 255) -> uint8    train_real.py:38-41 (depth2img,
//   before cv2's colour-map lookup, which is a table of OpenCV's and stays on the host)
// done on the device so that a frame leaves HBM as 3 + 1 bytes per pixel instead of 20.
#include "ced_common.hpp"

namespace ced {

__device__ __forceinline__ uint8_t to_u8(float v)
{
    // numpy's float32 -> uint8 cast truncates toward zero; rendered colours and normalised depths lie in [0, 255]
    // (values outside, and NaN from a constant depth image, are clamped / mapped to 0 here)
    v = v >= 0.0f ? v : 0.0f;
    v = v <= 255.0f ? v : 255.0f;
    return (uint8_t)(int)v;
}

__global__ void frame_to_rgb8_kernel(int H, int W, const float *__restrict__ rgb, int flip_w, uint8_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // destination pixel
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const int64_t src = (int64_t)y * W + (flip_w ? W - 1 - x : x);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[3 * i + c] = to_u8(rgb[3 * src + c] * 255.0f);
}

// order-preserving map of a float onto unsigned integers (so integer atomics give the float min / max)
__device__ __forceinline__ uint32_t ordered_key(float f)
{
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float from_ordered_key(uint32_t k)
{
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void depth_range_init_kernel(uint32_t *range)
{
    range[0] = 0xffffffffu;     // min key
    range[1] = 0u;              // max key
}

__global__ __launch_bounds__(256) void depth_range_kernel(int64_t n, const float *__restrict__ depth, uint32_t *range)
{
    uint32_t lo = 0xffffffffu, hi = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t k = ordered_key(depth[i]);
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const uint32_t lo2 = __shfl_xor(lo, m), hi2 = __shfl_xor(hi, m);
        lo = lo2 < lo ? lo2 : lo;
        hi = hi2 > hi ? hi2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&range[0], lo);
        atomicMax(&range[1], hi);
    }
}

__global__ void depth_to_u8_kernel(int H, int W, const float *__restrict__ depth, const uint32_t *__restrict__ range,
                                   int flip_w, uint8_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)H * W) return;
    const int y = (int)(i / W), x = (int)(i - (int64_t)y * W);
    const int64_t src = (int64_t)y * W + (flip_w ? W - 1 - x : x);
    const float lo = from_ordered_key(range[0]), hi = from_ordered_key(range[1]);
    const float d = (depth[src] - lo) / (hi - lo);
    out[i] = to_u8(d * 255.0f);
}

}  // namespace ced

extern "C" int ced_frame_to_rgb8(int32_t height, int32_t width, const float *rgb, int32_t flip_w, uint8_t *out, void *stream)
{
    CED_REQUIRE(height >= 0 && width >= 0, "frame_to_rgb8: negative size");
    const int64_t n = (int64_t)height * width;
    if (n == 0) return CED_OK;
    CED_REQUIRE(rgb && out, "frame_to_rgb8: null pointer");
    hipLaunchKernelGGL(ced::frame_to_rgb8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (int)height, (int)width, rgb, (int)flip_w, out);
    return ced::check_launch("frame_to_rgb8");
}

extern "C" int ced_depth_to_u8(int32_t height, int32_t width, const float *depth, int32_t flip_w, uint8_t *out,
                               void *workspace /* 8 bytes */, void *stream)
{
    CED_REQUIRE(height >= 0 && width >= 0, "depth_to_u8: negative size");
    const int64_t n = (int64_t)height * width;
    if (n == 0) return CED_OK;
    CED_REQUIRE(depth && out && workspace, "depth_to_u8: null pointer");
    hipStream_t st = (hipStream_t)stream;
    uint32_t *range = (uint32_t *)workspace;
    hipLaunchKernelGGL(ced::depth_range_init_kernel, dim3(1), dim3(1), 0, st, range);
    int64_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(ced::depth_range_kernel, dim3((unsigned)blocks), dim3(256), 0, st, n, depth, range);
    hipLaunchKernelGGL(ced::depth_to_u8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (int)height, (int)width,
                       depth, (const uint32_t *)range, (int)flip_w, out);
    return ced::check_launch("depth_to_u8");
}
