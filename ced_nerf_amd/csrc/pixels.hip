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

// Rendered pixels arrive in the order the rays were marched in (8x8-tile order; with several GPUs every rank's shard
// of it, all-gathered): row i of the source goes to raster pixel dest[i] (rows with dest >= n_pixels are padding).
// One pass writes the three raster images -- and, when asked, the 8-bit colour frame as well, so that a video frame
// never exists as a float raster image in memory.
__global__ __launch_bounds__(256) void scatter_pixels_kernel(int64_t n_rows, const float *__restrict__ src_rgb, int s_rgb,
                                                             const float *__restrict__ src_op, int s_op,
                                                             const float *__restrict__ src_dp, int s_dp,
                                                             const int64_t *__restrict__ dest, int64_t n_pixels,
                                                             float *__restrict__ rgb, float *__restrict__ opacity,
                                                             float *__restrict__ depth, uint8_t *__restrict__ rgb8,
                                                             int width, int flip_w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rows) return;
    const int64_t p = dest[i];
    if (p < 0 || p >= n_pixels) return;
    const float r = src_rgb[i * s_rgb], g = src_rgb[i * s_rgb + 1], b = src_rgb[i * s_rgb + 2];
    if (rgb) { rgb[3 * p] = r; rgb[3 * p + 1] = g; rgb[3 * p + 2] = b; }
    if (opacity) opacity[p] = src_op[i * s_op];
    if (depth) depth[p] = src_dp[i * s_dp];
    if (rgb8) {
        // frames are [F, H, W]: flip inside the pixel's own row
        const int64_t row = p / width;
        const int x = (int)(p - row * width);
        const int64_t q = row * width + (flip_w ? width - 1 - x : x);
        rgb8[3 * q] = to_u8(r * 255.0f); rgb8[3 * q + 1] = to_u8(g * 255.0f); rgb8[3 * q + 2] = to_u8(b * 255.0f);
    }
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

extern "C" int ced_scatter_pixels(int64_t n_rows, const float *src_rgb, int32_t stride_rgb, const float *src_opacity,
                                  int32_t stride_opacity, const float *src_depth, int32_t stride_depth, const int64_t *dest,
                                  int64_t n_pixels, float *rgb, float *opacity, float *depth, uint8_t *rgb8, int32_t width,
                                  int32_t flip_w, void *stream)
{
    CED_REQUIRE(n_rows >= 0 && n_pixels >= 0, "scatter_pixels: negative size");
    if (n_rows == 0) return CED_OK;
    CED_REQUIRE(src_rgb && dest && stride_rgb >= 3, "scatter_pixels: null source / index or stride_rgb < 3");
    CED_REQUIRE((!opacity || (src_opacity && stride_opacity >= 1)) && (!depth || (src_depth && stride_depth >= 1)),
                "scatter_pixels: an output without its source");
    CED_REQUIRE(!rgb8 || width >= 1, "scatter_pixels: rgb8 needs the image width");
    hipLaunchKernelGGL(ced::scatter_pixels_kernel, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_rows, src_rgb, (int)stride_rgb, src_opacity, (int)stride_opacity, src_depth, (int)stride_depth, dest,
                       n_pixels, rgb, opacity, depth, rgb8, (int)width, (int)flip_w);
    return ced::check_launch("scatter_pixels");
}
