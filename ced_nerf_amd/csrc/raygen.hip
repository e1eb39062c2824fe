// On-device ray generation for full frames: the camera models whose rays the
// reference builds on the host and uploads every frame (15 MB at 800x800).
//   pinhole  : datasets/dnerf_synthetic.py:191-221 and gui.py:43-86 (OpenGL or OpenCV convention);
//   hypercam : datasets/hyper_cam.py:210-252 (Camera.pixels_to_rays on get_pixel_centers(), :299-303),
//              including the 10-step Newton undistortion of :22-91, as used at datasets/hypernerf.py:172-173.
// float32 throughout, operations in the reference's order.
#include "ced_common.hpp"

namespace ced {

struct PinholeArgs {
    int width, height;
    float fx, fy, cx, cy;
    float c2w[12];          // row-major [3][4]
    float sign;             // -1 OpenGL (y up, looking down -z), +1 OpenCV
    float *origins, *viewdirs, *directions;   // [H*W,3]; directions may be NULL
};

__global__ __launch_bounds__(256) void pinhole_rays_kernel(PinholeArgs A)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)A.width * A.height) return;
    const float x = (float)(i % A.width), y = (float)(i / A.width);
    const float cam[3] = { (x - A.cx + 0.5f) / A.fx, ((y - A.cy + 0.5f) / A.fy) * A.sign, A.sign };
    float d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) d[r] = (cam[0] * A.c2w[4 * r] + cam[1] * A.c2w[4 * r + 1]) + cam[2] * A.c2w[4 * r + 2];
    const float nrm = __builtin_sqrtf((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        A.origins[3 * i + r] = A.c2w[4 * r + 3];
        A.viewdirs[3 * i + r] = d[r] / nrm;
        if (A.directions) A.directions[3 * i + r] = d[r];
    }
}

struct HyperCamArgs {
    int width, height;
    float orientation[9];   // row-major world->camera rotation
    float position[3];
    float focal, ppx, ppy, skew, aspect;
    float k1, k2, k3, p1, p2;
    int distorted;
    float *origins, *viewdirs;
};

__global__ __launch_bounds__(256) void hypercam_rays_kernel(HyperCamArgs A)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)A.width * A.height) return;
    const float px = (float)(i % A.width) + 0.5f, py = (float)(i / A.width) + 0.5f;
    float y = (py - A.ppy) / (A.focal * A.aspect);
    float x = (px - A.ppx - y * A.skew) / A.focal;
    if (A.distorted) {
        const float xd = x, yd = y;
        for (int it = 0; it < 10; ++it) {
            const float r = x * x + y * y;
            const float d = 1.0f + r * (A.k1 + r * (A.k2 + A.k3 * r));
            const float fx = d * x + 2.0f * A.p1 * x * y + A.p2 * (r + 2.0f * x * x) - xd;
            const float fy = d * y + 2.0f * A.p2 * x * y + A.p1 * (r + 2.0f * y * y) - yd;
            const float d_r = A.k1 + r * (2.0f * A.k2 + 3.0f * A.k3 * r);
            const float d_x = 2.0f * x * d_r, d_y = 2.0f * y * d_r;
            const float fx_x = d + d_x * x + 2.0f * A.p1 * y + 6.0f * A.p2 * x;
            const float fx_y = d_y * x + 2.0f * A.p1 * x + 2.0f * A.p2 * y;
            const float fy_x = d_x * y + 2.0f * A.p2 * y + 2.0f * A.p1 * x;
            const float fy_y = d + d_y * y + 2.0f * A.p2 * x + 6.0f * A.p1 * y;
            const float den = fy_x * fx_y - fx_x * fy_y;
            const float xn = fx * fy_y - fy * fx_y, yn = fy * fx_x - fx * fy_x;
            const bool ok = __builtin_fabsf(den) > 1e-9f;
            x = x + (ok ? xn / den : 0.0f);
            y = y + (ok ? yn / den : 0.0f);
        }
    }
    float l[3] = { x, y, 1.0f };
    const float ln = __builtin_sqrtf((l[0] * l[0] + l[1] * l[1]) + l[2] * l[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) l[r] = l[r] / ln;
    float w[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)      // orientation^T @ local
        w[r] = (A.orientation[r] * l[0] + A.orientation[3 + r] * l[1]) + A.orientation[6 + r] * l[2];
    const float wn = __builtin_sqrtf((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float v = w[r] / wn;          // rays_dir (hyper_cam.py:249); already unit, so viewdirs = rays_dir / |rays_dir|
        A.origins[3 * i + r] = A.position[r];
        A.viewdirs[3 * i + r] = v;
    }
}

}  // namespace ced

extern "C" int ced_generate_rays_pinhole(int32_t width, int32_t height, float fx, float fy, float cx, float cy,
                                         const float *c2w_host, int32_t opengl, float *origins, float *viewdirs,
                                         float *directions, void *stream)
{
    CED_REQUIRE(width > 0 && height > 0, "generate_rays_pinhole: bad image size");
    CED_REQUIRE(c2w_host && origins && viewdirs, "generate_rays_pinhole: null pointer");
    ced::PinholeArgs A{};
    A.width = width; A.height = height; A.fx = fx; A.fy = fy; A.cx = cx; A.cy = cy;
    for (int i = 0; i < 12; ++i) A.c2w[i] = c2w_host[i];
    A.sign = opengl ? -1.0f : 1.0f;
    A.origins = origins; A.viewdirs = viewdirs; A.directions = directions;
    const int64_t n = (int64_t)width * height;
    hipLaunchKernelGGL(ced::pinhole_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return ced::check_launch("generate_rays_pinhole");
}

extern "C" int ced_generate_rays_hypercam(int32_t width, int32_t height, const float *orientation_host,
                                          const float *position_host, float focal_length, float principal_x,
                                          float principal_y, float skew, float pixel_aspect_ratio,
                                          const float *radial3_host, const float *tangential2_host, float *origins,
                                          float *viewdirs, void *stream)
{
    CED_REQUIRE(width > 0 && height > 0, "generate_rays_hypercam: bad image size");
    CED_REQUIRE(orientation_host && position_host && origins && viewdirs, "generate_rays_hypercam: null pointer");
    ced::HyperCamArgs A{};
    A.width = width; A.height = height;
    for (int i = 0; i < 9; ++i) A.orientation[i] = orientation_host[i];
    for (int i = 0; i < 3; ++i) A.position[i] = position_host[i];
    A.focal = focal_length; A.ppx = principal_x; A.ppy = principal_y; A.skew = skew; A.aspect = pixel_aspect_ratio;
    A.k1 = radial3_host ? radial3_host[0] : 0.0f; A.k2 = radial3_host ? radial3_host[1] : 0.0f;
    A.k3 = radial3_host ? radial3_host[2] : 0.0f;
    A.p1 = tangential2_host ? tangential2_host[0] : 0.0f; A.p2 = tangential2_host ? tangential2_host[1] : 0.0f;
    A.distorted = (A.k1 != 0.0f || A.k2 != 0.0f || A.k3 != 0.0f || A.p1 != 0.0f || A.p2 != 0.0f) ? 1 : 0;
    A.origins = origins; A.viewdirs = viewdirs;
    const int64_t n = (int64_t)width * height;
    hipLaunchKernelGGL(ced::hypercam_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, A);
    return ced::check_launch("generate_rays_hypercam");
}
