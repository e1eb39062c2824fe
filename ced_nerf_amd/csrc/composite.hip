// Front-to-back alpha compositing over ray-packed samples.
// Replaces nerfacc.render_weight_from_density / render_transmittance_from_density /
// accumulate_along_rays(_) / render_visibility_from_density (un-vendored CUDA ops; call sites
// cednerf/render.py:52-54,81-87,158-169 and cednerf/utils.py:115-125,274-299) and restates the
// Taichi kernel cednerf/taichi_kernel/volume_render_test.py:4-59.  One lane owns one ray and
// walks its samples in order, so every per-ray sum has the oracle's summation order (bit-exact
// transmittance => bit-exact visibility masks and termination decisions).  These passes stream
// ~30 B per sample; they are bandwidth-trivial next to the field kernel.
#include <cfloat>

#include "ced_common.hpp"

namespace ced {

__global__ __launch_bounds__(256) void render_weights_kernel(int64_t n_rays, const int64_t *__restrict__ packed,
                                                             const float *__restrict__ t0, const float *__restrict__ t1,
                                                             const float *__restrict__ sig,
                                                             const float *__restrict__ prefix, float *__restrict__ w,
                                                             float *__restrict__ tr, float *__restrict__ al)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int64_t s0 = packed[2 * r], cnt = packed[2 * r + 1];
    float acc = 0.0f;
    for (int64_t i = s0; i < s0 + cnt; ++i) {
        float sd = sig[i] * (t1[i] - t0[i]);
        float a = 1.0f - det_expf(-sd);
        float t = det_expf(-acc);
        if (prefix) t = t * prefix[i];
        if (al) al[i] = a;
        if (tr) tr[i] = t;
        if (w) w[i] = t * a;
        acc = acc + sd;
    }
}

__global__ __launch_bounds__(256) void accumulate_kernel(int64_t n_rays, const int64_t *__restrict__ packed,
                                                         const float *__restrict__ w, const float *__restrict__ v,
                                                         int C, float *__restrict__ out)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int64_t s0 = packed[2 * r], cnt = packed[2 * r + 1];
    if (cnt == 0) return;
    for (int c = 0; c < C; ++c) {
        float acc = out[r * C + c];
        for (int64_t i = s0; i < s0 + cnt; ++i) acc = acc + (v ? w[i] * v[i * C + c] : w[i]);
        out[r * C + c] = acc;
    }
}

__global__ __launch_bounds__(256) void visibility_kernel(int64_t n_rays, const int64_t *__restrict__ packed,
                                                         const float *__restrict__ t0, const float *__restrict__ t1,
                                                         const float *__restrict__ sig, float eps, float alpha_thre,
                                                         uint8_t *__restrict__ mask)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int64_t s0 = packed[2 * r], cnt = packed[2 * r + 1];
    float acc = 0.0f;
    for (int64_t i = s0; i < s0 + cnt; ++i) {
        float sd = sig[i] * (t1[i] - t0[i]);
        float a = 1.0f - det_expf(-sd);
        float t = det_expf(-acc);
        bool vis = t >= eps;
        if (alpha_thre > 0.0f) vis = vis && (a >= alpha_thre);
        mask[i] = vis ? 1 : 0;
        acc = acc + sd;
    }
}

// weights (prefix_trans = 1 - opacity[ray]) + rgb / opacity / depth accumulation, one pass.
// Same operation order as the unfused sequence of cednerf/utils.py:274-299.
__global__ __launch_bounds__(256) void composite_prefix_kernel(int64_t n_rays, const int64_t *__restrict__ packed,
                                                               const float *__restrict__ t0,
                                                               const float *__restrict__ t1,
                                                               const float *__restrict__ sig,
                                                               const float *__restrict__ rgbs, float *__restrict__ rgb,
                                                               float *__restrict__ opacity, float *__restrict__ depth)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    int64_t s0 = packed[2 * r], cnt = packed[2 * r + 1];
    if (cnt == 0) return;
    float op = opacity[r];
    const float prefix = 1.0f - op;
    float c0 = rgb[3 * r], c1 = rgb[3 * r + 1], c2 = rgb[3 * r + 2], dp = depth[r];
    float acc = 0.0f;
    for (int64_t i = s0; i < s0 + cnt; ++i) {
        float ts = t0[i], te = t1[i];
        float sd = sig[i] * (te - ts);
        float a = 1.0f - det_expf(-sd);
        float t = det_expf(-acc) * prefix;
        float w = t * a;
        c0 = c0 + w * rgbs[3 * i];
        c1 = c1 + w * rgbs[3 * i + 1];
        c2 = c2 + w * rgbs[3 * i + 2];
        op = op + w;
        dp = dp + w * ((ts + te) / 2.0f);
        acc = acc + sd;
    }
    rgb[3 * r] = c0; rgb[3 * r + 1] = c1; rgb[3 * r + 2] = c2;
    opacity[r] = op;
    depth[r] = dp;
}

// composite_prefix + the ray bookkeeping of cednerf/utils.py:301-307 (mask update, alive count, sample count)
__global__ __launch_bounds__(256) void composite_step_kernel(int64_t n_rays, const int64_t *__restrict__ packed,
                                                             const float *__restrict__ t0, const float *__restrict__ t1,
                                                             const float *__restrict__ sig,
                                                             const float *__restrict__ rgbs, float *__restrict__ rgb,
                                                             float *__restrict__ opacity, float *__restrict__ depth,
                                                             float opc_thres, int n_samples_iter,
                                                             uint8_t *__restrict__ ray_mask,
                                                             unsigned long long *__restrict__ stats)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t cnt = 0;
    bool alive = false;
    if (r < n_rays) {
        int64_t s0 = packed[2 * r];
        cnt = packed[2 * r + 1];
        float op = opacity[r];
        if (cnt > 0) {
            const float prefix = 1.0f - op;
            float c0 = rgb[3 * r], c1 = rgb[3 * r + 1], c2 = rgb[3 * r + 2], dp = depth[r];
            float acc = 0.0f;
            for (int64_t i = s0; i < s0 + cnt; ++i) {
                float ts = t0[i], te = t1[i];
                float sd = sig[i] * (te - ts);
                float a = 1.0f - det_expf(-sd);
                float t = det_expf(-acc) * prefix;
                float w = t * a;
                c0 = c0 + w * rgbs[3 * i];
                c1 = c1 + w * rgbs[3 * i + 1];
                c2 = c2 + w * rgbs[3 * i + 2];
                op = op + w;
                dp = dp + w * ((ts + te) / 2.0f);
                acc = acc + sd;
            }
            rgb[3 * r] = c0; rgb[3 * r + 1] = c1; rgb[3 * r + 2] = c2;
            opacity[r] = op;
            depth[r] = dp;
        }
        alive = (op <= opc_thres) && (cnt == (int64_t)n_samples_iter);
        ray_mask[r] = alive ? 1 : 0;
    }
    // wave-level reduction, one atomic pair per wave
    const unsigned long long ballot = __ballot(alive);
    int c = (int)cnt;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off, 64);
    if ((threadIdx.x & 63) == 0) {
        const int n_alive = __builtin_popcountll(ballot);
        if (n_alive) atomicAdd(&stats[0], (unsigned long long)n_alive);
        if (c) atomicAdd(&stats[1], (unsigned long long)c);
    }
}

__global__ __launch_bounds__(256) void composite_test_kernel(int64_t n_alive, const float *__restrict__ sigmas,
                                                             const float *__restrict__ rgbs,
                                                             const float *__restrict__ t_start,
                                                             const float *__restrict__ t_end,
                                                             const int64_t *__restrict__ pack, int64_t *alive,
                                                             float T_thr, float a_thr, float *opacity, float *depth,
                                                             float *rgb)
{
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_alive) return;
    int64_t start = pack[2 * n], steps = pack[2 * n + 1];
    int64_t ray = alive[n];
    if (steps == 0) { alive[n] = -1; return; }
    float T = 1.0f - opacity[ray];
    float c0 = 0.0f, c1 = 0.0f, c2 = 0.0f, dacc = 0.0f, oacc = 0.0f;
    for (int64_t s = 0; s < steps; ++s) {
        int64_t i = start + s;
        float delta = t_end[i] - t_start[i];
        float a = 1.0f - det_expf(-sigmas[i] * delta);
        if (a > a_thr) {
            float w = a * T;
            float tmid = (t_start[i] + t_end[i]) / 2.0f;
            c0 = c0 + w * rgbs[3 * i];
            c1 = c1 + w * rgbs[3 * i + 1];
            c2 = c2 + w * rgbs[3 * i + 2];
            dacc = dacc + w * tmid;
            oacc = oacc + w;
            T = T * (1.0f - a);
            if (T <= T_thr) { alive[n] = -1; break; }
        }
    }
    rgb[3 * ray] += c0; rgb[3 * ray + 1] += c1; rgb[3 * ray + 2] += c2;
    depth[ray] += dacc;
    opacity[ray] += oacc;
}

__global__ __launch_bounds__(256) void finalize_kernel(int64_t n_rays, const float *__restrict__ bkgd,
                                                       float *__restrict__ rgb, const float *__restrict__ opacity,
                                                       float *__restrict__ depth)
{
    int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    float op = opacity[r];
    if (bkgd) {
        float rem = 1.0f - op;
        rgb[3 * r] = rgb[3 * r] + bkgd[0] * rem;
        rgb[3 * r + 1] = rgb[3 * r + 1] + bkgd[1] * rem;
        rgb[3 * r + 2] = rgb[3 * r + 2] + bkgd[2] * rem;
    }
    depth[r] = depth[r] / __builtin_fmaxf(op, FLT_EPSILON);
}

static inline dim3 grid_for(int64_t n) { return dim3((unsigned)((n + 255) / 256)); }

// Backward of the training-time compositing: d(colors, opacities, depths)/d(sigmas, rgbs) of
// cednerf/render.py:158-169.  One lane per ray, two sweeps: forward for the total optical depth, then backward
// with the suffix sum S_i = sum_{k>i} g_k w_k:
//   g_i = <d_color, rgb_i> + d_opacity + d_depth * t_mid_i,   d rgb_i = w_i d_color,
//   d sigma_i = dt_i * (g_i (T_i - w_i) - S_i).
__global__ __launch_bounds__(256) void composite_backward_kernel(int64_t n_rays, const int64_t *__restrict__ packed,
                                                                 const float *__restrict__ t0, const float *__restrict__ t1,
                                                                 const float *__restrict__ sig, const float *__restrict__ rgbs,
                                                                 const float *__restrict__ d_color,
                                                                 const float *__restrict__ d_opacity,
                                                                 const float *__restrict__ d_depth,
                                                                 float *__restrict__ d_sig, float *__restrict__ d_rgbs)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rays) return;
    const int64_t s0 = packed[2 * r], cnt = packed[2 * r + 1];
    if (cnt <= 0) return;
    const float dc0 = d_color[3 * r], dc1 = d_color[3 * r + 1], dc2 = d_color[3 * r + 2];
    const float dop = d_opacity ? d_opacity[r] : 0.0f, ddp = d_depth ? d_depth[r] : 0.0f;
    float total = 0.0f;
    for (int64_t i = s0; i < s0 + cnt; ++i) total = total + sig[i] * (t1[i] - t0[i]);
    float acc_after = total, suffix = 0.0f;
    for (int64_t i = s0 + cnt - 1; i >= s0; --i) {
        const float ts = t0[i], te = t1[i];
        const float dt = te - ts;
        const float sd = sig[i] * dt;
        // optical depth before the sample; the first sample's is exactly 0 (no cancellation residue in T_0 = 1)
        const float acc_before = (i == s0) ? 0.0f : (acc_after - sd);
        const float T = __expf(-acc_before);
        const float a = 1.0f - __expf(-sd);
        const float w = T * a;
        const float g = ((dc0 * rgbs[3 * i] + dc1 * rgbs[3 * i + 1]) + dc2 * rgbs[3 * i + 2]) + dop + ddp * ((ts + te) * 0.5f);
        d_sig[i] = dt * (g * (T - w) - suffix);
        d_rgbs[3 * i] = w * dc0;
        d_rgbs[3 * i + 1] = w * dc1;
        d_rgbs[3 * i + 2] = w * dc2;
        suffix = suffix + g * w;
        acc_after = acc_before;
    }
}

// reduce_along_rays (cednerf/render.py:8-39): out[ray, c] (+)= weights[i, c or 0] * values[i, c] for every sample i of
// the ray -- torch's scatter_reduce_ into a zero tensor.  Like that op on the GPU, arbitrary (unsorted) ray indices
// are accumulated with float atomics; consecutive samples of one ray -- the ray-packed case of every caller -- are
// summed across the wave first, so a run costs one atomic.  counts[ray] += 1 per sample serves reduce="mean".
__global__ __launch_bounds__(256) void reduce_along_rays_kernel(int64_t n, int n_ch, const int64_t *__restrict__ ray_indices,
                                                                const float *__restrict__ values,
                                                                const float *__restrict__ weights, int w_ch, int64_t n_rays,
                                                                float *__restrict__ out, int32_t *__restrict__ counts)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = i < n;
    const int64_t r = in ? ray_indices[i] : -1;
    const bool ok = in && r >= 0 && r < n_rays;
    // runs of equal ray index inside the wave: the last lane of a run adds the run's sum
    const int64_t r_prev = __shfl_up(r, 1, 64), r_next = __shfl_down(r, 1, 64);
    const bool head = lane == 0 || r_prev != r;
    const bool tail = lane == 63 || r_next != r;
    for (int c = 0; c < n_ch; ++c) {
        float v = ok ? values[i * n_ch + c] : 0.0f;
        if (ok && weights) v = weights[i * w_ch + (w_ch > 1 ? c : 0)] * v;
        int h = head ? 1 : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float v_up = __shfl_up(v, d, 64);
            const int h_up = __shfl_up(h, d, 64);
            if (lane >= d && !h) { v += v_up; h |= h_up; }
        }
        if (ok && tail) unsafeAtomicAdd(out + r * n_ch + c, v);
    }
    if (counts) {
        int k = 1, h = head ? 1 : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int k_up = __shfl_up(k, d, 64);
            const int h_up = __shfl_up(h, d, 64);
            if (lane >= d && !h) { k += k_up; h |= h_up; }
        }
        if (ok && tail) atomicAdd(counts + r, k);
    }
}

// reduce="mean" of scatter_reduce_ with include_self=True (the default the reference relies on): the zero the output
// started from counts as one element
__global__ __launch_bounds__(256) void reduce_mean_finish_kernel(int64_t n_rays, int n_ch, const int32_t *__restrict__ counts,
                                                                 float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rays * n_ch) return;
    out[i] = out[i] / (float)(counts[i / n_ch] + 1);
}

}  // namespace ced

extern "C" int ced_composite_backward(int64_t n_rays, const int64_t *packed_info, const float *t_starts, const float *t_ends,
                                      const float *sigmas, const float *rgbs, const float *d_color, const float *d_opacity,
                                      const float *d_depth, float *d_sigmas, float *d_rgbs, void *stream)
{
    CED_REQUIRE(n_rays >= 0, "composite_backward: n_rays < 0");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(packed_info && t_starts && t_ends && sigmas && rgbs && d_color && d_sigmas && d_rgbs,
                "composite_backward: null pointer");
    hipLaunchKernelGGL(ced::composite_backward_kernel, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, n_rays, packed_info, t_starts, t_ends, sigmas, rgbs, d_color, d_opacity, d_depth,
                       d_sigmas, d_rgbs);
    return ced::check_launch("composite_backward");
}

namespace ced {
}  // namespace ced

extern "C" int ced_render_weights(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                                  const float *t_ends, const float *sigmas, const float *prefix_trans, float *weights,
                                  float *trans, float *alphas, void *stream)
{
    CED_REQUIRE(n_rays >= 0, "render_weights: n_rays < 0");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(packed_info && t_starts && t_ends && sigmas, "render_weights: null pointer");
    hipLaunchKernelGGL(ced::render_weights_kernel, ced::grid_for(n_rays), dim3(256), 0, (hipStream_t)stream, n_rays,
                       packed_info, t_starts, t_ends, sigmas, prefix_trans, weights, trans, alphas);
    return ced::check_launch("render_weights");
}

extern "C" int ced_accumulate_along_rays(int64_t n_rays, const int64_t *packed_info, const float *weights,
                                         const float *values, int32_t n_channels, float *out, void *stream)
{
    CED_REQUIRE(n_rays >= 0 && n_channels >= 1, "accumulate_along_rays: bad sizes");
    CED_REQUIRE(values || n_channels == 1, "accumulate_along_rays: values == NULL requires one channel");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(packed_info && weights && out, "accumulate_along_rays: null pointer");
    hipLaunchKernelGGL(ced::accumulate_kernel, ced::grid_for(n_rays), dim3(256), 0, (hipStream_t)stream, n_rays,
                       packed_info, weights, values, (int)n_channels, out);
    return ced::check_launch("accumulate_along_rays");
}

extern "C" int ced_visibility_mask(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                                   const float *t_ends, const float *sigmas, float early_stop_eps, float alpha_thre,
                                   uint8_t *mask, void *stream)
{
    CED_REQUIRE(n_rays >= 0, "visibility_mask: n_rays < 0");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(packed_info && t_starts && t_ends && sigmas && mask, "visibility_mask: null pointer");
    hipLaunchKernelGGL(ced::visibility_kernel, ced::grid_for(n_rays), dim3(256), 0, (hipStream_t)stream, n_rays,
                       packed_info, t_starts, t_ends, sigmas, early_stop_eps, alpha_thre, mask);
    return ced::check_launch("visibility_mask");
}

extern "C" int ced_composite_prefix(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                                    const float *t_ends, const float *sigmas, const float *rgbs, float *rgb,
                                    float *opacity, float *depth, void *stream)
{
    CED_REQUIRE(n_rays >= 0, "composite_prefix: n_rays < 0");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(packed_info && t_starts && t_ends && sigmas && rgbs && rgb && opacity && depth,
                "composite_prefix: null pointer");
    hipLaunchKernelGGL(ced::composite_prefix_kernel, ced::grid_for(n_rays), dim3(256), 0, (hipStream_t)stream, n_rays,
                       packed_info, t_starts, t_ends, sigmas, rgbs, rgb, opacity, depth);
    return ced::check_launch("composite_prefix");
}

extern "C" int ced_composite_step(int64_t n_rays, const int64_t *packed_info, const float *t_starts,
                                  const float *t_ends, const float *sigmas, const float *rgbs, float *rgb,
                                  float *opacity, float *depth, float opc_thres, int32_t n_samples_iter,
                                  uint8_t *ray_mask, int64_t *stats, void *stream)
{
    CED_REQUIRE(n_rays >= 0, "composite_step: n_rays < 0");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(packed_info && rgb && opacity && depth && ray_mask && stats, "composite_step: null pointer");
    hipLaunchKernelGGL(ced::composite_step_kernel, ced::grid_for(n_rays), dim3(256), 0, (hipStream_t)stream, n_rays,
                       packed_info, t_starts, t_ends, sigmas, rgbs, rgb, opacity, depth, opc_thres,
                       (int)n_samples_iter, ray_mask, reinterpret_cast<unsigned long long *>(stats));
    return ced::check_launch("composite_step");
}

extern "C" int ced_composite_test(int64_t n_alive, const float *sigmas, const float *rgbs, const float *t_start,
                                  const float *t_end, const int64_t *pack_info, int64_t *alive_indices,
                                  float T_threshold, float alpha_threshold, float *opacity, float *depth, float *rgb,
                                  void *stream)
{
    CED_REQUIRE(n_alive >= 0, "composite_test: n_alive < 0");
    if (n_alive == 0) return CED_OK;
    CED_REQUIRE(sigmas && rgbs && t_start && t_end && pack_info && alive_indices && opacity && depth && rgb,
                "composite_test: null pointer");
    hipLaunchKernelGGL(ced::composite_test_kernel, ced::grid_for(n_alive), dim3(256), 0, (hipStream_t)stream, n_alive,
                       sigmas, rgbs, t_start, t_end, pack_info, alive_indices, T_threshold, alpha_threshold, opacity,
                       depth, rgb);
    return ced::check_launch("composite_test");
}

extern "C" int ced_finalize_pixels(int64_t n_rays, const float *bkgd, float *rgb, const float *opacity, float *depth,
                                   void *stream)
{
    CED_REQUIRE(n_rays >= 0, "finalize_pixels: n_rays < 0");
    if (n_rays == 0) return CED_OK;
    CED_REQUIRE(rgb && opacity && depth, "finalize_pixels: null pointer");
    hipLaunchKernelGGL(ced::finalize_kernel, ced::grid_for(n_rays), dim3(256), 0, (hipStream_t)stream, n_rays, bkgd, rgb,
                       opacity, depth);
    return ced::check_launch("finalize_pixels");
}

extern "C" int ced_reduce_along_rays(int64_t n_samples, const int64_t *ray_indices, const float *values, int32_t n_channels,
                                     const float *weights, int32_t weight_channels, int64_t n_rays, int32_t mean,
                                     float *out, int32_t *counts_workspace, void *stream)
{
    CED_REQUIRE(n_samples >= 0 && n_rays >= 0 && n_channels >= 1, "reduce_along_rays: bad sizes");
    CED_REQUIRE(!weights || weight_channels == 1 || weight_channels == n_channels,
                "reduce_along_rays: weights must have 1 or n_channels columns");
    CED_REQUIRE(out != nullptr || n_rays == 0, "reduce_along_rays: null output");
    CED_REQUIRE(!mean || counts_workspace, "reduce_along_rays: reduce=\"mean\" needs a counts workspace of n_rays int32");
    if (n_rays == 0) return CED_OK;
    if (hipMemsetAsync(out, 0, (size_t)n_rays * n_channels * 4, (hipStream_t)stream) != hipSuccess ||
        (mean && hipMemsetAsync(counts_workspace, 0, (size_t)n_rays * 4, (hipStream_t)stream) != hipSuccess))
        return ced::check_launch("reduce_along_rays (memset)");
    if (n_samples > 0) {
        CED_REQUIRE(ray_indices && values, "reduce_along_rays: null pointer");
        hipLaunchKernelGGL(ced::reduce_along_rays_kernel, ced::grid_for(n_samples), dim3(256), 0, (hipStream_t)stream, n_samples,
                           (int)n_channels, ray_indices, values, weights, (int)weight_channels, n_rays, out,
                           mean ? counts_workspace : (int32_t *)nullptr);
    }
    if (mean)
        hipLaunchKernelGGL(ced::reduce_mean_finish_kernel, ced::grid_for(n_rays * n_channels), dim3(256), 0, (hipStream_t)stream,
                           n_rays, (int)n_channels, counts_workspace, out);
    return ced::check_launch("reduce_along_rays");
}
