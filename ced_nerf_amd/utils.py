"""Render drivers: host mirror of cednerf/utils.py (render_image :46-150, render_image_test :153-318).

Same signatures and return tuples as the reference; the per-sample work (positions, field,
weights, per-ray accumulation) runs in the HIP kernels through `ced_nerf_amd.ops`.
"""
from __future__ import annotations

import collections
import random
from typing import Optional

import numpy as np
import torch

from . import ops
from .nerfacc_api import OccGridEstimator, march_packed, ray_aabb_intersect, sort_intersections
from .render import rendering

# datasets/utils.py:8,13-15
Rays = collections.namedtuple("Rays", ("origins", "viewdirs"))


def namedtuple_map(fn, tup):
    """Apply `fn` to each element of `tup` and cast to `tup`'s namedtuple."""
    return type(tup)(*(None if x is None else fn(x) for x in tup))


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


class _TruncExp(torch.autograd.Function):
    """cednerf/utils.py:27-43: exp in float32; the backward multiplies by exp(clamp(x, max=15)) so that large
    pre-activations give finite gradients."""

    @staticmethod
    def forward(ctx, x):
        x = x.float()
        ctx.save_for_backward(x)
        return torch.exp(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.exp(torch.clamp(x, max=15))


trunc_exp = _TruncExp.apply


_EVAL_PASS_RAYS = 1 << 21     # rays per internal eval pass of render_image


def _flatten_rays(rays: Rays):
    rays_shape = rays.origins.shape
    if len(rays_shape) == 3:
        height, width, _ = rays_shape
        num_rays = height * width
        rays = namedtuple_map(lambda r: r.reshape([num_rays] + list(r.shape[2:])), rays)
    else:
        num_rays, _ = rays_shape
    return rays, rays_shape, num_rays


@torch.no_grad()
def render_image(
    radiance_field: torch.nn.Module,
    estimator: OccGridEstimator,
    rays: Rays,
    near_plane: float = 0.0,
    far_plane: float = 1e10,
    render_step_size: float = 1e-3,
    render_bkgd: Optional[torch.Tensor] = None,
    cone_angle: float = 0.0,
    alpha_thre: float = 0.0,
    test_chunk_size: int = 8192,
    timestamps: Optional[torch.Tensor] = None,
    native: Optional[bool] = None,
):
    """Render the pixels of an image (cednerf/utils.py:46-150).
    Returns (colors, opacities, depths, n_rendering_samples, extras[list per chunk]).
    In eval mode with the HIP field the pass runs on ced_render_image: the visibility filter of `sampling` and
    `rendering` from ONE field evaluation per sample, rays stopped where their transmittance falls below the filter's
    threshold -- the same arrays bit for bit (tests/test_gpu_parity.py).  native=False forces the staged composition
    (sampling with sigma_fn, then rendering) that training mode always uses."""
    if timestamps is None:
        raise NotImplementedError("DNGPradianceField needs timestamps (dnerf path of cednerf/utils.py:78-86)")
    rays, rays_shape, num_rays = _flatten_rays(rays)
    results, extra_info = [], []
    if native is None:
        native = (not radiance_field.training) and hasattr(radiance_field, "_descriptor") and rays.origins.is_cuda
    if native:
        assert not radiance_field.training, "the native render_image pass is the eval path"
    # The reference renders 8192 rays per pass in eval mode to bound memory (cednerf/utils.py:59,108-112).
    # Rays are independent, so here a pass covers up to `_EVAL_PASS_RAYS` rays (HBM is not the
    # constraint on MI355X) and its outputs are then cut into the reference's `test_chunk_size`
    # pieces, so the returned `extras` list and every value in it are those of the chunked loop.
    training = bool(radiance_field.training)
    if training:
        pass_rays = torch.iinfo(torch.int32).max
    else:
        pass_rays = max(test_chunk_size, (_EVAL_PASS_RAYS // test_chunk_size) * test_chunk_size)
    for i in range(0, num_rays, pass_rays):
        chunk_rays = namedtuple_map(lambda r: r[i:i + pass_rays].contiguous().float(), rays)
        ts = timestamps[i:i + pass_rays] if training else timestamps
        n_pass = chunk_rays.origins.shape[0]
        two_pass = native and float(alpha_thre) > 0.0
        if native and not two_pass:
            bk = None if render_bkgd is None else render_bkgd.to(chunk_rays.origins.device, torch.float32).reshape(-1).contiguous()
            chunked = n_pass > test_chunk_size
            tq = ts.reshape(-1).float().contiguous()

            def run(t_all0, t_all1, packed_all):
                return ops.render_image_eval_native(radiance_field._descriptor(), chunk_rays.origins, chunk_rays.viewdirs,
                                                    packed_all, t_all0, t_all1, 1e-4, 0.0, tq, False, bk,
                                                    chunk_rays=test_chunk_size if chunked else 0)

            # The march of every ray to the far plane needs the sample total before it can store (count pass, scan, fill
            # pass).  Consecutive frames of a video march about the same number of samples, so once a pass of this size
            # has been rendered the march runs in ONE pass into arrays 25 % larger than the last total; should a frame
            # not fit (the device-side total says so after the render) it is redone with the exact two-pass march.
            hints = estimator.__dict__.setdefault("_march_totals", {})
            hint = hints.get(n_pass)
            out = None
            if hint is not None and estimator.one_pass_march:
                cap = int(hint * 1.25) + 65536
                t_all0, t_all1, packed_all, total_dev = estimator.march_onepass(
                    chunk_rays.origins, chunk_rays.viewdirs, near_plane, far_plane, render_step_size, cone_angle, cap)
                out = run(t_all0, t_all1, packed_all)
                total = int(total_dev.item())
                hints[n_pass] = total
                if total > cap:
                    out = None
            if out is None:
                t_all0, t_all1, _, packed_all = estimator.march(
                    chunk_rays.origins, chunk_rays.viewdirs, near_plane=near_plane, far_plane=far_plane,
                    render_step_size=render_step_size, stratified=False, cone_angle=cone_angle, want_ray_indices=False)
                hints[n_pass] = int(t_all0.shape[0])
                out = run(t_all0, t_all1, packed_all)
            rgb, opacity, depth, ex, offsets, _ = out
            extras = {k: ex[k] for k in ("weights", "alphas", "trans", "sigmas", "rgbs", "ray_indices", "t_starts", "t_ends")}
            # a pass's pixels stay whole (cutting them into chunks only to concatenate them again would be ~250 slice
            # calls); the per-chunk `extras` are views cut by one split() per array
            results.append([rgb, opacity, depth, int(extras["t_starts"].shape[0])])
            if not chunked:
                extra_info.append(extras)
                continue
            starts = list(range(0, n_pass, test_chunk_size)) + [n_pass]
            bounds = offsets[torch.tensor(starts, device=offsets.device)].tolist()
            sizes = [bounds[c + 1] - bounds[c] for c in range(len(starts) - 1)]
            cut = {k: v.split(sizes) for k, v in extras.items()}
            extra_info.extend([{k: cut[k][c] for k in cut} for c in range(len(sizes))])
            continue

        def sigma_fn(t_starts, t_ends, ray_indices):
            return radiance_field.query_rays(chunk_rays.origins, chunk_rays.viewdirs, ray_indices, t_starts, t_ends,
                                             ts, want_rgb=False)[1]

        def rgb_sigma_fn(t_starts, t_ends, ray_indices):
            return radiance_field.query_rays(chunk_rays.origins, chunk_rays.viewdirs, ray_indices, t_starts, t_ends,
                                             ts, want_rgb=True)

        # With an alpha threshold most marched samples may be dropped without ending their ray (C3, C4): the native pass
        # is then the filter alone on the density-only kernel (front to back, rays stopped at the transmittance test)
        # and the whole field only on the survivors -- `sampling` and `rendering` as the reference stages them.
        ray_indices, t_starts, t_ends = estimator.sampling(
            chunk_rays.origins, chunk_rays.viewdirs, sigma_fn=sigma_fn, near_plane=near_plane, far_plane=far_plane,
            render_step_size=render_step_size, stratified=training, cone_angle=cone_angle, alpha_thre=alpha_thre,
            sigma_field=(radiance_field, ts, training) if two_pass else None)
        rgb, opacity, depth, extras = rendering(t_starts, t_ends, ray_indices, n_rays=n_pass,
                                                rgb_sigma_fn=rgb_sigma_fn, render_bkgd=render_bkgd)
        extras["ray_indices"] = ray_indices
        extras["t_starts"] = t_starts
        extras["t_ends"] = t_ends
        if training or n_pass <= test_chunk_size:
            results.append([rgb, opacity, depth, len(t_starts)])
            extra_info.append(extras)
            continue
        # cut the pass into the reference's chunks: samples are sorted by ray, so each chunk owns a
        # contiguous sample range (one small device->host copy of the range boundaries)
        starts = list(range(0, n_pass, test_chunk_size)) + [n_pass]
        bounds = torch.searchsorted(ray_indices, torch.tensor(starts, device=rgb.device)).tolist()
        sizes = [bounds[c + 1] - bounds[c] for c in range(len(starts) - 1)]
        extras["ray_indices"] = ray_indices % test_chunk_size         # relative to the ray's chunk
        cut = {k: v.split(sizes) for k, v in extras.items()}
        results.append([rgb, opacity, depth, len(t_starts)])          # the pixels of a pass stay whole
        extra_info.extend([{k: cut[k][c] for k in cut} for c in range(len(sizes))])
    colors, opacities, depths, n_rendering_samples = [
        torch.cat(r, dim=0) if isinstance(r[0], torch.Tensor) else r for r in zip(*results)
    ]
    return (colors.view((*rays_shape[:-1], -1)), opacities.view((*rays_shape[:-1], -1)),
            depths.view((*rays_shape[:-1], -1)), sum(n_rendering_samples), extra_info)


@torch.no_grad()
def render_image_test(
    max_samples: int,
    radiance_field: torch.nn.Module,
    estimator: OccGridEstimator,
    rays: Rays,
    near_plane: float = 0.0,
    far_plane: float = 1e10,
    render_step_size: float = 1e-3,
    render_bkgd: Optional[torch.Tensor] = None,
    cone_angle: float = 0.0,
    alpha_thre: float = 0.0,
    early_stop_eps: float = 1e-4,
    timestamps: Optional[torch.Tensor] = None,
    tracer=None,
    field_stream=None,
    field_max_workgroups: int = 0,
):
    """Iterative eval renderer with per-iteration early termination (cednerf/utils.py:153-318).
    Returns (rgb, opacity, depth, total_samples).  `alpha_thre` is accepted and unused, as in the
    reference.  The whole loop runs inside the native library (ced_render_image_test): four
    launches per iteration, the N_samples schedule computed on the device; `render_image_test_staged` is the
    same algorithm driven from Python through the nerfacc-shaped ops."""
    if timestamps is None:
        raise NotImplementedError("DNGPradianceField needs timestamps (dnerf path of cednerf/utils.py:186-194)")
    rays, rays_shape, N_rays = _flatten_rays(rays)
    rays_o = rays.origins.contiguous().float()
    rays_d = rays.viewdirs.contiguous().float()
    bk = None if render_bkgd is None else render_bkgd.to(rays_o.device, torch.float32).reshape(-1).contiguous()
    ts = timestamps.reshape(-1).float().contiguous()
    rgb, opacity, depth, total = ops.render_image_test_native(
        radiance_field._descriptor(), rays_o, rays_d, estimator.binaries, estimator.aabbs.contiguous(), near_plane,
        far_plane, render_step_size, cone_angle, early_stop_eps, max_samples, ts, bool(radiance_field.training), bk,
        tracer=tracer, field_stream=field_stream, accel=estimator.occupancy_accel(), max_workgroups=field_max_workgroups)
    return (rgb.view((*rays_shape[:-1], -1)), opacity.view((*rays_shape[:-1], -1)),
            depth.view((*rays_shape[:-1], -1)), total)


@torch.no_grad()
def render_frames_test(
    max_samples: int,
    radiance_field: torch.nn.Module,
    estimator: OccGridEstimator,
    rays: Rays,
    near_plane: float = 0.0,
    far_plane: float = 1e10,
    render_step_size: float = 1e-3,
    render_bkgd: Optional[torch.Tensor] = None,
    cone_angle: float = 0.0,
    alpha_thre: float = 0.0,
    early_stop_eps: float = 1e-4,
    timestamps: Optional[torch.Tensor] = None,
    tracer=None,
    field_stream=None,
    field_max_workgroups: int = 0,
    exchange=None,
):
    """`render_image_test` for a stack of frames in one native call (ced_render_frames_test): rays.origins / viewdirs
    are [F, ..., 3] (F <= 64 frames of equal size), timestamps holds F times.  The frames share the launches of an
    iteration but each keeps its own reference loop, so frame f of the result equals
    `render_image_test(rays[f], timestamps[f])` bit for bit.  Returns (rgb [F,...,3], opacity [F,...,1],
    depth [F,...,1], [total_samples of every frame]).
    exchange (ops.ScheduleExchange): rays[f] is this rank's share of frame f, whose other rays other ranks render; the
    loop of every frame is then the WHOLE image's (N_rays // N_alive over all ranks, cednerf/utils.py:231-235)."""
    if timestamps is None:
        raise NotImplementedError("DNGPradianceField needs timestamps (dnerf path of cednerf/utils.py:186-194)")
    if radiance_field.training:
        raise NotImplementedError("render_frames_test renders eval frames (one time per frame)")
    shape = tuple(rays.origins.shape)
    assert len(shape) >= 3 and shape[-1] == 3 and tuple(rays.viewdirs.shape) == shape, "rays must be [F, ..., 3]"
    n_frames = shape[0]
    rays_o = rays.origins.reshape(-1, 3).contiguous().float()
    rays_d = rays.viewdirs.reshape(-1, 3).contiguous().float()
    bk = None if render_bkgd is None else render_bkgd.to(rays_o.device, torch.float32).reshape(-1).contiguous()
    ts = timestamps.reshape(-1).float().contiguous()
    assert ts.numel() == n_frames, "one timestamp per frame"
    rgb, opacity, depth, totals = ops.render_frames_test_native(
        radiance_field._descriptor(), n_frames, rays_o, rays_d, estimator.binaries, estimator.aabbs.contiguous(),
        near_plane, far_plane, render_step_size, cone_angle, early_stop_eps, max_samples, ts, bk,
        tracer=tracer, field_stream=field_stream, accel=estimator.occupancy_accel(), max_workgroups=field_max_workgroups,
        exchange=exchange)
    return rgb.view((*shape[:-1], 3)), opacity.view((*shape[:-1], 1)), depth.view((*shape[:-1], 1)), totals


@torch.no_grad()
def render_image_test_staged(
    max_samples: int,
    radiance_field: torch.nn.Module,
    estimator: OccGridEstimator,
    rays: Rays,
    near_plane: float = 0.0,
    far_plane: float = 1e10,
    render_step_size: float = 1e-3,
    render_bkgd: Optional[torch.Tensor] = None,
    cone_angle: float = 0.0,
    alpha_thre: float = 0.0,
    early_stop_eps: float = 1e-4,
    timestamps: Optional[torch.Tensor] = None,
):
    """render_image_test (cednerf/utils.py:153-318) staged from Python through the nerfacc-shaped
    ops (count -> scan -> fill marching, field, composite_step): the reference's own structure,
    kept for per-kernel profiling and as a cross-check of the native loop."""
    if timestamps is None:
        raise NotImplementedError("DNGPradianceField needs timestamps (dnerf path of cednerf/utils.py:186-194)")
    rays, rays_shape, N_rays = _flatten_rays(rays)
    rays_o = rays.origins.contiguous().float()
    rays_d = rays.viewdirs.contiguous().float()
    device = rays_o.device
    opacity = torch.zeros(N_rays, 1, device=device)
    depth = torch.zeros(N_rays, 1, device=device)
    rgb = torch.zeros(N_rays, 3, device=device)
    ray_mask = torch.ones(N_rays, device=device).bool()
    min_samples = 1 if cone_angle == 0 else 4
    iter_samples = total_samples = 0
    near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
    far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
    aabbs = estimator.aabbs.contiguous()
    t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
    t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    opc_thres = 1 - early_stop_eps
    op_flat, dp_flat = opacity.view(-1), depth.view(-1)
    binaries = estimator.binaries
    # Per-iteration bookkeeping stays on the device: the composite kernel updates the ray mask and
    # counts the surviving rays / composited samples; the host reads both with one 16-byte copy
    # (the single sync per iteration that the reference's `ray_mask.sum().item()` also pays).
    stats = torch.zeros((max(int(max_samples), 1) + 1, 2), device=device, dtype=torch.int64)
    host_stats = torch.zeros((2,), dtype=torch.int64).pin_memory()
    stream = torch.cuda.current_stream()
    N_alive = N_rays
    it = 0
    while iter_samples < max_samples:
        if N_alive == 0:
            break
        N_samples = max(min(N_rays // N_alive, 64), min_samples)
        iter_samples += N_samples
        bound = N_alive * N_samples                       # host-known upper bound of this iteration's samples
        counts = torch.empty((N_rays,), device=device, dtype=torch.int64)
        march = (rays_o, rays_d, binaries, aabbs, near_planes, far_planes, render_step_size, cone_angle, N_samples,
                 ray_mask, t_sorted, t_indices, hits)
        ops.traverse_grids_raw(*march, 0, counts=counts)
        incl = torch.cumsum(counts, 0)
        t_starts = torch.empty((bound,), device=device, dtype=torch.float32)
        t_ends = torch.empty((bound,), device=device, dtype=torch.float32)
        ray_indices = torch.empty((bound,), device=device, dtype=torch.int64)
        packed_info = torch.empty((N_rays, 2), device=device, dtype=torch.int64)
        # fill; the termination planes overwrite the near planes in place (cednerf/utils.py:301)
        ops.traverse_grids_raw(*march, 3, base=incl, counts=counts, t_starts=t_starts, t_ends=t_ends,
                               ray_indices=ray_indices, termination_planes=near_planes, packed_info_out=packed_info)
        rgbs, sigmas = radiance_field.query_rays(rays_o, rays_d, ray_indices, t_starts, t_ends, timestamps,
                                                 want_rgb=True, n_dev=incl[-1:])
        ops.composite_step_(packed_info, t_starts, t_ends, sigmas, rgbs, rgb, op_flat, dp_flat, opc_thres, N_samples,
                            ray_mask, stats[it])
        host_stats.copy_(stats[it], non_blocking=True)
        stream.synchronize()
        N_alive = int(host_stats[0])
        total_samples += int(host_stats[1])
        it += 1

    bk = None if render_bkgd is None else render_bkgd.to(device, torch.float32).reshape(-1).contiguous()
    ops.finalize_pixels_(bk, rgb, op_flat, dp_flat)
    return (rgb.view((*rays_shape[:-1], -1)), opacity.view((*rays_shape[:-1], -1)),
            depth.view((*rays_shape[:-1], -1)), total_samples)


# nerfacc's example name for the same role (BASELINE.json north_star)
render_image_with_occgrid = render_image
