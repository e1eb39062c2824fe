"""Volume rendering of ray-packed samples: host mirror of cednerf/render.py over the HIP kernels."""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
from torch import Tensor

from . import ops
from .nerfacc_api import (_packed_info_from, accumulate_along_rays, render_transmittance_from_density,
                          render_weight_from_density)


def reduce_along_rays(ray_indices: Tensor, values: Tensor, n_rays: Optional[int] = None,
                      weights: Optional[Tensor] = None, reduce: str = "mean") -> Tensor:
    """cednerf/render.py:8-39: per-ray scatter_reduce_ of weights * values (the training extras of `rendering`,
    render.py:101-124), on the HIP kernel ced_reduce_along_rays.  Same asserts and error behaviour as the reference."""
    assert ray_indices.dim() == 1 and values.dim() == 2
    if not values.is_cuda:
        raise NotImplementedError("Only support cuda inputs.")
    if weights is not None:
        assert values.dim() == 2 and values.shape[0] == weights.shape[0], \
            "Invalid shapes: {} vs {}".format(values.shape, weights.shape)
    if ray_indices.numel() == 0:
        assert n_rays is not None
        return torch.zeros((n_rays, values.shape[-1]), device=values.device)
    if n_rays is None:
        n_rays = int(ray_indices.max()) + 1
    if reduce not in ("sum", "mean"):
        raise NotImplementedError(f"reduce={reduce!r}: the reference uses 'mean' (default) and 'sum'")
    w = None
    if weights is not None:
        w = weights.float().reshape(weights.shape[0], -1).contiguous()
        if w.shape[1] not in (1, values.shape[1]):
            w = w.expand(-1, values.shape[1]).contiguous()
    return ops.reduce_along_rays(ray_indices.long().contiguous(), values.float().contiguous(), int(n_rays), w,
                                 mean=(reduce == "mean"))


def render_weight_from_density_prefix(t_starts: Tensor, t_ends: Tensor, sigmas: Tensor, prefix_trans: Tensor,
                                      packed_info: Optional[Tensor] = None, ray_indices: Optional[Tensor] = None,
                                      n_rays: Optional[int] = None) -> Tuple[Tensor, Tensor, Tensor]:
    """cednerf/render.py:42-56: (weights, trans, alphas) with a caller-supplied prefix transmittance."""
    trans, alphas = render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info, ray_indices, n_rays,
                                                      prefix_trans)
    return trans * alphas, trans, alphas


@torch.no_grad()
def rendering(t_starts: Tensor, t_ends: Tensor, ray_indices: Tensor, n_rays: int, rgb_sigma_fn: Callable,
              render_bkgd: Optional[Tensor] = None):
    """cednerf/render.py:58-176 (eval path): field -> weights -> rgb/opacity/depth per ray.
    Returns (colors [n,3], opacities [n,1], depths [n,1], extras)."""
    rgbs, sigma_results = rgb_sigma_fn(t_starts, t_ends, ray_indices)
    sigmas = sigma_results["density"].squeeze(-1) if isinstance(sigma_results, dict) else sigma_results
    assert rgbs.shape[-1] == 3, "rgbs must have 3 channels, got {}".format(rgbs.shape)
    assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N, 1)! Got {}".format(sigmas.shape)
    packed = _packed_info_from(ray_indices, n_rays)
    weights, trans, alphas = render_weight_from_density(t_starts, t_ends, sigmas, packed_info=packed)
    extras = {"weights": weights, "alphas": alphas, "trans": trans, "sigmas": sigmas, "rgbs": rgbs}
    colors = accumulate_along_rays(weights, values=rgbs, packed_info=packed)
    opacities = accumulate_along_rays(weights, values=None, packed_info=packed)
    depths = accumulate_along_rays(weights, values=(t_starts + t_ends)[..., None] / 2.0, packed_info=packed)
    bk = None
    if render_bkgd is not None:
        bk = render_bkgd.to(colors.device, torch.float32).reshape(-1).contiguous()
    op_flat = opacities.reshape(-1)
    dp_flat = depths.reshape(-1)
    ops.finalize_pixels_(bk, colors, op_flat, dp_flat)
    return colors, opacities, depths, extras


class _CompositeFunction(torch.autograd.Function):
    """(sigmas [S], rgbs [S,3]) -> un-normalised (colors [n,3], opacities [n,1], depths [n,1]) with the HIP forward
    (render_weight_from_density + accumulate_along_rays) and the HIP backward ced_composite_backward."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, t_starts, t_ends, packed):
        sig = sigmas.detach().float().contiguous(); rgb = rgbs.detach().float().contiguous()
        weights, _, _ = render_weight_from_density(t_starts, t_ends, sig, packed_info=packed)
        colors = accumulate_along_rays(weights, values=rgb, packed_info=packed)
        opacities = accumulate_along_rays(weights, values=None, packed_info=packed)
        depths = accumulate_along_rays(weights, values=(t_starts + t_ends)[..., None] / 2.0, packed_info=packed)
        ctx.save_for_backward(sig, rgb, t_starts, t_ends, packed)
        return colors, opacities, depths

    @staticmethod
    def backward(ctx, d_colors, d_opacities, d_depths):
        sig, rgb, t_starts, t_ends, packed = ctx.saved_tensors
        d_sig, d_rgb = ops.composite_backward(packed, t_starts, t_ends, sig, rgb, d_colors.float().contiguous(),
                                              d_opacities.float().reshape(-1).contiguous(),
                                              d_depths.float().reshape(-1).contiguous())
        return d_sig, d_rgb, None, None, None


def rendering_train(t_starts: Tensor, t_ends: Tensor, ray_indices: Tensor, n_rays: int, rgb_sigma_fn: Callable,
                    render_bkgd: Optional[Tensor] = None):
    """cednerf/render.py:58-176 with gradients (the training call, train_real.py:339-350): `rgb_sigma_fn` returns
    differentiable (rgbs, sigmas); compositing runs on the HIP kernels in both directions.  First pieces of the
    training path: returns (colors, opacities, depths, extras) like `rendering`."""
    rgbs, sigma_results = rgb_sigma_fn(t_starts, t_ends, ray_indices)
    sigmas = sigma_results["density"].squeeze(-1) if isinstance(sigma_results, dict) else sigma_results
    assert rgbs.shape[-1] == 3, "rgbs must have 3 channels, got {}".format(rgbs.shape)
    assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N, 1)! Got {}".format(sigmas.shape)
    packed = _packed_info_from(ray_indices, n_rays)
    colors, opacities, depths = _CompositeFunction.apply(sigmas, rgbs, t_starts.contiguous(), t_ends.contiguous(), packed)
    depths = depths / opacities.clamp_min(torch.finfo(torch.float32).eps)
    if render_bkgd is not None:
        colors = colors + render_bkgd.to(colors.device, torch.float32) * (1.0 - opacities)
    return colors, opacities, depths, {"sigmas": sigmas, "rgbs": rgbs}
