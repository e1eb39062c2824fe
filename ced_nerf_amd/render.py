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
        w = weights.float().reshape(weights.shape[0], -1)
        if w.shape[1] not in (1, values.shape[1]):
            w = w.expand(-1, values.shape[1])
    return _ReduceFn.apply(ray_indices.long().contiguous(), values.float(), w, int(n_rays), reduce == "mean")


class _ReduceFn(torch.autograd.Function):
    """ced_reduce_along_rays with the gradient of torch's scatter_reduce_ (sum, or mean with include_self): the
    forward is the HIP kernel, the backward two element-wise gathers."""

    @staticmethod
    def forward(ctx, ray_indices, values, weights, n_rays, mean):
        v = values.detach().contiguous()
        w = None if weights is None else weights.detach().contiguous()
        out = ops.reduce_along_rays(ray_indices, v, n_rays, w, mean=mean)
        ctx.save_for_backward(ray_indices, v, w if w is not None else torch.empty(0, device=v.device))
        ctx.has_w, ctx.mean, ctx.n_rays = w is not None, mean, n_rays
        return out

    @staticmethod
    def backward(ctx, d_out):
        ri, v, w = ctx.saved_tensors
        g = d_out.float()
        if ctx.mean:
            counts = torch.bincount(ri, minlength=ctx.n_rays).to(g.dtype) + 1.0
            g = g / counts[:, None]
        g = g[ri]                                                   # [S, C]
        d_v = (g * w) if ctx.has_w else g
        d_w = None
        if ctx.has_w and ctx.needs_input_grad[2]:
            d_w = g * v
            if w.shape[1] == 1:
                d_w = d_w.sum(dim=1, keepdim=True)
        return None, (d_v if ctx.needs_input_grad[1] else None), d_w, None, None


class _WeightsFn(torch.autograd.Function):
    """(weights, trans, alphas) of render_weight_from_density as differentiable functions of sigmas: forward on the
    HIP kernel (ced_render_weights), backward in closed form (element-wise + one per-ray suffix sum):
      w_i = T_i a_i, T_i = exp(-sum_{j<i} s_j d_j), a_i = 1 - exp(-s_i d_i)
      d s_i = d_i [ (dw_i T_i + da_i)(1 - a_i) - sum_{k>i} (dw_k w_k + dT_k T_k) ]"""

    @staticmethod
    def forward(ctx, sigmas, t_starts, t_ends, packed):
        sig = sigmas.detach().float().contiguous()
        w, tr, al = render_weight_from_density(t_starts, t_ends, sig, packed_info=packed)
        ctx.save_for_backward(w, tr, al, t_starts, t_ends, packed)
        return w, tr, al

    @staticmethod
    def backward(ctx, d_w, d_t, d_a):
        w, tr, al, t0, t1, packed = ctx.saved_tensors
        delta = t1 - t0
        d_w = torch.zeros_like(w) if d_w is None else d_w.float()
        d_t = torch.zeros_like(w) if d_t is None else d_t.float()
        d_a = torch.zeros_like(w) if d_a is None else d_a.float()
        u = d_w * w + d_t * tr
        cs = torch.cumsum(u.double(), 0)                            # inclusive, over all rays
        start, cnt = packed[:, 0], packed[:, 1]
        has = cnt > 0
        ray_end = torch.zeros(packed.shape[0], device=w.device, dtype=torch.float64)
        ray_end[has] = cs[(start + cnt - 1)[has]]
        ray_of = torch.repeat_interleave(torch.arange(packed.shape[0], device=w.device), cnt)
        suffix = (ray_end[ray_of] - cs).float()                     # sum over k > i inside the ray
        d_sig = delta * ((d_w * tr + d_a) * (1.0 - al) - suffix)
        return d_sig, None, None, None


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
    extras = {"sigmas": sigmas, "rgbs": rgbs}
    # training extras of cednerf/render.py:101-124: per-ray reductions of the prediction heads' losses
    internal = sigma_results.get("interal_output") if isinstance(sigma_results, dict) else None
    if internal is not None and ("latent_losses" in internal or "weight_losses" in internal):
        weights, trans, _ = _WeightsFn.apply(sigmas, t_starts.contiguous(), t_ends.contiguous(), packed)
        extras["weights"], extras["trans"] = weights, trans
        selector = internal["selector"]
        if "latent_losses" in internal:
            extras["latent_losses"] = reduce_along_rays(ray_indices, values=internal["latent_losses"], n_rays=n_rays,
                                                        reduce="sum", weights=weights[:, None].detach())
        if "weight_losses" in internal:
            target_weights = trans[:, None]
            p_weight = internal["weight_losses"].float()
            weight_loss = torch.nn.functional.huber_loss(p_weight, target_weights, reduction="none")
            extras["weight_losses"] = reduce_along_rays(ray_indices, values=weight_loss * selector[:, None].to(weight_loss.dtype),
                                                        n_rays=n_rays, weights=weights[:, None])
    return colors, opacities, depths, extras
