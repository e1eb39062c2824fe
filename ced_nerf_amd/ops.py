"""Torch-tensor front-end of the C ABI (``include/cednerf_hip.h``).

PyTorch is used for device memory and streams only; all arithmetic happens in the HIP kernels.
Every wrapper validates device / dtype / contiguity (mirroring the reference's asserts,
cednerf/render.py:16-22,74-79) and enqueues on the current torch stream.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _lib, profiling
from .hashgrid import level_tables


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: Optional[torch.Tensor], dtype: torch.dtype, name: str, allow_none: bool = False):
    if t is None:
        if allow_none:
            return None
        raise ValueError(f"{name} is required")
    if not t.is_cuda:
        raise NotImplementedError(f"Only support cuda inputs ({name} is on {t.device}).")
    _check_current_device(t.device.index, name)
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    return t


def _check_current_device(index, name: str) -> None:
    """The library launches on the CURRENT device and stream (`_stream()`), with raw pointers: a tensor of another
    device would be dereferenced on the wrong GPU (a memory fault, not an exception).  Refuse instead."""
    cur = torch.cuda.current_device()
    if index is not None and index != cur:
        raise RuntimeError(f"{name} lives on cuda:{index} but the current device is cuda:{cur}: wrap the call in "
                           f"`with torch.cuda.device({index}):` (the kernels launch on the current device's stream)")


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _as_u8(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.view(torch.uint8) if t.dtype == torch.bool else t


# ----------------------------------------------------------------------------------------------
# field parameters on the device
# ----------------------------------------------------------------------------------------------
def make_hash_desc(table: torch.Tensor, base_res: int, max_res: int, n_levels: int, log2_hashmap_size: int,
                   temporal: bool = False) -> Tuple[_lib.HashDesc, Dict]:
    tabs = level_tables(base_res, max_res, n_levels, log2_hashmap_size)
    width = 8 if temporal else 2
    if table.dtype not in (torch.float32, torch.float16):
        raise TypeError("hash table must be float32 or float16")
    if tuple(table.shape) != (tabs["total"], width) or not table.is_contiguous() or not table.is_cuda:
        raise ValueError(f"hash table must be a contiguous cuda tensor of shape {(tabs['total'], width)}, "
                         f"got {tuple(table.shape)} on {table.device}")
    d = _lib.HashDesc()
    d.n_levels = n_levels
    d.table_dtype = 0 if table.dtype == torch.float32 else 1
    d.temporal = int(bool(temporal))
    for l in range(n_levels):
        d.scale[l] = float(tabs["scale"][l])
        d.res[l] = int(tabs["res"][l])
        d.offset[l] = int(tabs["offset"][l])
        d.size[l] = int(tabs["size"][l])
        d.hashed[l] = int(tabs["hashed"][l])
    d.table = table.data_ptr()
    d.total_entries = int(tabs["total"])
    return d, tabs


def pack_field_weights(use_div_offsets: bool, time_mode: int, xyz_wrap, mlp_base, mlp_head,
                       mlp_precision: int = _lib.MLP_F32) -> np.ndarray:
    """Host: natural W[out][in] float32 arrays -> MFMA-fragment-order blob (ced_pack_field_weights for the
    fp32 kernel, ced_pack_field_weights_half for the f16x2 / f16 kernels; the latter returns uint32 words)."""
    L = _lib.lib()
    mats = [np.ascontiguousarray(np.asarray(w, np.float32)) for w in list(xyz_wrap) + list(mlp_base) + list(mlp_head)]
    base_in = 41 if time_mode else 32
    n_mo = 6 if use_div_offsets else 3
    want = [(64, 32), (64, 64), (64, 64), (n_mo, 64), (64, base_in), (16, 64), (64, 19), (64, 64), (3, 64)]
    got = [m.shape for m in mats]
    if got != want:
        raise ValueError(f"weight shapes {got} do not match the DNGPradianceField layout {want}")
    if mlp_precision == _lib.MLP_F32_HEAD16X2:          # fp32 sigma chain + fp16 fragments of the colour head, fp32-blob sized
        n = int(L.ced_packed_weight_floats(int(use_div_offsets), int(time_mode)))
        out = np.zeros((n,), np.float32)
        rc = L.ced_pack_field_weights_mixed(int(use_div_offsets), int(time_mode),
                                            *[m.ctypes.data_as(C.c_void_p) for m in mats], out.ctypes.data_as(C.c_void_p))
        _lib.check(rc, "pack_field_weights_mixed")
        return out
    if mlp_precision != _lib.MLP_F32:
        n = int(L.ced_packed_weight_words(int(use_div_offsets), int(time_mode), int(mlp_precision)))
        if n <= 0:
            raise ValueError(f"mlp_precision={mlp_precision}")
        out = np.zeros((n,), np.uint32)
        rc = L.ced_pack_field_weights_half(int(use_div_offsets), int(time_mode), int(mlp_precision),
                                           *[m.ctypes.data_as(C.c_void_p) for m in mats], out.ctypes.data_as(C.c_void_p))
        _lib.check(rc, "pack_field_weights_half")
        return out
    n = int(L.ced_packed_weight_floats(int(use_div_offsets), int(time_mode)))
    out = np.zeros((n,), np.float32)
    rc = L.ced_pack_field_weights(int(use_div_offsets), int(time_mode),
                                  *[m.ctypes.data_as(C.c_void_p) for m in mats], out.ctypes.data_as(C.c_void_p))
    _lib.check(rc, "pack_field_weights")
    return out


# ----------------------------------------------------------------------------------------------
# marching
# ----------------------------------------------------------------------------------------------
def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane=-float("inf"), far_plane=float("inf"),
                       miss_value=float("inf")):
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d"); _chk(aabbs, torch.float32, "aabbs")
    assert rays_o.ndim == 2 and rays_o.shape[-1] == 3 and rays_o.shape == rays_d.shape
    assert aabbs.ndim == 2 and aabbs.shape[-1] == 6
    n, m = rays_o.shape[0], aabbs.shape[0]
    t_mins = torch.empty((n, m), device=rays_o.device, dtype=torch.float32)
    t_maxs = torch.empty_like(t_mins)
    hits = torch.empty((n, m), device=rays_o.device, dtype=torch.bool)
    rc = _lib.lib().ced_ray_aabb_intersect(n, _p(rays_o), _p(rays_d), m, _p(aabbs), near_plane, far_plane,
                                           miss_value, _p(t_mins), _p(t_maxs), _p(hits), _stream())
    _lib.check(rc, "ray_aabb_intersect")
    return t_mins, t_maxs, hits


def sort_intersections(t_mins, t_maxs):
    """ced_sort_intersections: (t_sorted [n, 2m] f32, t_indices [n, 2m] int64) = torch.sort(cat([t_mins, t_maxs], -1),
    stable=True) per ray, in one launch."""
    _chk(t_mins, torch.float32, "t_mins"); _chk(t_maxs, torch.float32, "t_maxs")
    assert t_mins.ndim == 2 and t_mins.shape == t_maxs.shape
    n, m = t_mins.shape
    t_sorted = torch.empty((n, 2 * m), device=t_mins.device, dtype=torch.float32)
    t_indices = torch.empty((n, 2 * m), device=t_mins.device, dtype=torch.int64)
    rc = _lib.lib().ced_sort_intersections(n, m, _p(t_mins), _p(t_maxs), _p(t_sorted), _p(t_indices), _stream())
    _lib.check(rc, "sort_intersections")
    return t_sorted, t_indices


def traverse_grids_raw(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle, limit,
                       rays_mask, t_sorted, t_indices, hits, mode, base=None, counts=None, t_starts=None,
                       t_ends=None, ray_indices=None, termination_planes=None, packed_info_out=None):
    L = _lib.lib()
    n = rays_o.shape[0]
    m, res = binaries.shape[0], binaries.shape[1]
    with profiling.span("traverse", n):
        rc = L.ced_traverse_grids(n, _p(rays_o), _p(rays_d), _p(_as_u8(binaries)), m, res, _p(aabbs), _p(near_planes),
                                  _p(far_planes), float(step_size), float(cone_angle), int(limit),
                                  _p(_as_u8(rays_mask)), _p(t_sorted), _p(t_indices), _p(_as_u8(hits)), int(mode),
                                  _p(base), _p(counts), _p(t_starts), _p(t_ends), _p(ray_indices),
                                  _p(termination_planes), _p(packed_info_out), _stream())
    _lib.check(rc, "traverse_grids")


# ----------------------------------------------------------------------------------------------
# hash grid / field
# ----------------------------------------------------------------------------------------------
def hash_encode(desc: _lib.HashDesc, x: torch.Tensor, t: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, torch.float32, "x")
    assert x.ndim == 2 and x.shape[-1] == 3
    if t is not None:
        _chk(t, torch.float32, "t")
        assert t.numel() == x.shape[0]
    out = torch.empty((x.shape[0], 2 * desc.n_levels), device=x.device, dtype=torch.float32)
    rc = _lib.lib().ced_hash_encode(C.byref(desc), x.shape[0], _p(x), _p(t), _p(out), _stream())
    _lib.check(rc, "hash_encode")
    return out


def hash_encode_backward(desc: _lib.HashDesc, x: torch.Tensor, dy: torch.Tensor,
                         grad_table: Optional[torch.Tensor] = None, want_dx: bool = True, dx_scaled: bool = False,
                         want_table: bool = True):
    """ced_hash_encode_backward: (grad_table [E,2] fp32 -- accumulated into when given, else fresh --, dx [n,3] or None).
    want_table=False: the position gradient alone (grad_table returned as None)."""
    _chk(x, torch.float32, "x"); _chk(dy, torch.float32, "dy")
    n = x.shape[0]
    assert x.shape == (n, 3) and dy.numel() == n * 2 * desc.n_levels, f"{x.shape} v.s. {dy.shape}"
    assert want_table or want_dx
    if not want_table:
        grad_table = None
    elif grad_table is None:
        grad_table = torch.zeros((int(desc.total_entries), 2), device=x.device, dtype=torch.float32)
    else:
        _chk(grad_table, torch.float32, "grad_table")
        assert grad_table.shape == (int(desc.total_entries), 2)
    dx = torch.empty((n, 3), device=x.device, dtype=torch.float32) if want_dx else None
    rc = _lib.lib().ced_hash_encode_backward(C.byref(desc), n, _p(x), _p(dy), _p(grad_table), _p(dx), int(dx_scaled),
                                             _stream())
    _lib.check(rc, "hash_encode_backward")
    return grad_table, dx


def hash_encode_backward_temporal(desc: _lib.HashDesc, x: torch.Tensor, t: torch.Tensor, dy: torch.Tensor,
                                  grad_table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ced_hash_encode_backward_temporal (hash_encoder_inter.py:202-275): grad_table [E, 8] fp32, accumulated into when
    given.  The temporal encoder has a table gradient only (the reference's autograd function returns none for positions)."""
    _chk(x, torch.float32, "x"); _chk(dy, torch.float32, "dy"); _chk(t, torch.float32, "t")
    n = x.shape[0]
    assert desc.temporal, "not a temporal table"
    assert x.shape == (n, 3) and t.numel() == n and dy.numel() == n * 2 * desc.n_levels, f"{x.shape} {t.shape} {dy.shape}"
    if grad_table is None:
        grad_table = torch.zeros((int(desc.total_entries), 8), device=x.device, dtype=torch.float32)
    else:
        _chk(grad_table, torch.float32, "grad_table")
        assert grad_table.shape == (int(desc.total_entries), 8)
    rc = _lib.lib().ced_hash_encode_backward_temporal(C.byref(desc), n, _p(x), _p(t), _p(dy), _p(grad_table), _stream())
    _lib.check(rc, "hash_encode_backward_temporal")
    return grad_table


def field_forward(desc: _lib.FieldDesc, positions, t, directions=None, want_geo=False):
    _chk(positions, torch.float32, "positions"); _chk(t, torch.float32, "t")
    n = positions.shape[0]
    assert positions.shape == (n, 3) and t.numel() == n
    dev = positions.device
    rgb = None
    if directions is not None:
        _chk(directions, torch.float32, "directions")
        assert directions.shape == positions.shape, f"{positions.shape} v.s. {directions.shape}"
        rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
    sigma = torch.empty((n,), device=dev, dtype=torch.float32)
    geo = torch.empty((n, 15), device=dev, dtype=torch.float32) if want_geo else None
    rc = _lib.lib().ced_field_forward(C.byref(desc), n, _p(positions), _p(t), _p(directions), _p(rgb), _p(sigma),
                                      _p(geo), _stream())
    _lib.check(rc, "field_forward")
    return rgb, sigma, geo


def field_forward_rays(desc: _lib.FieldDesc, rays_o, rays_d, ray_indices, t_starts, t_ends, timestamps,
                       t_per_ray: bool, want_rgb: bool, n_dev: Optional[torch.Tensor] = None):
    """n_dev: optional device int64 scalar; the kernel evaluates min(len(ray_indices), n_dev) samples."""
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d")
    _chk(ray_indices, torch.int64, "ray_indices")
    _chk(t_starts, torch.float32, "t_starts"); _chk(t_ends, torch.float32, "t_ends")
    _chk(timestamps, torch.float32, "timestamps")
    n = ray_indices.shape[0]
    assert t_starts.shape == (n,) and t_ends.shape == (n,)
    if t_per_ray:
        assert timestamps.numel() == rays_o.shape[0], "per-ray timestamps must have one entry per ray"
    dev = rays_o.device
    rgb = torch.empty((n, 3), device=dev, dtype=torch.float32) if want_rgb else None
    sigma = torch.empty((n,), device=dev, dtype=torch.float32)
    with profiling.span("field", n):
        rc = _lib.lib().ced_field_forward_rays(C.byref(desc), n, _p(n_dev), _p(rays_o), _p(rays_d), _p(ray_indices),
                                               _p(t_starts), _p(t_ends), _p(timestamps), int(bool(t_per_ray)),
                                               int(bool(want_rgb)), _p(rgb), _p(sigma), _stream())
    _lib.check(rc, "field_forward_rays")
    return rgb, sigma


# ----------------------------------------------------------------------------------------------
# compositing
# ----------------------------------------------------------------------------------------------
def render_weights(packed_info, t_starts, t_ends, sigmas, prefix_trans=None, want=(True, True, True)):
    _chk(packed_info, torch.int64, "packed_info"); _chk(t_starts, torch.float32, "t_starts")
    _chk(t_ends, torch.float32, "t_ends"); _chk(sigmas, torch.float32, "sigmas")
    _chk(prefix_trans, torch.float32, "prefix_trans", allow_none=True)
    outs = [torch.empty_like(t_starts) if w else None for w in want]
    if t_starts.shape[0] == 0:
        return tuple(outs)
    rc = _lib.lib().ced_render_weights(packed_info.shape[0], _p(packed_info), _p(t_starts), _p(t_ends), _p(sigmas),
                                       _p(prefix_trans), _p(outs[0]), _p(outs[1]), _p(outs[2]), _stream())
    _lib.check(rc, "render_weights")
    return tuple(outs)


def accumulate_along_rays_(packed_info, weights, values, outputs):
    _chk(packed_info, torch.int64, "packed_info"); _chk(weights, torch.float32, "weights")
    _chk(values, torch.float32, "values", allow_none=True); _chk(outputs, torch.float32, "outputs")
    n_rays = packed_info.shape[0]
    C_ = outputs.shape[-1]
    assert outputs.shape == (n_rays, C_)
    if values is not None:
        assert values.shape == (weights.shape[0], C_), f"Invalid shapes: {values.shape} vs {weights.shape}"
    if weights.shape[0] == 0:
        return outputs
    rc = _lib.lib().ced_accumulate_along_rays(n_rays, _p(packed_info), _p(weights), _p(values), C_, _p(outputs),
                                              _stream())
    _lib.check(rc, "accumulate_along_rays")
    return outputs


def reduce_along_rays(ray_indices, values, n_rays: int, weights=None, mean: bool = False):
    """ced_reduce_along_rays: out [n_rays, C] = sum (or torch's include_self mean) of weights * values per ray."""
    _chk(ray_indices, torch.int64, "ray_indices"); _chk(values, torch.float32, "values")
    _chk(weights, torch.float32, "weights", allow_none=True)
    n, c = values.shape
    wc = 0 if weights is None else weights.shape[1]
    out = torch.empty((n_rays, c), device=values.device, dtype=torch.float32)
    counts = torch.empty((n_rays,), device=values.device, dtype=torch.int32) if mean else None
    rc = _lib.lib().ced_reduce_along_rays(n, _p(ray_indices), _p(values), c, _p(weights), wc, n_rays, int(bool(mean)), _p(out),
                                          _p(counts), _stream())
    _lib.check(rc, "reduce_along_rays")
    return out


def visibility_mask(packed_info, t_starts, t_ends, sigmas, early_stop_eps, alpha_thre):
    _chk(packed_info, torch.int64, "packed_info")
    mask = torch.empty(t_starts.shape, device=t_starts.device, dtype=torch.bool)
    if t_starts.shape[0] == 0:
        return mask
    rc = _lib.lib().ced_visibility_mask(packed_info.shape[0], _p(packed_info), _p(t_starts), _p(t_ends), _p(sigmas),
                                        float(early_stop_eps), float(alpha_thre), _p(mask), _stream())
    _lib.check(rc, "visibility_mask")
    return mask


def composite_backward(packed_info, t_starts, t_ends, sigmas, rgbs, d_color, d_opacity=None, d_depth=None):
    """ced_composite_backward: (d_sigmas [S], d_rgbs [S,3]) of the un-normalised colors / opacities / depths."""
    _chk(packed_info, torch.int64, "packed_info"); _chk(sigmas, torch.float32, "sigmas"); _chk(rgbs, torch.float32, "rgbs")
    _chk(t_starts, torch.float32, "t_starts"); _chk(t_ends, torch.float32, "t_ends"); _chk(d_color, torch.float32, "d_color")
    n_rays, S = packed_info.shape[0], sigmas.shape[0]
    assert rgbs.shape == (S, 3) and d_color.shape == (n_rays, 3)
    d_sig = torch.zeros((S,), device=sigmas.device, dtype=torch.float32)
    d_rgb = torch.zeros((S, 3), device=sigmas.device, dtype=torch.float32)
    if S == 0 or n_rays == 0:
        return d_sig, d_rgb
    for t, nm in ((d_opacity, "d_opacity"), (d_depth, "d_depth")):
        if t is not None:
            _chk(t, torch.float32, nm)
            assert t.numel() == n_rays
    rc = _lib.lib().ced_composite_backward(n_rays, _p(packed_info), _p(t_starts), _p(t_ends), _p(sigmas), _p(rgbs),
                                           _p(d_color), _p(d_opacity), _p(d_depth), _p(d_sig), _p(d_rgb), _stream())
    _lib.check(rc, "composite_backward")
    return d_sig, d_rgb


def frame_to_rgb8(rgb, flip_w: bool = True):
    """ced_frame_to_rgb8: [H,W,3] float colours -> [H,W,3] uint8 (x 255, truncated), flipped along the width as the
    reference's video frames are (train_real.py:556)."""
    _chk(rgb, torch.float32, "rgb")
    assert rgb.dim() == 3 and rgb.shape[2] == 3, "frame_to_rgb8: rgb [H,W,3]"
    out = torch.empty(rgb.shape, device=rgb.device, dtype=torch.uint8)
    _lib.check(_lib.lib().ced_frame_to_rgb8(rgb.shape[0], rgb.shape[1], _p(rgb), int(bool(flip_w)), _p(out), _stream()),
               "frame_to_rgb8")
    return out


def depth_to_u8(depth, flip_w: bool = True):
    """ced_depth_to_u8: [H,W] (or [H,W,1]) depths -> [H,W] uint8 of the min-max normalised image (depth2img of
    train_real.py:38-41 before cv2's colour-map lookup)."""
    _chk(depth, torch.float32, "depth")
    if depth.dim() == 3:
        assert depth.shape[2] == 1
        depth = depth[..., 0]
    assert depth.dim() == 2, "depth_to_u8: depth [H,W]"
    out = torch.empty(depth.shape, device=depth.device, dtype=torch.uint8)
    ws = torch.empty((2,), device=depth.device, dtype=torch.int32)
    _lib.check(_lib.lib().ced_depth_to_u8(depth.shape[0], depth.shape[1], _p(depth), int(bool(flip_w)), _p(out), _p(ws),
                                          _stream()), "depth_to_u8")
    return out


def scatter_pixels(dest, n_pixels: int, src_rgb, src_opacity, src_depth, want_rgb8_width: int = 0, flip_w: bool = True):
    """ced_scatter_pixels: rows (marching / gather order) -> raster images.  Sources are 2-D views whose last-dim
    slices may be strided (e.g. columns of the gathered [rows, 5] payload).  Returns (rgb [n,3], opacity [n,1],
    depth [n,1], rgb8 [n,3] uint8 or None)."""
    _chk(dest, torch.int64, "dest")
    n_rows = dest.shape[0]
    for nm, t in (("src_rgb", src_rgb), ("src_opacity", src_opacity), ("src_depth", src_depth)):
        assert t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape[0] == n_rows and t.stride(1) == 1, nm
        _check_current_device(t.device.index, nm)
    dev = dest.device
    rgb = torch.empty((n_pixels, 3), device=dev, dtype=torch.float32)
    opacity = torch.empty((n_pixels, 1), device=dev, dtype=torch.float32)
    depth = torch.empty((n_pixels, 1), device=dev, dtype=torch.float32)
    rgb8 = torch.empty((n_pixels, 3), device=dev, dtype=torch.uint8) if want_rgb8_width else None
    rc = _lib.lib().ced_scatter_pixels(n_rows, _p(src_rgb), src_rgb.stride(0), _p(src_opacity), src_opacity.stride(0),
                                       _p(src_depth), src_depth.stride(0), _p(dest), n_pixels, _p(rgb), _p(opacity),
                                       _p(depth), _p(rgb8), int(want_rgb8_width), int(bool(flip_w)), _stream())
    _lib.check(rc, "scatter_pixels")
    return rgb, opacity, depth, rgb8


def linear(x, w, transpose_w: bool = False, relu: bool = False, mask=None):
    """ced_linear: y = x w^T (w [n_out, n_in]) or, transpose_w, y = x w (w [n_in, n_out]); optional ReLU on the result
    and optional mask (y *= mask > 0, the fused ReLU derivative of the layer below)."""
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "w"); _chk(mask, torch.float32, "mask", allow_none=True)
    assert x.dim() == 2 and w.dim() == 2
    n, n_in = x.shape
    n_out = w.shape[1] if transpose_w else w.shape[0]
    if mask is not None:
        assert mask.shape == (n, n_out), f"mask {tuple(mask.shape)} vs output {(n, n_out)}"
    y = torch.empty((n, n_out), device=x.device, dtype=torch.float32)
    rc = _lib.lib().ced_linear(n, _p(x), n_in, _p(w), w.shape[0], w.shape[1], int(bool(transpose_w)), n_out, int(bool(relu)),
                               _p(mask), _p(y), _stream())
    _lib.check(rc, "linear")
    return y


def mlp_chain(x, weights, backward: bool = False, masks=None, want=None, relu_last: bool = False):
    """ced_mlp_chain.  Forward: x [n, K0], weights [W_1 .. W_L] (W_l [N_l, N_{l-1}]) -> [a_1 .. a_L] (ReLU on all but the
    last).  Backward: x = dy [n, N_L], masks[l] = the forward input of layer l (None for no mask), want[l] = whether the
    gradient with respect to layer l's input is to be stored -> [g_0 .. g_{L-1}] (None where not wanted)."""
    _chk(x, torch.float32, "x")
    L = len(weights)
    assert x.dim() == 2 and 1 <= L <= 6
    for w in weights:
        _chk(w, torch.float32, "w")
    widths = [weights[0].shape[1]] + [w.shape[0] for w in weights]
    for l in range(1, L):
        assert weights[l].shape[1] == widths[l], "mlp_chain: layer widths do not chain"
    n = x.shape[0]
    dev = x.device
    if not backward:
        assert x.shape[1] == widths[0]
        outs = [torch.empty((n, widths[l + 1]), device=dev, dtype=torch.float32) for l in range(L)]
        mk = [None] * L
    else:
        assert x.shape[1] == widths[L]
        want = [True] * L if want is None else list(want)
        outs = [torch.empty((n, widths[l]), device=dev, dtype=torch.float32) if want[l] else None for l in range(L)]
        mk = [None] * L if masks is None else list(masks)
        for l in range(L):
            _chk(mk[l], torch.float32, "mask", allow_none=True)
            assert mk[l] is None or mk[l].shape == (n, widths[l])
    vp = lambda ts: (C.c_void_p * L)(*[(t.data_ptr() if t is not None and t.numel() > 0 else None) for t in ts])
    rc = _lib.lib().ced_mlp_chain(n, L, int(bool(backward)), _p(x), (C.c_int32 * (L + 1))(*widths), vp(weights), vp(outs),
                                  vp(mk), int(bool(relu_last)), _stream())
    _lib.check(rc, "mlp_chain")
    return outs


def mlp_backward_dw_supported(widths) -> bool:
    """Shapes ced_mlp_backward_dw handles: input <= 48, 1..3 hidden layers of 64, output <= 32."""
    L = len(widths) - 1
    return 2 <= L <= 4 and 1 <= widths[0] <= 48 and 1 <= widths[L] <= 32 and all(w == 64 for w in widths[1:L])


def mlp_backward_dw(dy, weights, acts, want_g0: bool):
    """ced_mlp_backward_dw: dy [n, N_L], weights [W_0 .. W_H], acts [x, a_1 .. a_H] (the forward's layer inputs) ->
    (g0 [n, K0] or None, [dW_0 .. dW_H])."""
    _chk(dy, torch.float32, "dy")
    L = len(weights)
    widths = [weights[0].shape[1]] + [w.shape[0] for w in weights]
    assert mlp_backward_dw_supported(widths) and len(acts) == L
    n, dev = dy.shape[0], dy.device
    for w, a, k in zip(weights, acts, widths):
        _chk(w, torch.float32, "w"); _chk(a, torch.float32, "act")
        assert a.shape == (n, k), f"{a.shape} v.s. {(n, k)}"
    assert dy.shape == (n, widths[L])
    total = sum(widths[l] * widths[l + 1] for l in range(L))
    dws = torch.empty((total,), device=dev, dtype=torch.float32)
    g0 = torch.empty((n, widths[0]), device=dev, dtype=torch.float32) if want_g0 else None
    wa = (C.c_int32 * (L + 1))(*widths)
    nbytes = int(_lib.lib().ced_mlp_backward_dw_workspace_bytes(n, L, wa))
    ws = torch.empty((max(nbytes, 4) // 4,), device=dev, dtype=torch.float32)
    vp = lambda ts: (C.c_void_p * L)(*[(t.data_ptr() if t.numel() > 0 else None) for t in ts])
    rc = _lib.lib().ced_mlp_backward_dw(n, L, _p(dy), wa, vp(weights), vp(acts), _p(g0), _p(dws), _p(ws), nbytes, _stream())
    _lib.check(rc, "mlp_backward_dw")
    out, o = [], 0
    for l in range(L):
        out.append(dws[o:o + widths[l] * widths[l + 1]].view(widths[l + 1], widths[l]))
        o += widths[l] * widths[l + 1]
    return g0, out


def train_inputs(n, rays_o=None, rays_d=None, ray_indices=None, t_starts=None, t_ends=None, timestamps=None,
                 positions=None, directions=None):
    """ced_train_inputs -> (pos [n,3], enc [n,32], sh [n,4], t [n]).  Rays mode (ray_indices given: int64 [n],
    timestamps per RAY) or explicit mode (positions / directions [n,3], timestamps per sample)."""
    ref = ray_indices if ray_indices is not None else positions
    dev = ref.device
    f32 = torch.float32
    for t_, nm in ((rays_o, "rays_o"), (rays_d, "rays_d"), (t_starts, "t_starts"), (t_ends, "t_ends"),
                   (timestamps, "timestamps"), (positions, "positions"), (directions, "directions")):
        _chk(t_, f32, nm, allow_none=True)
    _chk(ray_indices, torch.int64, "ray_indices", allow_none=True)
    pos = torch.empty((n, 3), device=dev, dtype=f32)
    enc = torch.empty((n, 32), device=dev, dtype=f32)
    sh = torch.empty((n, 4), device=dev, dtype=f32)
    t = torch.empty((n,), device=dev, dtype=f32)
    rc = _lib.lib().ced_train_inputs(n, _p(rays_o), _p(rays_d), _p(ray_indices), _p(t_starts), _p(t_ends), _p(timestamps),
                                     _p(positions), _p(directions), _p(pos), _p(enc), _p(sh), _p(t), _stream())
    _lib.check(rc, "train_inputs")
    return pos, enc, sh, t


def train_warp(pos, mo, aabb6, moving_step: float, use_div_offsets: bool):
    """ced_train_warp -> (xn clamped [n,3], move [n,3], selector [n] as 0 / 1 floats).  aabb6: six python floats."""
    _chk(pos, torch.float32, "pos"); _chk(mo, torch.float32, "mo")
    n = pos.shape[0]
    xn, move = torch.empty_like(pos), torch.empty_like(pos)
    sel = torch.empty((n,), device=pos.device, dtype=torch.float32)
    rc = _lib.lib().ced_train_warp(n, _p(pos), _p(mo), mo.shape[1], int(bool(use_div_offsets)), float(moving_step),
                                   (C.c_float * 6)(*aabb6), _p(xn), _p(move), _p(sel), _stream())
    _lib.check(rc, "train_warp")
    return xn, move, sel


def train_warp_backward(pos, mo, aabb6, moving_step: float, use_div_offsets: bool, d_xn, d_move=None):
    _chk(pos, torch.float32, "pos"); _chk(mo, torch.float32, "mo"); _chk(d_xn, torch.float32, "d_xn")
    _chk(d_move, torch.float32, "d_move", allow_none=True)
    d_mo = torch.empty_like(mo)
    rc = _lib.lib().ced_train_warp_backward(pos.shape[0], _p(pos), _p(mo), mo.shape[1], int(bool(use_div_offsets)),
                                            float(moving_step), (C.c_float * 6)(*aabb6), _p(d_xn), _p(d_move), _p(d_mo),
                                            _stream())
    _lib.check(rc, "train_warp_backward")
    return d_mo


def train_head_in(bout, sh, selector):
    """ced_train_head_in -> (head_in [n,19], sigma [n])."""
    _chk(bout, torch.float32, "bout"); _chk(sh, torch.float32, "sh"); _chk(selector, torch.float32, "selector")
    assert bout.dim() == 2 and bout.shape[1] == 16 and sh.shape == (bout.shape[0], 4)
    n = bout.shape[0]
    head_in = torch.empty((n, 19), device=bout.device, dtype=torch.float32)
    sigma = torch.empty((n,), device=bout.device, dtype=torch.float32)
    rc = _lib.lib().ced_train_head_in(n, _p(bout), _p(sh), _p(selector), _p(head_in), _p(sigma), _stream())
    _lib.check(rc, "train_head_in")
    return head_in, sigma


def train_head_in_backward(bout, selector, d_head_in, d_sigma):
    _chk(bout, torch.float32, "bout"); _chk(selector, torch.float32, "selector")
    _chk(d_head_in, torch.float32, "d_head_in", allow_none=True); _chk(d_sigma, torch.float32, "d_sigma", allow_none=True)
    d_bout = torch.empty_like(bout)
    rc = _lib.lib().ced_train_head_in_backward(bout.shape[0], _p(bout), _p(selector), _p(d_head_in), _p(d_sigma), _p(d_bout),
                                               _stream())
    _lib.check(rc, "train_head_in_backward")
    return d_bout


def weight_grad(x, dy):
    """ced_weight_grad: dW [n_out, n_in] = dy^T x over the sample stream (x [S, n_in], dy [S, n_out], fp32)."""
    _chk(x, torch.float32, "x"); _chk(dy, torch.float32, "dy")
    assert x.dim() == 2 and dy.dim() == 2 and x.shape[0] == dy.shape[0], "weight_grad: x [S, n_in], dy [S, n_out]"
    n, n_in, n_out = x.shape[0], x.shape[1], dy.shape[1]
    if x.data_ptr() % 16:
        x = x.clone()
    if dy.data_ptr() % 16:
        dy = dy.clone()
    dw = torch.empty((n_out, n_in), device=x.device, dtype=torch.float32)
    nbytes = int(_lib.lib().ced_weight_grad_workspace_bytes(n, n_out, n_in))
    ws = torch.empty((max(nbytes, 4) // 4,), device=x.device, dtype=torch.float32)
    rc = _lib.lib().ced_weight_grad(n, _p(x), n_in, _p(dy), n_out, _p(dw), _p(ws), nbytes, _stream())
    _lib.check(rc, "weight_grad")
    return dw


def composite_prefix_(packed_info, t_starts, t_ends, sigmas, rgbs, rgb, opacity, depth):
    _chk(packed_info, torch.int64, "packed_info"); _chk(rgbs, torch.float32, "rgbs")
    for nm, t in (("rgb", rgb), ("opacity", opacity), ("depth", depth)):
        _chk(t, torch.float32, nm)
    if t_starts.shape[0] == 0:
        return
    with profiling.span("composite", t_starts.shape[0]):
        rc = _lib.lib().ced_composite_prefix(packed_info.shape[0], _p(packed_info), _p(t_starts), _p(t_ends),
                                             _p(sigmas), _p(rgbs), _p(rgb), _p(opacity), _p(depth), _stream())
    _lib.check(rc, "composite_prefix")


def composite_step_(packed_info, t_starts, t_ends, sigmas, rgbs, rgb, opacity, depth, opc_thres, n_samples_iter,
                    ray_mask, stats):
    """composite_prefix_ + mask / alive-count / sample-count bookkeeping (cednerf/utils.py:301-307)."""
    _chk(packed_info, torch.int64, "packed_info"); _chk(stats, torch.int64, "stats")
    for nm, t in (("rgb", rgb), ("opacity", opacity), ("depth", depth)):
        _chk(t, torch.float32, nm)
    assert ray_mask.dtype == torch.bool and ray_mask.is_cuda and ray_mask.shape[0] == packed_info.shape[0]
    with profiling.span("composite", 0):
        rc = _lib.lib().ced_composite_step(packed_info.shape[0], _p(packed_info), _p(t_starts), _p(t_ends),
                                           _p(sigmas), _p(rgbs), _p(rgb), _p(opacity), _p(depth), float(opc_thres),
                                           int(n_samples_iter), _p(ray_mask), _p(stats), _stream())
    _lib.check(rc, "composite_step")


def composite_test_(sigmas, rgbs, t_start, t_end, pack_info, alive_indices, T_threshold, alpha_threshold, opacity,
                    depth, rgb):
    _chk(pack_info, torch.int64, "pack_info"); _chk(alive_indices, torch.int64, "alive_indices")
    rc = _lib.lib().ced_composite_test(alive_indices.shape[0], _p(sigmas), _p(rgbs), _p(t_start), _p(t_end),
                                       _p(pack_info), _p(alive_indices), float(T_threshold), float(alpha_threshold),
                                       _p(opacity), _p(depth), _p(rgb), _stream())
    _lib.check(rc, "composite_test")


def finalize_pixels_(bkgd, rgb, opacity, depth):
    _chk(bkgd, torch.float32, "render_bkgd", allow_none=True)
    rc = _lib.lib().ced_finalize_pixels(rgb.shape[0], _p(bkgd), _p(rgb), _p(opacity), _p(depth), _stream())
    _lib.check(rc, "finalize_pixels")


# ----------------------------------------------------------------------------------------------
# frame renderer (the whole render_image_test loop in one native call)
# ----------------------------------------------------------------------------------------------
def build_occupancy_accel(binaries: torch.Tensor) -> torch.Tensor:
    """ced_build_occupancy_accel: the brick distance field of an occupancy grid [m,res,res,res] (bool / uint8), as a
    uint8 tensor to hand to the frame renderer.  Rebuild it when the grid changes (train_real.py:332-336)."""
    assert binaries.is_cuda and binaries.is_contiguous() and binaries.ndim == 4
    _check_current_device(binaries.device.index, "binaries")
    m, res = binaries.shape[0], binaries.shape[1]
    L = _lib.lib()
    need = int(L.ced_occupancy_accel_bytes(m, res))
    if need < 0:
        raise ValueError("build_occupancy_accel: unsupported grid size")
    accel = torch.empty((need,), device=binaries.device, dtype=torch.uint8)
    _lib.check(L.ced_build_occupancy_accel(_p(_as_u8(binaries)), m, res, _p(accel), need, _stream()), "build_occupancy_accel")
    return accel


def _with_workgroups(desc: _lib.FieldDesc, max_workgroups: int) -> _lib.FieldDesc:
    """The descriptor with its per-call launch property `max_workgroups` set (a copy when it differs)."""
    if int(max_workgroups) == int(desc.max_workgroups):
        return desc
    d2 = _lib.FieldDesc()
    C.memmove(C.byref(d2), C.byref(desc), C.sizeof(_lib.FieldDesc))
    d2.max_workgroups = int(max_workgroups)
    return d2


class FrameTracer:
    """Event pairs + per-iteration counters for ced_render_image_test (see ced_frame_trace)."""

    def __init__(self, capacity: int = 96, with_events: bool = True):
        self.capacity = capacity
        self.struct = _lib.FrameTrace()
        self.struct.capacity = capacity
        self._alive = (C.c_int64 * capacity)()
        self._nsamp = (C.c_int64 * capacity)()
        self._samples = (C.c_int64 * capacity)()
        self.struct.iter_alive = self._alive
        self.struct.iter_n_samples = self._nsamp
        self.struct.iter_samples = self._samples
        self.events = None
        if with_events:
            self.events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                           for _ in range(capacity)]
            for a, b in self.events:       # torch creates the hipEvent lazily on the first record
                a.record(); b.record()
            self._b = (C.c_void_p * capacity)(*[e[0].cuda_event for e in self.events])
            self._e = (C.c_void_p * capacity)(*[e[1].cuda_event for e in self.events])
            self.struct.field_begin = self._b
            self.struct.field_end = self._e

        self.stamps = None

    def enable_device_stamps(self, device) -> None:
        """Have the field kernel itself stamp, per iteration, when its first workgroup started and its last one finished
        (device wall clock): the interval the kernel was executing -- the event pair also counts its wait for CUs."""
        self.stamps = torch.zeros((self.capacity, 2), device=device, dtype=torch.int64)
        self.struct.field_stamps = C.c_void_p(self.stamps.data_ptr())

    def field_intervals_device(self):
        """(start, end) of every field launch in ms on the device wall clock (after the stream has been synchronised);
        None for an iteration whose launch had no workgroup with work."""
        n = min(int(self.struct.n_iters), self.capacity)
        khz = int(_lib.lib().ced_wall_clock_khz())
        st = self.stamps[:n].cpu().numpy().astype(np.uint64)
        out = []
        for a, b in st:
            out.append(None if (a == np.uint64(0xffffffffffffffff) or b == 0) else (float(a) / khz, float(b) / khz))
        return out

    def iterations(self):
        n = min(int(self.struct.n_iters), self.capacity)
        return [dict(n_alive=int(self._alive[i]), n_samples=int(self._nsamp[i]), n_new=int(self._samples[i]))
                for i in range(n)]

    def field_intervals(self, ref: "torch.cuda.Event"):
        """(begin, end) of every field launch in ms after `ref` (an event recorded earlier on the device)."""
        n = min(int(self.struct.n_iters), self.capacity)
        return [(ref.elapsed_time(self.events[i][0]), ref.elapsed_time(self.events[i][1])) for i in range(n)]

    def field_ms(self):
        """Per-iteration field-kernel durations (ms); call after the stream has been synchronised."""
        n = min(int(self.struct.n_iters), self.capacity)
        return [self.events[i][0].elapsed_time(self.events[i][1]) for i in range(n)]


_frame_ws = {}
_frame_ws_lock = __import__("threading").Lock()


def render_image_test_native(desc: _lib.FieldDesc, rays_o, rays_d, binaries, aabbs, near_plane, far_plane,
                             render_step_size, cone_angle, early_stop_eps, max_samples, timestamps, t_per_ray, bkgd,
                             tracer: Optional[FrameTracer] = None, field_stream: Optional[torch.cuda.Stream] = None,
                             accel: Optional[torch.Tensor] = None, max_workgroups: int = 0):
    """ced_render_image_test.  Returns (rgb [n,3], opacity [n,1], depth [n,1], total_samples).
    field_stream: optional stream shared by concurrently rendered frames for their field kernels; accel: the grid's
    brick distance field (build_occupancy_accel; None = built inside the call); max_workgroups: workgroups of a field
    launch (0 = one per CU)."""
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d")
    _chk(aabbs, torch.float32, "aabbs"); _chk(timestamps, torch.float32, "timestamps")
    _chk(bkgd, torch.float32, "render_bkgd", allow_none=True)
    assert rays_o.ndim == 2 and rays_o.shape[1] == 3 and rays_o.shape == rays_d.shape
    assert binaries.is_cuda and binaries.is_contiguous() and binaries.ndim == 4
    n = rays_o.shape[0]
    m, res = binaries.shape[0], binaries.shape[1]
    assert aabbs.shape == (m, 6)
    if t_per_ray:
        assert timestamps.numel() == n, "per-ray timestamps must have one entry per ray"
    dev = rays_o.device
    L = _lib.lib()
    need = int(L.ced_render_image_test_workspace_bytes(n, m, res, float(cone_angle), int(max_samples)))
    if need < 0:
        raise ValueError("render_image_test: unsupported sizes")
    # one workspace + pinned hand-shake buffer per (device, stream): frames in flight on different
    # streams (PipelinedRenderer) never share them
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    with _frame_ws_lock:
        ws = _frame_ws.get(key)
        if ws is None or ws[0].numel() < need:
            ws = (torch.empty((max(need, 1),), device=dev, dtype=torch.uint8),
                  ws[1] if ws is not None else torch.zeros((512,), dtype=torch.int64).pin_memory())
            _frame_ws[key] = ws
    rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
    opacity = torch.empty((n, 1), device=dev, dtype=torch.float32)
    depth = torch.empty((n, 1), device=dev, dtype=torch.float32)
    total = C.c_int64(0)
    rc = L.ced_render_image_test(C.byref(_with_workgroups(desc, max_workgroups)), n, _p(rays_o), _p(rays_d),
                                 _p(_as_u8(binaries)), m, res, _p(aabbs), _p(accel), float(near_plane), float(far_plane), float(render_step_size), float(cone_angle),
                                 float(early_stop_eps), int(max_samples), _p(timestamps), int(bool(t_per_ray)), _p(bkgd),
                                 _p(rgb), _p(opacity), _p(depth), _p(ws[0]), ws[0].numel(), C.c_void_p(ws[1].data_ptr()),
                                 C.byref(total), C.byref(tracer.struct) if tracer is not None else None,
                                 C.c_void_p(field_stream.cuda_stream) if field_stream is not None else None, _stream())
    _lib.check(rc, "render_image_test")
    return rgb, opacity, depth, int(total.value)


def march_all(rays_o, rays_d, binaries, aabbs, accel, near_planes, far_plane: float, step_size: float, cone_angle: float,
              want_ray_indices: bool = True, t_sorted=None, t_indices=None, hits=None):
    """ced_march_all: every ray marched to the far plane on the accelerated walk.  Several grid levels need the sorted
    ray / box events (t_sorted, t_indices, hits).  Returns (t_starts, t_ends, ray_indices or None, packed_info [n,2])."""
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d")
    _chk(aabbs, torch.float32, "aabbs"); _chk(near_planes, torch.float32, "near_planes")
    assert binaries.is_cuda and binaries.is_contiguous() and binaries.ndim == 4
    assert accel is not None and accel.is_cuda
    n = rays_o.shape[0]
    m, res = binaries.shape[0], binaries.shape[1]
    if m > 1:
        _chk(t_sorted, torch.float32, "t_sorted"); _chk(t_indices, torch.int64, "t_indices")
        assert t_sorted.shape == (n, 2 * m) and t_indices.shape == (n, 2 * m) and hits.shape == (n, m) and hits.is_contiguous()
    dev = rays_o.device
    L = _lib.lib()
    packed = torch.zeros((n, 2), device=dev, dtype=torch.int64)
    args = (n, _p(rays_o), _p(rays_d), _p(_as_u8(binaries)), m, res, _p(aabbs), _p(accel), _p(near_planes), float(far_plane),
            float(step_size), float(cone_angle), _p(t_sorted) if m > 1 else None, _p(t_indices) if m > 1 else None,
            _p(_as_u8(hits)) if m > 1 else None)
    _lib.check(L.ced_march_all(*args, 0, _p(packed), None, None, None, 0, None, _stream()), "march_all (count)")
    counts = packed[:, 1]
    incl = torch.cumsum(counts, 0)
    packed[:, 0] = incl - counts
    total = int(incl[-1].item()) if n > 0 else 0
    t_starts = torch.empty((total,), device=dev, dtype=torch.float32)
    t_ends = torch.empty((total,), device=dev, dtype=torch.float32)
    ray_indices = torch.empty((total,), device=dev, dtype=torch.int64) if want_ray_indices else None
    if total > 0:
        _lib.check(L.ced_march_all(*args, 1, _p(packed), _p(t_starts), _p(t_ends), _p(ray_indices), 0, None, _stream()),
                   "march_all (fill)")
    return t_starts, t_ends, ray_indices, packed


def march_all_onepass(rays_o, rays_d, binaries, aabbs, accel, near_planes, far_plane: float, step_size: float,
                      cone_angle: float, capacity: int, t_sorted=None, t_indices=None, hits=None):
    """ced_march_all, one pass (fill = 2) into arrays of `capacity` samples.  Returns (t_starts, t_ends, packed_info,
    total) with `total` a DEVICE int64 scalar: the samples marched; > capacity means some rays stored nothing.  Rays'
    ranges are in workgroup arrival order (not sorted by ray)."""
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d")
    _chk(aabbs, torch.float32, "aabbs"); _chk(near_planes, torch.float32, "near_planes")
    assert binaries.is_cuda and binaries.is_contiguous() and binaries.ndim == 4 and accel is not None and accel.is_cuda
    n = rays_o.shape[0]
    m, res = binaries.shape[0], binaries.shape[1]
    if m > 1:
        _chk(t_sorted, torch.float32, "t_sorted"); _chk(t_indices, torch.int64, "t_indices")
        assert t_sorted.shape == (n, 2 * m) and t_indices.shape == (n, 2 * m) and hits.shape == (n, m) and hits.is_contiguous()
    dev = rays_o.device
    packed = torch.empty((n, 2), device=dev, dtype=torch.int64)
    total = torch.zeros((1,), device=dev, dtype=torch.int64)
    t_starts = torch.empty((capacity,), device=dev, dtype=torch.float32)
    t_ends = torch.empty((capacity,), device=dev, dtype=torch.float32)
    rc = _lib.lib().ced_march_all(n, _p(rays_o), _p(rays_d), _p(_as_u8(binaries)), m, res, _p(aabbs), _p(accel),
                                  _p(near_planes), float(far_plane), float(step_size), float(cone_angle),
                                  _p(t_sorted) if m > 1 else None, _p(t_indices) if m > 1 else None,
                                  _p(_as_u8(hits)) if m > 1 else None, 2, _p(packed), _p(t_starts), _p(t_ends), None,
                                  int(capacity), _p(total), _stream())
    _lib.check(rc, "march_all (one pass)")
    return t_starts, t_ends, packed, total


_image_ws: Dict[tuple, tuple] = {}


def render_image_eval_native(desc: _lib.FieldDesc, rays_o, rays_d, packed_info, t_starts, t_ends, early_stop_eps,
                             alpha_thre, timestamps, t_per_ray, bkgd, chunk_rays: int = 0, max_workgroups: int = 0):
    """ced_render_image + ced_render_image_gather: `sampling` (visibility filter) and `rendering` of cednerf/utils.py:115-133
    from one field evaluation per sample.  (packed_info, t_starts, t_ends): the one-shot march of all rays.
    Returns (rgb [n,3], opacity [n,1], depth [n,1], extras, ray_offsets [n+1] int64, stats) where extras holds the
    kept samples' ray_indices (relative to chunks of `chunk_rays` rays when > 0), t_starts, t_ends, sigmas, rgbs,
    weights, trans, alphas in the reference's order, and stats = (samples evaluated, iterations)."""
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d")
    _chk(packed_info, torch.int64, "packed_info"); _chk(t_starts, torch.float32, "t_starts")
    _chk(t_ends, torch.float32, "t_ends"); _chk(timestamps, torch.float32, "timestamps")
    _chk(bkgd, torch.float32, "render_bkgd", allow_none=True)
    assert rays_o.ndim == 2 and rays_o.shape[1] == 3 and rays_o.shape == rays_d.shape
    n = rays_o.shape[0]
    n_all = t_starts.shape[0]
    assert packed_info.shape == (n, 2) and t_ends.shape == (n_all,)
    if t_per_ray:
        assert timestamps.numel() == n, "per-ray timestamps must have one entry per ray"
    dev = rays_o.device
    L = _lib.lib()
    need = int(L.ced_render_image_workspace_bytes(n, n_all))
    if need < 0:
        raise ValueError("render_image: unsupported sizes")
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    with _frame_ws_lock:
        ws = _image_ws.get(key)
        if ws is None or ws[0].numel() < need:
            ws = (torch.empty((max(need, 1),), device=dev, dtype=torch.uint8),
                  ws[1] if ws is not None else torch.zeros((512,), dtype=torch.int64).pin_memory())
            _image_ws[key] = ws
    rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
    opacity = torch.empty((n, 1), device=dev, dtype=torch.float32)
    depth = torch.empty((n, 1), device=dev, dtype=torch.float32)
    kept = torch.empty((n,), device=dev, dtype=torch.int32)
    stats = (C.c_int64 * 3)()
    rc = L.ced_render_image(C.byref(_with_workgroups(desc, max_workgroups)), n, _p(rays_o), _p(rays_d), n_all,
                            _p(packed_info), _p(t_starts), _p(t_ends), float(early_stop_eps), float(alpha_thre),
                            _p(timestamps), int(bool(t_per_ray)), _p(bkgd), _p(rgb), _p(opacity), _p(depth), _p(kept),
                            _p(ws[0]), ws[0].numel(), C.c_void_p(ws[1].data_ptr()), stats, None, _stream())
    _lib.check(rc, "render_image")
    processed = int(stats[0])
    offsets = torch.zeros((n + 1,), device=dev, dtype=torch.int64)
    torch.cumsum(kept, 0, out=offsets[1:])
    total = int(stats[2])                    # = offsets[-1], without another device round trip
    f = lambda *shape: torch.empty(shape, device=dev, dtype=torch.float32)
    extras = {"ray_indices": torch.empty((total,), device=dev, dtype=torch.int64), "t_starts": f(total), "t_ends": f(total),
              "sigmas": f(total), "rgbs": f(total, 3), "weights": f(total), "trans": f(total), "alphas": f(total)}
    if total > 0:               # (an alpha threshold can drop every sample of a faint scene)
        rc = L.ced_render_image_gather(n, n_all, processed, _p(ws[0]), ws[0].numel(), _p(offsets), int(chunk_rays),
                                       _p(extras["ray_indices"]), _p(extras["t_starts"]), _p(extras["t_ends"]),
                                       _p(extras["sigmas"]), _p(extras["rgbs"]), _p(extras["weights"]), _p(extras["trans"]),
                                       _p(extras["alphas"]), _stream())
        _lib.check(rc, "render_image_gather")
    return rgb, opacity, depth, extras, offsets, (processed, int(stats[1]))


def sampling_native(desc: _lib.FieldDesc, rays_o, rays_d, packed_info, t_starts, t_ends, early_stop_eps, alpha_thre,
                    timestamps, t_per_ray, max_workgroups: int = 0):
    """The visibility filter of OccGridEstimator.sampling (sigma_fn = the fused field) on ced_render_image's
    sampling-only mode: density evaluated front to back, rays stopped at the transmittance threshold.
    Returns the surviving (ray_indices, t_starts, t_ends) -- those of the filter over every marched sample."""
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d")
    _chk(packed_info, torch.int64, "packed_info"); _chk(t_starts, torch.float32, "t_starts")
    _chk(t_ends, torch.float32, "t_ends"); _chk(timestamps, torch.float32, "timestamps")
    n = rays_o.shape[0]
    n_all = t_starts.shape[0]
    assert packed_info.shape == (n, 2) and t_ends.shape == (n_all,)
    if t_per_ray:
        assert timestamps.numel() == n, "per-ray timestamps must have one entry per ray"
    dev = rays_o.device
    L = _lib.lib()
    need = int(L.ced_render_image_workspace_bytes(n, n_all))
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    with _frame_ws_lock:
        ws = _image_ws.get(key)
        if ws is None or ws[0].numel() < need:
            ws = (torch.empty((max(need, 1),), device=dev, dtype=torch.uint8),
                  ws[1] if ws is not None else torch.zeros((512,), dtype=torch.int64).pin_memory())
            _image_ws[key] = ws
    kept = torch.empty((n,), device=dev, dtype=torch.int32)
    stats = (C.c_int64 * 3)()
    rc = L.ced_render_image(C.byref(_with_workgroups(desc, max_workgroups)), n, _p(rays_o), _p(rays_d), n_all,
                            _p(packed_info), _p(t_starts), _p(t_ends), float(early_stop_eps), float(alpha_thre),
                            _p(timestamps), int(bool(t_per_ray)), None, None, None, None, _p(kept),
                            _p(ws[0]), ws[0].numel(), C.c_void_p(ws[1].data_ptr()), stats, None, _stream())
    _lib.check(rc, "render_image (sampling)")
    offsets = torch.zeros((n + 1,), device=dev, dtype=torch.int64)
    torch.cumsum(kept, 0, out=offsets[1:])
    total = int(offsets[-1].item()) if n > 0 else 0
    ray_indices = torch.empty((total,), device=dev, dtype=torch.int64)
    t0 = torch.empty((total,), device=dev, dtype=torch.float32)
    t1 = torch.empty((total,), device=dev, dtype=torch.float32)
    if total > 0:
        rc = L.ced_render_image_gather(n, n_all, int(stats[0]), _p(ws[0]), ws[0].numel(), _p(offsets), 0, _p(ray_indices),
                                       _p(t0), _p(t1), None, None, None, None, None, _stream())
        _lib.check(rc, "render_image_gather (sampling)")
    return ray_indices, t0, t1


class ScheduleExchange:
    """ced_shard_exchange for frames whose rays are dealt over the ranks of a torch.distributed group: per iteration the
    library hands the row of per-frame survivor counts to `reduce`, which enqueues an all-reduce(sum) of it on the
    calling thread's current stream (the stream the native call runs on), so that every rank's scheduling launch sees
    the image-global N_alive of cednerf/utils.py:231-235.

    One instance per concurrently running native call (a lane of PipelinedRenderer): the per-iteration rows live in a
    buffer of its own.  Collectives of one communicator must be issued in the same order on every rank, and the lanes'
    threads run independently -- so with several lanes NO lane issues its own collective: `issuer` (set by
    PipelinedRenderer) hands the row to the ONE collecting thread, which issues every lane's all-reduces and pixel gathers
    on ONE process group in an order that depends on the lanes' own message sequences only, never on timing
    (dist.PipelinedRenderer.render_steps).  Without an issuer (a renderer on its own) the all-reduce is issued here.

    global_rays: rays of the whole image; local_rays [n_frames]: the real (unpadded) rays of this rank's share of every
    frame.  `reduce_fn(tensor)` replaces the all-reduce (tests)."""

    def __init__(self, n_frames: int, global_rays: int, local_rays, device, max_samples: int, cone_angle: float,
                 group=None, reduce_fn=None):
        import torch.distributed as dist
        L = _lib.lib()
        self.n_frames = int(n_frames)
        self.iterations = int(L.ced_render_frames_test_iterations(float(cone_angle), int(max_samples)))
        self.host_words = (int(L.ced_render_frames_test_host_bytes(float(cone_angle), int(max_samples))) + 7) // 8
        self.counts = torch.zeros((self.iterations + 1, self.n_frames), device=device, dtype=torch.int64)
        self.local_rays = torch.as_tensor(local_rays, dtype=torch.int32).reshape(-1).to(device).contiguous()
        assert self.local_rays.numel() == self.n_frames
        self.group, self.error, self.calls = group, None, 0
        self._reduce_fn = reduce_fn
        self.issuer = None             # callable(exchange, row, stream_handle, iteration): see the class comment

        def _cb(user, counts_ptr, n_counts, iteration, stream):
            try:
                row = self.counts[iteration]
                assert row.data_ptr() == counts_ptr and n_counts == self.n_frames
                if self.issuer is not None and self._reduce_fn is None:
                    self.issuer(self, row, stream or 0, iteration)     # returns once the collective is enqueued on `stream`
                    self.calls += 1
                    return 0
                cur = torch.cuda.current_stream(row.device)
                if cur.cuda_stream != (stream or 0):           # not the thread's current stream: make it so
                    cur = torch.cuda.ExternalStream(stream, device=row.device)
                with torch.cuda.stream(cur):
                    if self._reduce_fn is not None:
                        self._reduce_fn(row)
                    else:
                        dist.all_reduce(row, op=dist.ReduceOp.SUM, group=self.group)
                self.calls += 1
                return 0
            except BaseException as e:                         # never let an exception cross the C frame
                self.error = e
                return 1

        self._cb = _lib.EXCHANGE_FN(_cb)
        self.struct = _lib.ShardExchange(int(global_rays), C.c_void_p(self.local_rays.data_ptr()),
                                         C.c_void_p(self.counts.data_ptr()), self._cb, None)


def render_frames_test_native(desc: _lib.FieldDesc, n_frames: int, rays_o, rays_d, binaries, aabbs, near_plane, far_plane,
                              render_step_size, cone_angle, early_stop_eps, max_samples, frame_times, bkgd,
                              tracer: Optional[FrameTracer] = None, field_stream: Optional[torch.cuda.Stream] = None,
                              accel: Optional[torch.Tensor] = None, max_workgroups: int = 0,
                              exchange: Optional[ScheduleExchange] = None):
    """ced_render_frames_test: `n_frames` frames (rays frame-major, [n_frames * rays_per_frame, 3]) through shared
    launches, every frame on its own schedule.  Returns (rgb [N,3], opacity [N,1], depth [N,1], [total_samples per frame]).
    exchange: the frames are this rank's shares of frames sharded over several ranks (ced_render_frames_test_sharded):
    one image-global schedule per frame."""
    _chk(rays_o, torch.float32, "rays_o"); _chk(rays_d, torch.float32, "rays_d")
    _chk(aabbs, torch.float32, "aabbs"); _chk(frame_times, torch.float32, "frame_times")
    _chk(bkgd, torch.float32, "render_bkgd", allow_none=True)
    assert rays_o.ndim == 2 and rays_o.shape[1] == 3 and rays_o.shape == rays_d.shape
    assert binaries.is_cuda and binaries.is_contiguous() and binaries.ndim == 4
    n = rays_o.shape[0]
    assert n_frames >= 1 and n % n_frames == 0, "rays must hold n_frames equal frames"
    assert frame_times.numel() == n_frames, "one time per frame"
    m, res = binaries.shape[0], binaries.shape[1]
    assert aabbs.shape == (m, 6)
    dev = rays_o.device
    L = _lib.lib()
    if exchange is None:
        need = int(L.ced_render_frames_test_workspace_bytes(n_frames, n // n_frames, m, res, float(cone_angle), int(max_samples)))
    else:
        need = int(L.ced_render_frames_test_sharded_workspace_bytes(n_frames, n // n_frames, int(exchange.struct.global_rays_per_frame),
                                                                    m, res, float(cone_angle), int(max_samples)))
    if need < 0:
        raise ValueError("render_frames_test: unsupported sizes (1..64 frames)")
    host_words = 512 if exchange is None else max(512, exchange.host_words)
    key = (dev.index, torch.cuda.current_stream().cuda_stream)
    with _frame_ws_lock:
        ws = _frame_ws.get(key)
        if ws is None or ws[0].numel() < need or ws[1].numel() < host_words:
            ws = (ws[0] if ws is not None and ws[0].numel() >= need else torch.empty((max(need, 1),), device=dev, dtype=torch.uint8),
                  ws[1] if ws is not None and ws[1].numel() >= host_words else torch.zeros((host_words,), dtype=torch.int64).pin_memory())
            _frame_ws[key] = ws
    rgb = torch.empty((n, 3), device=dev, dtype=torch.float32)
    opacity = torch.empty((n, 1), device=dev, dtype=torch.float32)
    depth = torch.empty((n, 1), device=dev, dtype=torch.float32)
    totals = (C.c_int64 * n_frames)()
    args = (C.byref(_with_workgroups(desc, max_workgroups)), n_frames, n // n_frames, _p(rays_o),
            _p(rays_d), _p(_as_u8(binaries)), m, res, _p(aabbs), _p(accel), float(near_plane), float(far_plane), float(render_step_size),
            float(cone_angle), float(early_stop_eps), int(max_samples), _p(frame_times), _p(bkgd),
            _p(rgb), _p(opacity), _p(depth), _p(ws[0]), ws[0].numel(), C.c_void_p(ws[1].data_ptr()),
            totals, C.byref(tracer.struct) if tracer is not None else None,
            C.c_void_p(field_stream.cuda_stream) if field_stream is not None else None, _stream())
    if exchange is None:
        rc = L.ced_render_frames_test(*args)
    else:
        assert exchange.n_frames == n_frames and exchange.counts.device == dev
        exchange.error = None
        rc = L.ced_render_frames_test_sharded(*args, C.byref(exchange.struct))
        if exchange.error is not None:
            raise exchange.error
    _lib.check(rc, "render_frames_test")
    return rgb, opacity, depth, [int(v) for v in totals]


# ----------------------------------------------------------------------------------------------
# occupancy-grid maintenance
# ----------------------------------------------------------------------------------------------
def occ_cell_points(cell_indices: torch.Tensor, noise: torch.Tensor, res: int, aabb) -> torch.Tensor:
    """Jittered sample position of every listed cell of one grid level (ced_occ_cell_points)."""
    _chk(cell_indices, torch.int64, "cell_indices"); _chk(noise, torch.float32, "noise")
    n = cell_indices.shape[0]
    assert noise.shape == (n, 3)
    pos = torch.empty((n, 3), device=cell_indices.device, dtype=torch.float32)
    ab = (C.c_float * 6)(*[float(v) for v in aabb])
    rc = _lib.lib().ced_occ_cell_points(n, _p(cell_indices), _p(noise), int(res), ab, _p(pos), _stream())
    _lib.check(rc, "occ_cell_points")
    return pos


def occ_ema_update_(occs: torch.Tensor, cell_ids: torch.Tensor, density: torch.Tensor, step_size: float,
                    ema_decay: float) -> None:
    """occs[cell_ids] = max(occs[cell_ids] * ema_decay, density * step_size), in place."""
    _chk(occs, torch.float32, "occs"); _chk(cell_ids, torch.int64, "cell_ids"); _chk(density, torch.float32, "density")
    assert density.shape == cell_ids.shape
    rc = _lib.lib().ced_occ_ema_update(cell_ids.shape[0], _p(cell_ids), _p(density), float(step_size), float(ema_decay),
                                       _p(occs), _stream())
    _lib.check(rc, "occ_ema_update")
