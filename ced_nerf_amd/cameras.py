"""Full-frame ray generation on the device.

The reference builds `Rays(origins, viewdirs)` on the host for every frame and uploads them
(datasets/dnerf_synthetic.py:191-242, gui.py:43-86, datasets/hypernerf.py:169-176); here the
camera parameters go to a HIP kernel that writes the [H,W,3] tensors directly.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from .utils import Rays


def _farr(values, n):
    a = np.asarray(values, np.float32).reshape(-1)
    if a.shape[0] != n:
        raise ValueError(f"expected {n} values, got {a.shape[0]}")
    return (C.c_float * n)(*[float(v) for v in a])


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pinhole_rays(K, c2w, width: int, height: int, opengl: bool = True, device="cuda") -> Rays:
    """Rays of a pinhole camera: K [3,3] intrinsics, c2w [3,4] (or [4,4]) camera-to-world."""
    K = np.asarray(K, np.float32)
    c2w = np.asarray(c2w, np.float32)[:3, :4]
    dev = torch.device(device)
    if dev.type != "cuda":
        raise NotImplementedError("Only support cuda devices (no CPU fallback).")
    o = torch.empty((height, width, 3), device=dev, dtype=torch.float32)
    d = torch.empty_like(o)
    with torch.cuda.device(dev):
        rc = _lib.lib().ced_generate_rays_pinhole(width, height, float(K[0, 0]), float(K[1, 1]), float(K[0, 2]),
                                                  float(K[1, 2]), _farr(c2w, 12), int(bool(opengl)),
                                                  C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()), None, _stream())
    _lib.check(rc, "generate_rays_pinhole")
    return Rays(origins=o, viewdirs=d)


def hypercam_rays(orientation, position, focal_length: float, principal_point: Sequence[float],
                  image_size: Sequence[int], skew: float = 0.0, pixel_aspect_ratio: float = 1.0,
                  radial_distortion: Optional[Sequence[float]] = None,
                  tangential_distortion: Optional[Sequence[float]] = None, device="cuda") -> Rays:
    """Rays through the pixel centres of a HyperNeRF camera (datasets/hyper_cam.py `Camera` fields)."""
    width, height = int(image_size[0]), int(image_size[1])
    dev = torch.device(device)
    if dev.type != "cuda":
        raise NotImplementedError("Only support cuda devices (no CPU fallback).")
    o = torch.empty((height, width, 3), device=dev, dtype=torch.float32)
    d = torch.empty_like(o)
    rad = _farr(radial_distortion, 3) if radial_distortion is not None else None
    tan = _farr(tangential_distortion, 2) if tangential_distortion is not None else None
    with torch.cuda.device(dev):
        rc = _lib.lib().ced_generate_rays_hypercam(width, height, _farr(orientation, 9), _farr(position, 3),
                                                   float(focal_length), float(principal_point[0]),
                                                   float(principal_point[1]), float(skew), float(pixel_aspect_ratio),
                                                   rad, tan, C.c_void_p(o.data_ptr()), C.c_void_p(d.data_ptr()),
                                                   _stream())
    _lib.check(rc, "generate_rays_hypercam")
    return Rays(origins=o, viewdirs=d)
