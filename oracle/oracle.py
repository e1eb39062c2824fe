"""CPU ORACLE driver (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

numpy front-end of ``cednerf_oracle.c`` plus the host control flow of the reference's render
drivers.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Parity status: see the header of ``cednerf_oracle.c`` ("parity unpinned" at
the nerfacc / tiny-cuda-nn boundaries; pinned by golden vectors of ``cednerf/encoder.py``, the
analytic known-answer tests and ``torch_oracle.py``).

Citations are relative to /root/reference.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess
from typing import Dict, List, Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libcednerf_oracle.so")
MAX_LEVELS = 16


def build(force: bool = False) -> str:
    """Compile the C oracle with the committed Makefile (gcc)."""
    srcs = [os.path.join(_HERE, f) for f in ("cednerf_oracle.c", "mfma_f16_model.h", "Makefile")]
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


class _HashT(C.Structure):
    _fields_ = [
        ("n_levels", C.c_int32), ("table_dtype", C.c_int32), ("temporal", C.c_int32), ("pad_", C.c_int32),
        ("scale", C.c_float * MAX_LEVELS), ("res", C.c_uint32 * MAX_LEVELS),
        ("offset", C.c_uint32 * MAX_LEVELS), ("size", C.c_uint32 * MAX_LEVELS),
        ("hashed", C.c_uint32 * MAX_LEVELS), ("table", C.c_void_p),
    ]


class _FieldT(C.Structure):
    _fields_ = [
        ("aabb", C.c_float * 6), ("moving_step", C.c_float), ("use_div_offsets", C.c_int32),
        ("time_mode", C.c_int32), ("base_in", C.c_int32),
        ("m_w0", C.c_void_p), ("m_w1", C.c_void_p), ("m_w2", C.c_void_p), ("m_w3", C.c_void_p),
        ("b_w0", C.c_void_p), ("b_w1", C.c_void_p),
        ("h_w0", C.c_void_p), ("h_w1", C.c_void_p), ("h_w2", C.c_void_p),
        ("hash", _HashT), ("mlp_half", C.c_int32), ("reserved", C.c_int32), ("half_cache", C.c_void_p),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.ced_o_expf.restype = C.c_float
        _lib.ced_o_expf.argtypes = [C.c_float]
        _lib.ced_o_sinf.restype = C.c_float
        _lib.ced_o_sinf.argtypes = [C.c_float]
        _lib.ced_o_sinpi_phase.restype = C.c_float
        _lib.ced_o_sinpi_phase.argtypes = [C.c_float, C.c_int]
        _lib.ced_o_skip_march.restype = C.c_float
        _lib.ced_o_skip_march.argtypes = [C.c_float] * 4
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def round_f16(x: np.ndarray) -> np.ndarray:
    """The C oracle's fp32 -> fp16-grid rounding (ties to even, saturating at +-65504)."""
    x = _f32(x)
    y = np.empty_like(x)
    lib().ced_o_round_f16(C.c_int64(x.size), _p(x), _p(y))
    return y


# ---------------------------------------------------------------------------------------------
# hash-grid level geometry (host, float64): hash_encoder_half.py:12-35 (align_to, res/scale),
# :268-292 (offsets, sizes, first hashed level); SURVEY A.6.
# ---------------------------------------------------------------------------------------------
def hash_levels(base_res: int = 16, max_res: int = 1024, n_levels: int = 16, log2_hashmap_size: int = 21) -> Dict:
    log_b = np.log(float(max_res) / float(base_res)) / float(n_levels - 1)
    T = 2 ** log2_hashmap_size
    scale, res, offset, size, hashed = [], [], [], [], []
    off = 0
    for l in range(n_levels):
        s = float(base_res) * np.exp(float(l) * log_b) - 1.0
        if abs(s - round(s)) < 1e-9:          # exact-integer scales (e.g. 1023) are snapped
            s = float(round(s))
        r = int(np.ceil(s)) + 1
        full = r ** 3
        full_aligned = ((full + 7) // 8) * 8
        sz = min(T, full_aligned)
        scale.append(np.float32(s)); res.append(r); offset.append(off); size.append(sz)
        hashed.append(1 if full > sz else 0)
        off += sz
    return dict(n_levels=n_levels, scale=np.array(scale, np.float32), res=np.array(res, np.uint32),
                offset=np.array(offset, np.uint32), size=np.array(size, np.uint32),
                hashed=np.array(hashed, np.uint32), total=off)


class OracleField:
    """DNGPradianceField (cednerf/model.py:97-488) evaluated by the C oracle."""

    MLP_MODES = {"f32": 0, "f16": 1, "f16x2": 2, "f32+h16x2": 3}

    def __init__(self, params: Dict, mlp_half=False, prepare: bool = True):
        """mlp_half: MLP arithmetic -- False / "f32": fp32 fmaf chains; True / "f16": the fp16-operand / fp32-accumulate
        class (tcnn FullyFusedMLP) as gfx950's matrix instruction computes it (mfma_f16_model.h);
        "f16x2": operands split into two fp16 numbers; "f32+h16x2": only mlp_head split.  Weights are passed in fp32
        and rounded / split inside the C code."""
        self.p = params
        h = params["hash"]
        self.levels = hash_levels(h["base_res"], h["max_res"], h["n_levels"], h["log2_hashmap_size"])
        table = np.ascontiguousarray(h["table"])
        assert table.dtype in (np.float32, np.float16)
        width = 8 if h.get("temporal", False) else 2
        assert table.shape == (self.levels["total"], width), (table.shape, self.levels["total"], width)
        self._keep = [table]
        ht = _HashT()
        ht.n_levels = h["n_levels"]
        ht.table_dtype = 0 if table.dtype == np.float32 else 1
        ht.temporal = 1 if h.get("temporal", False) else 0
        for l in range(h["n_levels"]):
            ht.scale[l] = float(self.levels["scale"][l])
            ht.res[l] = int(self.levels["res"][l])
            ht.offset[l] = int(self.levels["offset"][l])
            ht.size[l] = int(self.levels["size"][l])
            ht.hashed[l] = int(self.levels["hashed"][l])
        ht.table = table.ctypes.data
        self.hash_t = ht
        if "xyz_wrap" not in params:
            self.field_t = None
            return
        ft = _FieldT()
        aabb = _f32(params["aabb"])
        for i in range(6):
            ft.aabb[i] = float(aabb[i])
        ft.moving_step = float(np.float32(params["moving_step"]))
        ft.use_div_offsets = int(bool(params["use_div_offsets"]))
        ft.time_mode = int(params["time_mode"])
        ft.base_in = 41 if ft.time_mode else 32
        rw = _f32
        ft.mlp_half = self.MLP_MODES[mlp_half] if isinstance(mlp_half, str) else int(mlp_half)
        assert ft.mlp_half in (0, 1, 2, 3), mlp_half
        m = [rw(w) for w in params["xyz_wrap"]]
        b = [rw(w) for w in params["mlp_base"]]
        hd = [rw(w) for w in params["mlp_head"]]
        assert m[0].shape == (64, 32) and m[1].shape == (64, 64) and m[2].shape == (64, 64)
        assert m[3].shape == (6 if ft.use_div_offsets else 3, 64)
        assert b[0].shape == (64, ft.base_in) and b[1].shape == (16, 64)
        assert hd[0].shape == (64, 19) and hd[1].shape == (64, 64) and hd[2].shape == (3, 64)
        self._keep += m + b + hd
        ft.m_w0, ft.m_w1, ft.m_w2, ft.m_w3 = [w.ctypes.data for w in m]
        ft.b_w0, ft.b_w1 = [w.ctypes.data for w in b]
        ft.h_w0, ft.h_w1, ft.h_w2 = [w.ctypes.data for w in hd]
        ft.hash = ht
        ft.half_cache = None
        self.field_t = ft
        if ft.mlp_half and prepare:
            lib().ced_o_field_prepare(C.byref(ft))

    def __del__(self):
        ft = getattr(self, "field_t", None)
        if ft is not None and ft.half_cache:
            try:
                lib().ced_o_field_release(C.byref(ft))
            except Exception:
                pass

    # hash_encoder(x) -- model.py:384
    def hash_encode(self, x: np.ndarray, t: Optional[np.ndarray] = None) -> np.ndarray:
        x = _f32(x)
        n = x.shape[0]
        out = np.empty((n, 2 * self.p["hash"]["n_levels"]), np.float32)
        tt = _f32(t).reshape(-1) if t is not None else None
        lib().ced_o_hash_encode(C.byref(self.hash_t), C.c_int64(n), _p(x), _p(tt), _p(out))
        return out

    def hash_encode_backward(self, x: np.ndarray, dy: np.ndarray, want_dx: bool = True, dx_scaled: bool = False):
        """hash_encoder_backward_kernel (hash_encoder_half.py:164-226): (grad_table [E,2] float64, dx [n,3])."""
        x = _f32(x); dy = _f32(dy).reshape(x.shape[0], -1)
        n = x.shape[0]
        grad = np.zeros((self.levels["total"], 2), np.float64)
        dx = np.empty((n, 3), np.float32) if want_dx else None
        lib().ced_o_hash_encode_backward(C.byref(self.hash_t), C.c_int64(n), _p(x), _p(dy), _p(grad), _p(dx),
                                         C.c_int(int(dx_scaled)))
        return grad, dx

    def hash_encode_backward_temporal(self, x: np.ndarray, t: np.ndarray, dy: np.ndarray) -> np.ndarray:
        """hash_encoder_backward_kernel of the temporal table (hash_encoder_inter.py:202-275): grad_table [E,8] float64."""
        x = _f32(x); dy = _f32(dy).reshape(x.shape[0], -1); t = _f32(t).reshape(-1)
        assert self.p["hash"].get("temporal", False)
        grad = np.zeros((self.levels["total"], 8), np.float64)
        lib().ced_o_hash_encode_backward_temporal(C.byref(self.hash_t), C.c_int64(x.shape[0]), _p(x), _p(t), _p(dy), _p(grad))
        return grad

    def hash_indices(self, x: np.ndarray) -> np.ndarray:
        x = _f32(x)
        n = x.shape[0]
        out = np.empty((n, self.p["hash"]["n_levels"], 8), np.uint32)
        lib().ced_o_hash_indices(C.byref(self.hash_t), C.c_int64(n), _p(x), _p(out))
        return out

    # forward(positions, t, directions) -- model.py:468-488
    def forward(self, pos, t, dirs=None, want_geo=False, want_xnorm=False):
        pos = _f32(pos); t = _f32(t).reshape(-1)
        n = pos.shape[0]
        d = _f32(dirs) if dirs is not None else None
        rgb = np.empty((n, 3), np.float32) if d is not None else None
        sigma = np.empty((n,), np.float32)
        geo = np.empty((n, 15), np.float32) if want_geo else None
        xn = np.empty((n, 3), np.float32) if want_xnorm else None
        lib().ced_o_field_forward(C.byref(self.field_t), C.c_int64(n), _p(pos), _p(t), _p(d), _p(rgb),
                                  _p(sigma), _p(geo), _p(xn))
        out = {"rgb": rgb, "density": sigma}
        if want_geo:
            out["base_mlp_out"] = geo
        if want_xnorm:
            out["x_norm"] = xn
        return out

    # the sigma_fn / rgb_sigma_fn closures -- utils.py:74-104,181-195
    def forward_rays(self, rays_o, rays_d, ray_indices, t_starts, t_ends, timestamps, t_per_ray=False,
                     want_rgb=True):
        n = ray_indices.shape[0]
        rgb = np.empty((n, 3), np.float32) if want_rgb else None
        sigma = np.empty((n,), np.float32)
        ts = _f32(timestamps).reshape(-1)
        lib().ced_o_field_forward_rays(C.byref(self.field_t), C.c_int64(n), _p(rays_o), _p(rays_d),
                                       _p(np.ascontiguousarray(ray_indices, dtype=np.int64)), _p(_f32(t_starts)),
                                       _p(_f32(t_ends)), _p(ts), C.c_int(int(t_per_ray)), C.c_int(int(want_rgb)),
                                       _p(rgb), _p(sigma), None)
        return rgb, sigma


# ---------------------------------------------------------------------------------------------
# nerfacc restatement, numpy in / numpy out
# ---------------------------------------------------------------------------------------------
def enlarge_aabb(aabb, factor):
    aabb = np.asarray(aabb, np.float32)
    c = (aabb[:3] + aabb[3:]) / np.float32(2)
    e = (aabb[3:] - aabb[:3]) / np.float32(2)
    return np.concatenate([c - e * np.float32(factor), c + e * np.float32(factor)]).astype(np.float32)


def make_aabbs(roi_aabb, levels):
    return np.stack([enlarge_aabb(roi_aabb, 2 ** i) for i in range(levels)], 0)


def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane=-np.inf, far_plane=np.inf, miss_value=np.inf):
    rays_o = _f32(rays_o); rays_d = _f32(rays_d); aabbs = _f32(aabbs)
    n, m = rays_o.shape[0], aabbs.shape[0]
    t_mins = np.empty((n, m), np.float32); t_maxs = np.empty((n, m), np.float32)
    hits = np.empty((n, m), np.uint8)
    lib().ced_o_ray_aabb_intersect(C.c_int64(n), _p(rays_o), _p(rays_d), C.c_int(m), _p(aabbs),
                                   C.c_float(near_plane), C.c_float(far_plane), C.c_float(miss_value),
                                   _p(t_mins), _p(t_maxs), _p(hits))
    return t_mins, t_maxs, hits.astype(bool)


def sort_intersections(t_mins, t_maxs):
    """utils.py:219-225: sorted event list of the per-level entry/exit distances."""
    n, m = t_mins.shape
    cat = np.concatenate([t_mins, t_maxs], -1)
    if m > 1:
        t_indices = np.argsort(cat, axis=-1, kind="stable").astype(np.int64)
        t_sorted = np.take_along_axis(cat, t_indices, -1)
    else:
        t_sorted = cat
        t_indices = np.broadcast_to(np.arange(2 * m, dtype=np.int64), (n, 2 * m)).copy()
    return np.ascontiguousarray(t_sorted), np.ascontiguousarray(t_indices)


def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle,
                   traverse_steps_limit=0, over_allocate=False, rays_mask=None,
                   t_sorted=None, t_indices=None, hits=None):
    """Returns dict(t_starts, t_ends, ray_indices, packed_info[N,2], termination_planes).
    With over_allocate the arrays hold only the valid samples (the caller-side compaction of
    utils.py:265-267 is applied) and packed_info is (r*limit, n) as nerfacc reports it."""
    rays_o = _f32(rays_o); rays_d = _f32(rays_d); aabbs = _f32(aabbs)
    binaries = np.ascontiguousarray(binaries).astype(np.uint8)
    n = rays_o.shape[0]
    m, res = binaries.shape[0], binaries.shape[1]
    assert binaries.shape[1] == binaries.shape[2] == binaries.shape[3]
    near_planes = _f32(near_planes); far_planes = _f32(far_planes)
    if t_sorted is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    t_sorted = _f32(t_sorted); t_indices = np.ascontiguousarray(t_indices, np.int64)
    hits_u8 = np.ascontiguousarray(hits).astype(np.uint8)
    mask_u8 = None if rays_mask is None else np.ascontiguousarray(rays_mask).astype(np.uint8)
    counts = np.zeros((n,), np.int64)
    term = np.empty((n,), np.float32)
    L = lib()
    common = (C.c_int64(n), _p(rays_o), _p(rays_d), _p(binaries), C.c_int(m), C.c_int(res), _p(aabbs),
              _p(near_planes), _p(far_planes), C.c_float(step_size), C.c_float(cone_angle),
              C.c_int(int(traverse_steps_limit)), _p(mask_u8), _p(t_sorted), _p(t_indices), _p(hits_u8))
    L.ced_o_traverse_grids(*common, C.c_int(0), None, _p(counts), None, None, _p(term))
    base = np.zeros((n,), np.int64)
    base[1:] = np.cumsum(counts)[:-1]
    total = int(counts.sum())
    t_starts = np.empty((total,), np.float32); t_ends = np.empty((total,), np.float32)
    counts2 = np.zeros((n,), np.int64)
    L.ced_o_traverse_grids(*common, C.c_int(1), _p(base), _p(counts2), _p(t_starts), _p(t_ends), _p(term))
    assert np.array_equal(counts, counts2)
    ray_indices = np.repeat(np.arange(n, dtype=np.int64), counts)
    if over_allocate:
        packed = np.stack([np.arange(n, dtype=np.int64) * int(traverse_steps_limit), counts], -1)
    else:
        packed = np.stack([base, counts], -1)
    return dict(t_starts=t_starts, t_ends=t_ends, ray_indices=ray_indices, packed_info=packed,
                packed_compact=np.stack([base, counts], -1), termination_planes=term)


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans=None):
    n_rays = packed_info.shape[0]
    s = t_starts.shape[0]
    w = np.empty((s,), np.float32); tr = np.empty((s,), np.float32); al = np.empty((s,), np.float32)
    pt = _f32(prefix_trans) if prefix_trans is not None else None
    lib().ced_o_render_weights(C.c_int64(n_rays), _p(np.ascontiguousarray(packed_info, np.int64)), _p(_f32(t_starts)),
                               _p(_f32(t_ends)), _p(_f32(sigmas)), _p(pt), _p(w), _p(tr), _p(al))
    return w, tr, al


def accumulate_along_rays_(weights, values, packed_info, outputs):
    """In place: outputs[ray] += w*v (values None -> w)."""
    n_rays = packed_info.shape[0]
    Cn = outputs.shape[1]
    v = _f32(values) if values is not None else None
    assert outputs.dtype == np.float32 and outputs.flags.c_contiguous
    lib().ced_o_accumulate(C.c_int64(n_rays), _p(np.ascontiguousarray(packed_info, np.int64)), _p(_f32(weights)),
                           _p(v), C.c_int(Cn), _p(outputs))
    return outputs


def composite_backward(packed_info, t_starts, t_ends, sigmas, rgbs, d_color, d_opacity, d_depth):
    """Derivative of (colors, opacities, depths) of cednerf/render.py:158-169 w.r.t. (sigmas, rgbs), in float64."""
    n_rays = packed_info.shape[0]
    S = sigmas.shape[0]
    ds = np.zeros((S,), np.float64); dc = np.zeros((S, 3), np.float64)
    lib().ced_o_composite_backward(C.c_int64(n_rays), _p(np.ascontiguousarray(packed_info, dtype=np.int64)), _p(_f32(t_starts)),
                                   _p(_f32(t_ends)), _p(_f32(sigmas)), _p(_f32(rgbs)), _p(_f32(d_color)),
                                   _p(_f32(d_opacity).reshape(-1)), _p(_f32(d_depth).reshape(-1)), _p(ds), _p(dc))
    return ds, dc


def frame_to_rgb8(rgb, flip_w=True):
    """train_real.py:556: np.flip(rgb * 255, axis=1).astype(np.uint8) (float32 product, truncation)."""
    v = _f32(rgb) * np.float32(255.0)
    return (np.flip(v, axis=1) if flip_w else v).astype(np.uint8)


def depth_to_u8(depth, flip_w=True):
    """depth2img (train_real.py:38-41) up to the colour-map lookup: min-max normalise, x 255, uint8; flipped as the
    video frames are (train_real.py:557)."""
    d = _f32(depth)
    d = (d - d.min()) / (d.max() - d.min())
    v = d * np.float32(255.0)
    return (np.flip(v, axis=1) if flip_w else v).astype(np.uint8)


def weight_grad(x, dy):
    """dW [n_out, n_in] = dy^T x in float64 (checker of ced_weight_grad)."""
    x = _f32(x); dy = _f32(dy)
    dw = np.zeros((dy.shape[1], x.shape[1]), np.float64)
    lib().ced_o_weight_grad(C.c_int64(x.shape[0]), _p(x), C.c_int32(x.shape[1]), _p(dy), C.c_int32(dy.shape[1]), _p(dw))
    return dw


def visibility_mask(t_starts, t_ends, sigmas, packed_info, early_stop_eps, alpha_thre):
    s = t_starts.shape[0]
    mask = np.empty((s,), np.uint8)
    lib().ced_o_visibility(C.c_int64(packed_info.shape[0]), _p(np.ascontiguousarray(packed_info, np.int64)),
                           _p(_f32(t_starts)), _p(_f32(t_ends)), _p(_f32(sigmas)), C.c_float(early_stop_eps),
                           C.c_float(alpha_thre), _p(mask))
    return mask.astype(bool)


def _packed_from_indices(ray_indices, n_rays):
    counts = np.bincount(ray_indices, minlength=n_rays).astype(np.int64)
    base = np.zeros((n_rays,), np.int64)
    base[1:] = np.cumsum(counts)[:-1]
    return np.stack([base, counts], -1)


class OracleEstimator:
    """OccGridEstimator state: binaries [m,R,R,R] bool, aabbs [m,6], occs [m*R^3]."""

    def __init__(self, roi_aabb, resolution=128, levels=1, binaries=None, occs=None):
        self.aabbs = make_aabbs(roi_aabb, levels)
        self.binaries = np.zeros((levels, resolution, resolution, resolution), bool) if binaries is None else binaries
        self.occs = self.binaries.reshape(-1).astype(np.float32) if occs is None else occs

    # OccGridEstimator.sampling (utils.py:115-125), eval mode (stratified=False)
    def sampling(self, rays_o, rays_d, sigma_fn, near_plane=0.0, far_plane=1e10, render_step_size=1e-3,
                 early_stop_eps=1e-4, alpha_thre=0.0, cone_angle=0.0, near_jitter=None):
        n = rays_o.shape[0]
        near = np.full((n,), near_plane, np.float32)
        far = np.full((n,), far_plane, np.float32)
        if near_jitter is not None:           # stratified: near += U[0,1)*step, noise supplied by the caller
            near = near + _f32(near_jitter) * np.float32(render_step_size)
        tr = traverse_grids(rays_o, rays_d, self.binaries, self.aabbs, near, far, render_step_size, cone_angle)
        t0, t1, ri, packed = tr["t_starts"], tr["t_ends"], tr["ray_indices"], tr["packed_info"]
        n_marched = t0.shape[0]
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and sigma_fn is not None:
            alpha_thre = min(alpha_thre, float(self.occs.mean()))
            sig = sigma_fn(t0, t1, ri) if n_marched else np.empty((0,), np.float32)
            mask = visibility_mask(t0, t1, sig, packed, early_stop_eps, alpha_thre)
            ri, t0, t1 = ri[mask], t0[mask], t1[mask]
        return ri, t0, t1, n_marched


def occ_grid_update(occs, aabbs, res, lvl_indices, lvl_noise, occ_eval_fn, occ_thre=0.01, ema_decay=0.95):
    """nerfacc OccGridEstimator._update with the random draws supplied:
    x = aabb_min + (coord + noise)/res * extent; occs[id] = max(occs[id]*decay, occ_eval_fn(x));
    binaries = occs > min(mean(occs[occs >= 0]), occ_thre).  float32 throughout; returns (occs, binaries)."""
    f32 = np.float32
    occs = occs.astype(f32).copy()
    cells = res ** 3
    for lvl, idx in enumerate(lvl_indices):
        idx = np.asarray(idx, np.int64)
        if idx.size == 0:
            continue
        coords = np.stack([idx // (res * res), (idx // res) % res, idx % res], -1).astype(f32)
        x = ((coords + lvl_noise[lvl].astype(f32)) / f32(res)).astype(f32)
        ab = aabbs[lvl].astype(f32)
        pos = (ab[:3] + x * (ab[3:] - ab[:3])).astype(f32)
        occ = occ_eval_fn(pos).astype(f32)
        ids = lvl * cells + idx
        occs[ids] = np.maximum((occs[ids] * f32(ema_decay)).astype(f32), occ)
    visible = occs[occs >= 0]
    thre = min(f32(visible.astype(np.float64).mean()) if visible.size else f32(0), f32(occ_thre))
    return occs, occs > thre


# rendering(), cednerf/render.py:58-176 (eval: no training extras)
def rendering(t_starts, t_ends, ray_indices, n_rays, rgb_sigma_fn, render_bkgd=None):
    rgbs, sigmas = rgb_sigma_fn(t_starts, t_ends, ray_indices)
    packed = _packed_from_indices(ray_indices, n_rays)
    w, tr, al = render_weight_from_density(t_starts, t_ends, sigmas, packed)
    colors = np.zeros((n_rays, 3), np.float32)
    opac = np.zeros((n_rays, 1), np.float32)
    depth = np.zeros((n_rays, 1), np.float32)
    accumulate_along_rays_(w, rgbs, packed, colors)
    accumulate_along_rays_(w, None, packed, opac)
    tmid = ((t_starts + t_ends)[:, None] / np.float32(2.0)).astype(np.float32)
    accumulate_along_rays_(w, tmid, packed, depth)
    depth = depth / np.maximum(opac, np.finfo(np.float32).eps)
    if render_bkgd is not None:
        colors = colors + _f32(render_bkgd) * (np.float32(1.0) - opac)
    extras = dict(weights=w, alphas=al, trans=tr, sigmas=sigmas, rgbs=rgbs)
    return colors.astype(np.float32), opac, depth.astype(np.float32), extras


# render_image, cednerf/utils.py:46-150 (eval mode: chunked, timestamps broadcast)
def render_image(field: OracleField, est: OracleEstimator, rays_o, rays_d, near_plane=0.0, far_plane=1e10,
                 render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, alpha_thre=0.0,
                 test_chunk_size=8192, timestamps=None):
    shape = rays_o.shape
    o = _f32(rays_o).reshape(-1, 3); d = _f32(rays_d).reshape(-1, 3)
    n = o.shape[0]
    cols, opas, deps, extras_all = [], [], [], []
    total = 0
    n_marched_total = 0
    for i in range(0, n, test_chunk_size):
        co, cd = np.ascontiguousarray(o[i:i + test_chunk_size]), np.ascontiguousarray(d[i:i + test_chunk_size])

        def sigma_fn(t0, t1, ri):
            return field.forward_rays(co, cd, ri, t0, t1, timestamps, want_rgb=False)[1]

        def rgb_sigma_fn(t0, t1, ri):
            return field.forward_rays(co, cd, ri, t0, t1, timestamps, want_rgb=True)

        ri, t0, t1, n_marched = est.sampling(co, cd, sigma_fn, near_plane, far_plane, render_step_size,
                                             alpha_thre=alpha_thre, cone_angle=cone_angle)
        n_marched_total += n_marched
        c, a, dp, ex = rendering(t0, t1, ri, co.shape[0], rgb_sigma_fn, render_bkgd)
        ex.update(ray_indices=ri, t_starts=t0, t_ends=t1)
        cols.append(c); opas.append(a); deps.append(dp); extras_all.append(ex)
        total += t0.shape[0]
    out_shape = tuple(shape[:-1])
    return (np.concatenate(cols).reshape(out_shape + (3,)), np.concatenate(opas).reshape(out_shape + (1,)),
            np.concatenate(deps).reshape(out_shape + (1,)), total, extras_all, n_marched_total)


# render_image_test, cednerf/utils.py:153-318
def render_image_test(max_samples, field: OracleField, est: OracleEstimator, rays_o, rays_d, near_plane=0.0,
                      far_plane=1e10, render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, alpha_thre=0.0,
                      early_stop_eps=1e-4, timestamps=None, trace: Optional[List] = None, alive_reduce=None,
                      n_total: Optional[int] = None, n_real: Optional[int] = None):
    """alive_reduce / n_total / n_real: the rays are ONE process's share of an image of n_total rays (the first n_real
    of them real, the rest padding that is never alive); `alive_reduce(n)` returns the number of alive rays summed over
    the processes.  N_rays and N_alive of utils.py:231-235 are then the whole image's, as when one process renders it."""
    shape = rays_o.shape
    o = _f32(rays_o).reshape(-1, 3); d = _f32(rays_d).reshape(-1, 3)
    n = o.shape[0]
    opacity = np.zeros((n, 1), np.float32); depth = np.zeros((n, 1), np.float32); rgb = np.zeros((n, 3), np.float32)
    ray_mask = np.ones((n,), bool)
    if n_real is not None:
        ray_mask[n_real:] = False
    n_image = n if n_total is None else int(n_total)
    min_samples = 1 if cone_angle == 0 else 4
    iter_samples = total_samples = 0
    near_planes = np.full((n,), near_plane, np.float32)
    far_planes = np.full((n,), far_plane, np.float32)
    t_mins, t_maxs, hits = ray_aabb_intersect(o, d, est.aabbs)
    t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    opc_thres = np.float32(1 - early_stop_eps)
    while iter_samples < max_samples:
        n_alive = int(ray_mask.sum())
        if alive_reduce is not None:
            n_alive = int(alive_reduce(n_alive))
        if n_alive == 0:
            break
        n_samples = max(min(n_image // n_alive, 64), min_samples)
        iter_samples += n_samples
        tr = traverse_grids(o, d, est.binaries, est.aabbs, near_planes, far_planes, render_step_size, cone_angle,
                            n_samples, True, ray_mask, t_sorted, t_indices, hits)
        t0, t1, ri = tr["t_starts"], tr["t_ends"], tr["ray_indices"]
        packed = tr["packed_compact"]
        rgbs, sigmas = field.forward_rays(o, d, ri, t0, t1, timestamps, want_rgb=True)
        prefix = (np.float32(1) - opacity[ri, 0]).astype(np.float32)
        w, _, _ = render_weight_from_density(t0, t1, sigmas, packed, prefix_trans=prefix)
        accumulate_along_rays_(w, rgbs, packed, rgb)
        accumulate_along_rays_(w, None, packed, opacity)
        accumulate_along_rays_(w, ((t0 + t1)[:, None] / np.float32(2.0)).astype(np.float32), packed, depth)
        near_planes = tr["termination_planes"]
        ray_mask = np.logical_and(opacity.reshape(-1) <= opc_thres, tr["packed_info"][:, 1] == n_samples)
        total_samples += ri.shape[0]
        if trace is not None:
            trace.append(dict(n_alive=n_alive, n_samples=n_samples, n_new=int(ri.shape[0]),
                              counts=tr["packed_info"][:, 1].copy()))
    bk = _f32(render_bkgd) if render_bkgd is not None else np.zeros(3, np.float32)
    rgb = rgb + bk * (np.float32(1.0) - opacity)
    depth = depth / np.maximum(opacity, np.finfo(np.float32).eps)
    out_shape = tuple(shape[:-1])
    return (rgb.astype(np.float32).reshape(out_shape + (3,)), opacity.reshape(out_shape + (1,)),
            depth.astype(np.float32).reshape(out_shape + (1,)), total_samples)


def time_encode(t, move_norm=None, with_exp=False):
    t = _f32(t).reshape(-1)
    mv = _f32(move_norm).reshape(-1) if move_norm is not None else None
    out = np.empty((t.shape[0], 9), np.float32)
    lib().ced_o_time_encode_batch(C.c_int64(t.shape[0]), _p(t), _p(mv), C.c_int(int(with_exp)), _p(out))
    return out


# ---------------------------------------------------------------------------------------------
# ray generation, float32
# ---------------------------------------------------------------------------------------------
def pinhole_rays(K, c2w, width, height, opengl=True):
    """datasets/dnerf_synthetic.py:191-221 / gui.py:43-86.  Returns (origins, viewdirs) [H,W,3]."""
    f32 = np.float32
    K = np.asarray(K, f32); c2w = np.asarray(c2w, f32)[:3, :4]
    x, y = np.meshgrid(np.arange(width, dtype=f32), np.arange(height, dtype=f32), indexing="xy")
    s = f32(-1.0 if opengl else 1.0)
    cam = np.stack([(x - K[0, 2] + f32(0.5)) / K[0, 0], (y - K[1, 2] + f32(0.5)) / K[1, 1] * s, np.full_like(x, s)], -1)
    d = ((cam[..., 0:1] * c2w[:, 0] + cam[..., 1:2] * c2w[:, 1]) + cam[..., 2:3] * c2w[:, 2]).astype(f32)
    nrm = np.sqrt((d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]).astype(f32)
    o = np.broadcast_to(c2w[:, 3], d.shape).copy()
    return o, (d / nrm[..., None]).astype(f32)


def hypercam_rays(orientation, position, focal_length, principal_point, image_size, skew=0.0,
                  pixel_aspect_ratio=1.0, radial_distortion=None, tangential_distortion=None):
    """datasets/hyper_cam.py:210-252 on the pixel centres (:299-303), Newton undistortion :22-91."""
    f32 = np.float32
    R = np.asarray(orientation, f32); pos = np.asarray(position, f32)
    W, H = int(image_size[0]), int(image_size[1])
    rad = np.zeros(3, f32) if radial_distortion is None else np.asarray(radial_distortion, f32)
    tan = np.zeros(2, f32) if tangential_distortion is None else np.asarray(tangential_distortion, f32)
    f, skew, asp = f32(focal_length), f32(skew), f32(pixel_aspect_ratio)
    px, py = np.meshgrid(np.arange(W, dtype=f32) + f32(0.5), np.arange(H, dtype=f32) + f32(0.5))
    y = (py - f32(principal_point[1])) / (f * asp)
    x = (px - f32(principal_point[0]) - y * skew) / f
    if rad.any() or tan.any():
        k1, k2, k3 = rad; p1, p2 = tan
        xd, yd = x.copy(), y.copy()
        one, two, three, six = f32(1), f32(2), f32(3), f32(6)
        for _ in range(10):
            r = x * x + y * y
            d = one + r * (k1 + r * (k2 + k3 * r))
            fx = d * x + two * p1 * x * y + p2 * (r + two * x * x) - xd
            fy = d * y + two * p2 * x * y + p1 * (r + two * y * y) - yd
            d_r = k1 + r * (two * k2 + three * k3 * r)
            d_x = two * x * d_r; d_y = two * y * d_r
            fx_x = d + d_x * x + two * p1 * y + six * p2 * x
            fx_y = d_y * x + two * p1 * x + two * p2 * y
            fy_x = d_x * y + two * p2 * y + two * p1 * x
            fy_y = d + d_y * y + two * p2 * x + six * p1 * y
            den = fy_x * fx_y - fx_x * fy_y
            ok = np.abs(den) > f32(1e-9)
            safe = np.where(ok, den, one)
            x = x + np.where(ok, (fx * fy_y - fy * fx_y) / safe, f32(0))
            y = y + np.where(ok, (fy * fx_x - fx * fy_x) / safe, f32(0))
    l = np.stack([x, y, np.ones_like(x)], -1)
    l = l / np.sqrt((l[..., 0] * l[..., 0] + l[..., 1] * l[..., 1]) + l[..., 2] * l[..., 2])[..., None]
    w = np.stack([(R[0, r] * l[..., 0] + R[1, r] * l[..., 1]) + R[2, r] * l[..., 2] for r in range(3)], -1).astype(f32)
    w = w / np.sqrt((w[..., 0] * w[..., 0] + w[..., 1] * w[..., 1]) + w[..., 2] * w[..., 2])[..., None]
    return np.broadcast_to(pos, w.shape).copy(), w.astype(f32)
