"""Second, independent CPU restatement in plain PyTorch fp32 (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Vectorised torch ops (torch.sin / torch.exp / matmul / cumsum) instead of the C oracle's scalar
polynomial kernels and fused-multiply-add chains, so it shares no arithmetic code with either the
C oracle or the HIP kernels.  It pins the C oracle to ~1e-5 (tests/test_oracle_cpu.py) and is the
"pure-PyTorch path" BASELINE.md asks to time on the host cores.  Marching is taken from the C
oracle (in the reference, too, marching is a native op and only the field / compositing glue is
PyTorch).  Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import oracle as O

PRIME_Y, PRIME_Z = 2654435761, 805459861


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


class TorchField:
    """DNGPradianceField.forward, cednerf/model.py:354-488, with torch ops."""

    def __init__(self, params: Dict):
        self.p = params
        h = params["hash"]
        self.lv = O.hash_levels(h["base_res"], h["max_res"], h["n_levels"], h["log2_hashmap_size"])
        self.table = _t(h["table"]).float()
        self.temporal = bool(h.get("temporal", False))
        self.aabb = _t(params["aabb"])
        self.m = [_t(w) for w in params["xyz_wrap"]]
        self.b = [_t(w) for w in params["mlp_base"]]
        self.h = [_t(w) for w in params["mlp_head"]]

    # hash_encoder_half.py:112-161 / hash_encoder_inter.py:121-199
    def hash_encode(self, x: torch.Tensor, t: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = x.clamp(0.0, 1.0)
        outs = []
        n = x.shape[0]
        for l in range(self.lv["n_levels"]):
            scale = float(self.lv["scale"][l]); res = int(self.lv["res"][l])
            off = int(self.lv["offset"][l]); size = int(self.lv["size"][l]); hashed = bool(self.lv["hashed"][l])
            pos = x * scale + 0.5
            g = torch.floor(pos)
            fr = pos - g
            g = g.to(torch.int64)
            acc = torch.zeros((n, 2), dtype=torch.float32)
            for c in range(8):
                w = torch.ones((n,), dtype=torch.float32)
                pc = []
                for a in range(3):
                    if (c >> a) & 1:
                        pc.append(g[:, a] + 1); w = w * fr[:, a]
                    else:
                        pc.append(g[:, a]); w = w * (1.0 - fr[:, a])
                if hashed:
                    idx = (pc[0] & 0xFFFFFFFF) ^ ((pc[1] * PRIME_Y) & 0xFFFFFFFF) ^ ((pc[2] * PRIME_Z) & 0xFFFFFFFF)
                else:
                    idx = (pc[0] + pc[1] * res + pc[2] * res * res) & 0xFFFFFFFF
                e = off + idx % size
                if not self.temporal:
                    f = self.table[e]
                else:
                    ts = t.reshape(-1) * 3.0
                    lo = torch.floor(ts)
                    tf = (ts - lo)[:, None]
                    lo = lo.clamp(max=2.0).to(torch.int64)
                    ent = self.table[e]                                     # [n, 8]
                    k = (2 * lo)[:, None] + torch.arange(2)[None, :]
                    f = torch.gather(ent, 1, k) * (1.0 - tf) + torch.gather(ent, 1, k + 2) * tf
                acc = acc + w[:, None] * f
            outs.append(acc)
        return torch.cat(outs, -1)

    @staticmethod
    def time_encode(t: torch.Tensor, move_norm: Optional[torch.Tensor], with_exp: bool) -> torch.Tensor:
        """cednerf/encoder.py:6-44 / :46-90 written out per component."""
        t = t.reshape(-1, 1)
        cols = [t]
        if not with_exp:
            for shift in (0.0, 0.5 * math.pi):
                for k in range(4):
                    cols.append(torch.sin(t * (2 ** k) + shift))
        else:
            mv = move_norm.reshape(-1, 1)
            for k in range(4):
                att = torch.exp(-1.0 * (mv * (k * 2 ** k)))
                cols.append(torch.sin(t * (2 ** k)) * att)
                cols.append(torch.sin(t * (2 ** k) + 0.5 * math.pi) * att)
        return torch.cat(cols, -1)

    def forward(self, pos, t, dirs=None):
        pos = _t(pos); t = _t(t).reshape(-1, 1)
        x4 = torch.cat([pos, t], -1)
        enc = []
        for d in range(4):                      # tcnn Frequency, SURVEY A.7: [dim][freq][sin, cos]
            for k in range(4):
                ang = (2 ** k) * math.pi * x4[:, d]
                enc += [torch.sin(ang), torch.sin(ang + 0.5 * math.pi)]
        hcur = torch.stack(enc, -1)
        for i, w in enumerate(self.m):
            hcur = hcur @ w.T
            if i < len(self.m) - 1:
                hcur = torch.relu(hcur)
        step = float(np.float32(self.p["moving_step"]))
        if self.p["use_div_offsets"]:
            move = hcur[:, :3] * step + torch.tanh(hcur[:, 3:]) * step
        else:
            move = hcur * step
        xm = pos + move
        xn = (xm - self.aabb[:3]) / (self.aabb[3:] - self.aabb[:3])
        sel = ((xn > 0.0) & (xn < 1.0)).all(-1)
        feat = self.hash_encode(xn, t)
        if self.p["time_mode"]:
            mn = torch.linalg.norm(move, dim=-1)
            feat = torch.cat([feat, self.time_encode(t, mn, self.p["time_mode"] == 2)], -1)
        hb = torch.relu(feat @ self.b[0].T) @ self.b[1].T
        sigma = torch.exp(hb[:, 0] - 1.0) * sel
        out = {"density": sigma.numpy(), "base_mlp_out": hb[:, 1:].numpy(), "x_norm": xn.numpy()}
        if dirs is not None:
            d = _t(dirs)
            d = d / torch.linalg.norm(d, dim=-1, keepdim=True)
            v = ((d + 1.0) / 2.0) * 2.0 - 1.0
            sh = torch.stack([torch.full_like(v[:, 0], 0.28209479177387814), -0.48860251190291987 * v[:, 1],
                              0.48860251190291987 * v[:, 2], -0.48860251190291987 * v[:, 0]], -1)
            hh = torch.cat([sh, hb[:, 1:]], -1)
            hh = torch.relu(hh @ self.h[0].T)
            hh = torch.relu(hh @ self.h[1].T)
            out["rgb"] = torch.sigmoid(hh @ self.h[2].T).numpy()
        return out

    def forward_rays(self, rays_o, rays_d, ray_indices, t_starts, t_ends, timestamps, want_rgb=True):
        o = _t(rays_o)[ray_indices]; d = _t(rays_d)[ray_indices]
        pos = o + d * (_t(t_starts) + _t(t_ends))[:, None] / 2.0
        t = np.broadcast_to(np.asarray(timestamps, np.float32).reshape(-1)[:1], (pos.shape[0],)).copy()
        out = self.forward(pos.numpy(), t, d.numpy() if want_rgb else None)
        return out.get("rgb"), out["density"]


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info, prefix_trans=None):
    """SURVEY A.5 with torch.cumsum per ray (different summation order from the C oracle)."""
    t0, t1, sg = _t(t_starts), _t(t_ends), _t(sigmas)
    sd = sg * (t1 - t0)
    alphas = 1.0 - torch.exp(-sd)
    csum = torch.cumsum(sd.double(), 0)
    excl = csum - sd.double()
    starts = torch.from_numpy(np.ascontiguousarray(packed_info[:, 0]))
    counts = torch.from_numpy(np.ascontiguousarray(packed_info[:, 1]))
    ray_of = torch.repeat_interleave(torch.arange(packed_info.shape[0]), counts)
    base = excl[starts.clamp(max=max(sd.shape[0] - 1, 0))] if sd.shape[0] else excl
    trans = torch.exp(-(excl - base[ray_of])).float()
    if prefix_trans is not None:
        trans = trans * _t(prefix_trans)
    return (trans * alphas).numpy(), trans.numpy(), alphas.numpy()


def accumulate_along_rays(weights, values, packed_info):
    w = _t(weights)
    src = w[:, None] * _t(values) if values is not None else w[:, None]
    counts = torch.from_numpy(np.ascontiguousarray(packed_info[:, 1]))
    ray_of = torch.repeat_interleave(torch.arange(packed_info.shape[0]), counts)
    out = torch.zeros((packed_info.shape[0], src.shape[1]), dtype=torch.float32)
    out.index_add_(0, ray_of, src)
    return out.numpy()


def render_image_test(max_samples, field: TorchField, est: O.OracleEstimator, rays_o, rays_d, near_plane=0.0,
                      far_plane=1e10, render_step_size=1e-3, render_bkgd=None, cone_angle=0.0, alpha_thre=0.0,
                      early_stop_eps=1e-4, timestamps=None):
    """cednerf/utils.py:153-318 with the PyTorch field / compositing (marching from the C oracle)."""
    shape = rays_o.shape
    o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3); d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    n = o.shape[0]
    opacity = np.zeros((n, 1), np.float32); depth = np.zeros((n, 1), np.float32); rgb = np.zeros((n, 3), np.float32)
    ray_mask = np.ones((n,), bool)
    min_samples = 1 if cone_angle == 0 else 4
    iter_samples = total = 0
    near = np.full((n,), near_plane, np.float32); far = np.full((n,), far_plane, np.float32)
    t_mins, t_maxs, hits = O.ray_aabb_intersect(o, d, est.aabbs)
    t_sorted, t_indices = O.sort_intersections(t_mins, t_maxs)
    thr = np.float32(1 - early_stop_eps)
    while iter_samples < max_samples:
        n_alive = int(ray_mask.sum())
        if n_alive == 0:
            break
        n_samples = max(min(n // n_alive, 64), min_samples)
        iter_samples += n_samples
        tr = O.traverse_grids(o, d, est.binaries, est.aabbs, near, far, render_step_size, cone_angle, n_samples, True,
                              ray_mask, t_sorted, t_indices, hits)
        t0, t1, ri, packed = tr["t_starts"], tr["t_ends"], tr["ray_indices"], tr["packed_compact"]
        if ri.shape[0]:
            rgbs, sig = field.forward_rays(o, d, ri, t0, t1, timestamps)
            w, _, _ = render_weight_from_density(t0, t1, sig, packed, (1.0 - opacity[ri, 0]).astype(np.float32))
            rgb += accumulate_along_rays(w, rgbs, packed)
            opacity += accumulate_along_rays(w, None, packed)
            depth += accumulate_along_rays(w, ((t0 + t1) / 2.0)[:, None], packed)
        near = tr["termination_planes"]
        ray_mask = np.logical_and(opacity.reshape(-1) <= thr, tr["packed_info"][:, 1] == n_samples)
        total += ri.shape[0]
    bk = np.zeros(3, np.float32) if render_bkgd is None else np.asarray(render_bkgd, np.float32)
    rgb = rgb + bk * (1.0 - opacity)
    depth = depth / np.maximum(opacity, np.finfo(np.float32).eps)
    s = tuple(shape[:-1])
    return rgb.reshape(s + (3,)), opacity.reshape(s + (1,)), depth.reshape(s + (1,)), total
