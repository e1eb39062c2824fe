"""DNGPradianceField: host-side mirror of cednerf/model.py:97-488 over the fused HIP field kernel.

Same constructor signature, attributes (`aabb`, `training`, `MOVING_STEP`, ...) and call
conventions as the reference module; the tiny-cuda-nn sub-modules are replaced by plain
parameter tensors (hash table [E,2], bias-free MLP weights W[out][in]) that are re-packed into the
kernel's LDS layout whenever they change.  Forward only (rendering hot path); training is a
"next" row of SURVEY.md section 8f.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Union

import numpy as np
import torch

from . import _lib, ops
from .encoder import SinusoidalEncoder, SinusoidalEncoderWithExp
from .hashgrid import level_tables

MOVING_STEP = 1 / (4096 * 1)   # cednerf/model.py:26


def _xavier_(w: torch.Tensor, gen: Optional[torch.Generator]) -> torch.Tensor:
    n_out, n_in = w.shape
    lim = math.sqrt(6.0 / (n_in + n_out))
    with torch.no_grad():
        w.uniform_(-lim, lim, generator=gen)
    return w


class DNGPradianceField(torch.nn.Module):
    """Instance-NGP radiance field with a motion MLP (reference: cednerf/model.py:97)."""

    def __init__(
        self,
        aabb: Union[torch.Tensor, List[float]],
        num_dim: int = 3,
        use_viewdirs: bool = True,
        density_activation=None,
        geo_feat_dim: int = 15,
        base_resolution: int = 16,
        n_levels: int = 16,
        n_features_per_level: int = 2,
        dst_resolution: int = 4096,
        log2_hashmap_size: int = 19,
        use_feat_predict: bool = False,
        use_weight_predict: bool = False,
        moving_step: float = MOVING_STEP,
        use_div_offsets: bool = False,
        use_time_embedding: bool = False,
        use_time_attenuation: bool = False,
        time_inject_before_sigma: bool = True,
        hash4motion: bool = False,
        hash_dtype: torch.dtype = torch.float32,
        temporal_hash: bool = False,
        seed: Optional[int] = None,
        mlp_precision: str = "f32",
    ) -> None:
        super().__init__()
        if not isinstance(aabb, torch.Tensor):
            aabb = torch.tensor(aabb, dtype=torch.float32)
        self.register_buffer("aabb", aabb.float())
        # fixed by the fused kernel (they are the reference's defaults and what train_real.py uses)
        if num_dim != 3 or not use_viewdirs or geo_feat_dim != 15 or n_features_per_level != 2 or n_levels != 16:
            raise NotImplementedError("the HIP field kernel implements num_dim=3, use_viewdirs=True, "
                                      "geo_feat_dim=15, n_levels=16, n_features_per_level=2")
        if density_activation is not None:
            raise NotImplementedError("density_activation is fixed to trunc_exp(x - 1) (cednerf/model.py:105)")
        if hash4motion:
            raise NotImplementedError("hash4motion is never enabled by the reference (train_real.py:263)")
        if not time_inject_before_sigma:
            raise NotImplementedError("time_inject_before_sigma=False is never used by the reference")
        # use_feat_predict / use_weight_predict (run_hyper.sh:1 passes -f): the two heads only run under
        # `return_interal=self.training` (cednerf/model.py:428-443,479-484), i.e. they are inert in the eval path
        # this module implements; the flags are accepted and recorded, training with them is `train.TrainableField`.
        self.num_dim = num_dim
        self.use_viewdirs = use_viewdirs
        self.geo_feat_dim = geo_feat_dim
        self.geo_feat_dim_head = geo_feat_dim
        self.box_scale = (self.aabb[3:] - self.aabb[:3]).max() / 2
        self.use_feat_predict = use_feat_predict
        self.use_weight_predict = use_weight_predict
        self.use_time_embedding = use_time_embedding
        self.use_time_attenuation = use_time_attenuation
        self.MOVING_STEP = moving_step
        self.use_div_offsets = use_div_offsets
        self.time_inject_before_sigma = time_inject_before_sigma
        self.loose_move = False
        self.motion_input_dim = 3 + 1
        self.motion_output_dim = 3 * 2 if use_div_offsets else 3
        self.return_extra = False
        self.hash_cfg = dict(base_res=base_resolution, max_res=dst_resolution, n_levels=n_levels,
                             log2_hashmap_size=log2_hashmap_size, temporal=bool(temporal_hash))
        self.time_mode = 0 if not use_time_embedding else (2 if use_time_attenuation else 1)
        self.set_mlp_precision(mlp_precision)
        if use_time_embedding:
            self.time_encoder = SinusoidalEncoder(1, 0, 4, True)
            self.time_encoder_feat = SinusoidalEncoderWithExp(1, 0, 4, True)
        base_in = 32 + (9 if self.time_mode else 0)

        gen = None
        if seed is not None:
            gen = torch.Generator().manual_seed(seed)
        tabs = level_tables(base_resolution, dst_resolution, n_levels, log2_hashmap_size)
        width = 8 if temporal_hash else 2
        table = torch.empty((tabs["total"], width), dtype=torch.float32).uniform_(-1e-4, 1e-4, generator=gen)
        self.hash_table = torch.nn.Parameter(table.to(hash_dtype), requires_grad=False)
        P = lambda o, i: torch.nn.Parameter(_xavier_(torch.empty(o, i), gen), requires_grad=False)
        self.xyz_wrap = torch.nn.ParameterList([P(64, 32), P(64, 64), P(64, 64), P(self.motion_output_dim, 64)])
        self.mlp_base = torch.nn.ParameterList([P(64, base_in), P(16, 64)])
        self.mlp_head = torch.nn.ParameterList([P(64, 19), P(64, 64), P(3, 64)])
        self._packed: Optional[torch.Tensor] = None
        self._packed_key = None
        self._desc: Optional[_lib.FieldDesc] = None
        self._desc_lock = __import__("threading").Lock()     # frames in flight share one field

    def set_mlp_precision(self, mlp_precision: str) -> "DNGPradianceField":
        """Arithmetic of the three MLPs (include/cednerf_hip.h): "f32" = exact fp32 MFMA chain, bit-identical
        to the CPU oracle; "f16x2" = split-fp16 MFMA, fp32-grade results; "f16" = fp16 operands with fp32
        accumulation, the precision class of the reference's tiny-cuda-nn networks (cednerf/model.py:200-222)."""
        if mlp_precision not in _lib.MLP_PRECISIONS:
            raise ValueError(f"mlp_precision must be one of {sorted(_lib.MLP_PRECISIONS)}, got {mlp_precision!r}")
        self.mlp_precision = mlp_precision
        return self

    # ---- parameter plumbing -------------------------------------------------------------------
    @classmethod
    def from_params(cls, params: Dict, device="cuda", mlp_precision: str = "f32") -> "DNGPradianceField":
        """Build from the plain-numpy parameter dict of ced_nerf_amd.synthetic.init_field_params."""
        h = params["hash"]
        table = torch.from_numpy(np.ascontiguousarray(h["table"]))
        f = cls(aabb=torch.from_numpy(np.asarray(params["aabb"], np.float32)), dst_resolution=h["max_res"],
                base_resolution=h["base_res"], n_levels=h["n_levels"], log2_hashmap_size=h["log2_hashmap_size"],
                moving_step=params["moving_step"], use_div_offsets=params["use_div_offsets"],
                use_time_embedding=params["time_mode"] != 0, use_time_attenuation=params["time_mode"] == 2,
                hash_dtype=table.dtype, temporal_hash=h.get("temporal", False), mlp_precision=mlp_precision)
        with torch.no_grad():
            f.hash_table.data = table
            for dst, src in ((f.xyz_wrap, params["xyz_wrap"]), (f.mlp_base, params["mlp_base"]),
                             (f.mlp_head, params["mlp_head"])):
                for p, w in zip(dst, src):
                    p.data = torch.from_numpy(np.ascontiguousarray(w, dtype=np.float32))
        return f.to(device)

    def export_params(self) -> Dict:
        g = lambda p: p.detach().cpu().numpy()
        return dict(aabb=g(self.aabb), moving_step=float(self.MOVING_STEP), use_div_offsets=self.use_div_offsets,
                    time_mode=self.time_mode, hash=dict(table=g(self.hash_table), **self.hash_cfg),
                    xyz_wrap=[g(p) for p in self.xyz_wrap], mlp_base=[g(p) for p in self.mlp_base],
                    mlp_head=[g(p) for p in self.mlp_head])

    def __getstate__(self):            # copy.deepcopy / pickle: drop the device-side caches and the lock
        state = self.__dict__.copy()
        for k in ("_desc_lock", "_desc", "_packed", "_packed_key"):
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self._packed, self._packed_key, self._desc = None, None, None
        self._desc_lock = __import__("threading").Lock()

    def _weights(self):
        return list(self.xyz_wrap) + list(self.mlp_base) + list(self.mlp_head)

    def _descriptor(self) -> _lib.FieldDesc:
        with self._desc_lock:
            return self._descriptor_locked()

    def _descriptor_locked(self) -> _lib.FieldDesc:
        ws = self._weights()
        key = (tuple((w.data_ptr(), w._version) for w in ws), self.hash_table.data_ptr(), str(self.hash_table.device),
               self.aabb.data_ptr(), self.aabb._version, self.mlp_precision)
        if self._desc is not None and key == self._packed_key:
            return self._desc
        dev = self.hash_table.device
        if dev.type != "cuda":
            raise NotImplementedError("Only support cuda inputs: move the field to a GPU (no CPU fallback).")
        blob = ops.pack_field_weights(self.use_div_offsets, self.time_mode,
                                      [w.detach().cpu().numpy() for w in self.xyz_wrap],
                                      [w.detach().cpu().numpy() for w in self.mlp_base],
                                      [w.detach().cpu().numpy() for w in self.mlp_head],
                                      _lib.MLP_PRECISIONS[self.mlp_precision])
        self._packed = torch.from_numpy(blob).to(dev)
        hd, _ = ops.make_hash_desc(self.hash_table.data, **self.hash_cfg)
        d = _lib.FieldDesc()
        aabb = self.aabb.detach().cpu().numpy().astype(np.float32)
        for i in range(6):
            d.aabb[i] = float(aabb[i])
        d.moving_step = float(np.float32(self.MOVING_STEP))
        d.use_div_offsets = int(self.use_div_offsets)
        d.time_mode = int(self.time_mode)
        d.mlp_precision = _lib.MLP_PRECISIONS[self.mlp_precision]
        d.packed_weights = self._packed.data_ptr()
        d.packed_floats = int(self._packed.numel())
        d.hash = hd
        self._desc = d
        self._packed_key = key
        return d

    # ---- reference API ------------------------------------------------------------------------
    @torch.no_grad()
    def hash_encoder(self, x: torch.Tensor, t: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The `hash_encoder(x_move)` call of cednerf/model.py:384: x [N,3] in [0,1] -> [N,32]."""
        d = self._descriptor()
        return ops.hash_encode(d.hash, x.reshape(-1, 3).float().contiguous(),
                               None if t is None else t.reshape(-1).float().contiguous())

    @torch.no_grad()
    def query_density(self, x, t, return_feat: bool = False, return_interal: bool = False):
        """cednerf/model.py:367-445 (eval: no training-only `interal_output`)."""
        if return_interal and self.training and (self.use_feat_predict or self.use_weight_predict):
            raise NotImplementedError("the feature/weight prediction heads (cednerf/model.py:428-443) run in training "
                                      "only; this module is the eval path")
        shp = x.shape
        _, sigma, geo = ops.field_forward(self._descriptor(), x.reshape(-1, 3).float().contiguous(),
                                          t.reshape(-1).float().contiguous(), None, want_geo=return_feat)
        results = {"density": sigma.view(list(shp[:-1]) + [1])}
        if return_feat:
            results["base_mlp_out"] = geo.view(list(shp[:-1]) + [self.geo_feat_dim])
        return results

    @torch.no_grad()
    def forward(self, positions: torch.Tensor, t: torch.Tensor, directions: torch.Tensor = None):
        """cednerf/model.py:468-488: returns (rgb [N,3], {'density': [N,1], 'base_mlp_out': [N,15]})."""
        if self.use_viewdirs and (directions is not None):
            assert positions.shape == directions.shape, f"{positions.shape} v.s. {directions.shape}"
        if directions is None:
            raise NotImplementedError("directions are required (use_viewdirs=True)")
        shp = positions.shape
        rgb, sigma, geo = ops.field_forward(self._descriptor(), positions.reshape(-1, 3).float().contiguous(),
                                            t.reshape(-1).float().contiguous(),
                                            directions.reshape(-1, 3).float().contiguous(), want_geo=True)
        results = {"density": sigma.view(list(shp[:-1]) + [1]),
                   "base_mlp_out": geo.view(list(shp[:-1]) + [self.geo_feat_dim])}
        return rgb.view(list(shp[:-1]) + [3]), results

    # fused closures of the render drivers (cednerf/utils.py:74-104,181-195)
    @torch.no_grad()
    def query_rays(self, rays_o, rays_d, ray_indices, t_starts, t_ends, timestamps, want_rgb: bool = True,
                   n_dev: Optional[torch.Tensor] = None):
        ts = timestamps.reshape(-1).float().contiguous()
        per_ray = bool(self.training)
        return ops.field_forward_rays(self._descriptor(), rays_o, rays_d, ray_indices, t_starts, t_ends, ts, per_ray,
                                      want_rgb, n_dev=n_dev)


def make_occ_eval_fn(radiance_field: "DNGPradianceField", timestamps: torch.Tensor, render_step_size: float):
    """The occ_eval_fn closure of train_real.py:324-328: a random training timestamp per point,
    density(x, t) * render_step_size."""
    def occ_eval_fn(x):
        t_idxs = torch.randint(0, len(timestamps), (x.shape[0],), device=x.device)
        t = timestamps[t_idxs]
        return radiance_field.query_density(x, t)["density"] * render_step_size
    return occ_eval_fn


# spelling used by BASELINE.json's north_star
DNGPRadianceField = DNGPradianceField
