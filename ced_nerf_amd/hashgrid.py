"""Host-side geometry of the multi-resolution hash grid (product code).

Per-level scale / resolution / offset / size / hashed tables are computed once on the host in
float64 and handed to the HIP kernels as tables (SURVEY.md section 7 "hard parts"): the arithmetic follows
cednerf/taichi_kernel/hash_encoder_half.py:12-35 (align_to, res_in_level_np, scale_in_level_np) and
:268-292 (offsets, hash_map_sizes, begin_fast_hash_level) of the reference.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

MAX_LEVELS = 16


def level_tables(base_res: int = 16, max_res: int = 1024, n_levels: int = 16,
                 log2_hashmap_size: int = 21) -> Dict:
    if not (1 <= n_levels <= MAX_LEVELS):
        raise ValueError(f"n_levels must be in [1,{MAX_LEVELS}], got {n_levels}")
    max_params = 1 << int(log2_hashmap_size)
    log_b = np.log(float(max_res) / float(base_res)) / float(max(n_levels - 1, 1))
    scale = np.zeros(n_levels, np.float32)
    res = np.zeros(n_levels, np.uint32)
    offset = np.zeros(n_levels, np.uint32)
    size = np.zeros(n_levels, np.uint32)
    hashed = np.zeros(n_levels, np.uint32)
    running = 0
    for lvl in range(n_levels):
        s = float(base_res) * float(np.exp(float(lvl) * log_b)) - 1.0
        nearest = float(np.rint(s))
        if abs(s - nearest) < 1e-9:      # 16*64-1 evaluates to 1022.9999999999997: keep it integral
            s = nearest
        r = int(np.ceil(s)) + 1
        full = r * r * r
        aligned = -(-full // 8) * 8
        n_entries = min(max_params, aligned)
        scale[lvl] = np.float32(s)
        res[lvl] = r
        offset[lvl] = running
        size[lvl] = n_entries
        hashed[lvl] = 1 if full > n_entries else 0
        running += n_entries
    if running >= 2 ** 31:
        raise ValueError("hash table too large for 32-bit entry indices")
    return dict(n_levels=n_levels, scale=scale, res=res, offset=offset, size=size, hashed=hashed,
                total=running, log2_hashmap_size=int(log2_hashmap_size))



class HashEncoder(torch.nn.Module):
    """Trainable multi-resolution hash grid with the interface of the reference's `HashEncoder`
    (cednerf/taichi_kernel/hash_encoder_half.py:231-385): `forward(positions [N,3] in [0,1]) -> [N, levels*2]`,
    parameters in `hash_table [E,2]` (uniform +-1e-4, :313), gradients for the table and the positions through the
    HIP kernels ced_hash_encode / ced_hash_encode_backward.  `max_params` is the per-level cap (2**log2_hashmap_size).
    First piece of the training path."""

    def __init__(self, max_params: float = 2 ** 19, levels: int = 16, base_res: float = 16.0, max_res: float = 2048.0,
                 feature_per_level: int = 2, device="cuda", true_position_gradient: bool = False):
        super().__init__()
        # False: dL/dx as the reference's kernel computes it (no per-level `scale` factor, :212-226); True: the
        # gradient of the forward w.r.t. x (what a position-predicting network in front of the grid needs)
        self.true_position_gradient = bool(true_position_gradient)
        if feature_per_level != 2:
            raise NotImplementedError("the HIP hash grid stores 2 features per entry")
        log2T = int(round(np.log2(max_params)))
        if 2 ** log2T != int(max_params):
            raise ValueError("max_params must be a power of two")
        self.cfg = dict(n_levels=int(levels), max_res=int(max_res), base_res=int(base_res), log2_hashmap_size=log2T)
        tabs = level_tables(int(base_res), int(max_res), int(levels), log2T)
        self.hash_level = int(levels)
        self.out_dim = self.n_output_dims = 2 * int(levels)
        self.begin_fast_hash_level = int(np.argmax(tabs["hashed"])) if tabs["hashed"].any() else int(levels)
        table = torch.empty((tabs["total"], 2), dtype=torch.float32, device=device).uniform_(-1e-4, 1e-4)
        self.hash_table = torch.nn.Parameter(table, requires_grad=True)

    def _desc(self, table):
        from . import ops
        d, _ = ops.make_hash_desc(table, self.cfg["base_res"], self.cfg["max_res"], self.cfg["n_levels"],
                                  self.cfg["log2_hashmap_size"], False)
        return d

    def forward(self, positions: torch.Tensor) -> torch.Tensor:
        return _HashEncodeFunction.apply(positions.reshape(-1, 3).float().contiguous(), self.hash_table, self)


class _HashEncodeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, table, module):
        from . import ops
        if x.device.type != "cuda":
            raise NotImplementedError("Only support cuda inputs.")
        tab = table.detach().contiguous()
        out = ops.hash_encode(module._desc(tab), x.detach())
        ctx.save_for_backward(x.detach(), tab)
        ctx.module = module
        return out

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, tab = ctx.saved_tensors
        grad_table, dx = ops.hash_encode_backward(ctx.module._desc(tab), x, dy.float().contiguous(),
                                                  want_dx=ctx.needs_input_grad[0],
                                                  dx_scaled=ctx.module.true_position_gradient)
        return dx, grad_table, None


class TemporalHashEncoder(HashEncoder):
    """The reference's temporal `HashEncoder` (cednerf/taichi_kernel/hash_encoder_inter.py:281-430): entries hold four
    key-frames x two features, `forward(xyzt [N,4])` interpolates the key-frames k, k + 1 of t linearly (k = min(floor(3t),
    2)) before the trilinear sum, and the backward yields the TABLE gradient only (:202-275, :403-420: positions and
    times get none).  Same HIP kernels as the fused field's `temporal_hash=True`."""

    def __init__(self, max_params: float = 2 ** 19, levels: int = 16, base_res: float = 16.0, max_res: float = 2048.0,
                 feature_per_level: int = 2, device="cuda"):
        super().__init__(max_params, levels, base_res, max_res, feature_per_level, device)
        total = self.hash_table.shape[0]
        table = torch.empty((total, 8), dtype=torch.float32, device=device).uniform_(-1e-4, 1e-4)
        self.hash_table = torch.nn.Parameter(table, requires_grad=True)

    def _desc(self, table):
        from . import ops
        d, _ = ops.make_hash_desc(table, self.cfg["base_res"], self.cfg["max_res"], self.cfg["n_levels"],
                                  self.cfg["log2_hashmap_size"], True)
        return d

    def forward(self, xyzt: torch.Tensor) -> torch.Tensor:
        xyzt = xyzt.reshape(-1, 4).float()
        return _TemporalHashEncodeFunction.apply(xyzt[:, :3].contiguous(), xyzt[:, 3].contiguous(), self.hash_table, self)


class _TemporalHashEncodeFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, t, table, module):
        from . import ops
        if x.device.type != "cuda":
            raise NotImplementedError("Only support cuda inputs.")
        tab = table.detach().contiguous()
        out = ops.hash_encode(module._desc(tab), x.detach(), t.detach())
        ctx.save_for_backward(x.detach(), t.detach(), tab)
        ctx.module = module
        return out

    @staticmethod
    def backward(ctx, dy):
        from . import ops
        x, t, tab = ctx.saved_tensors
        grad_table = ops.hash_encode_backward_temporal(ctx.module._desc(tab), x, t, dy.float().contiguous())
        return None, None, grad_table, None
