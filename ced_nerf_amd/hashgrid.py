"""Host-side geometry of the multi-resolution hash grid (product code).

Per-level scale / resolution / offset / size / hashed tables are computed once on the host in
float64 and handed to the HIP kernels as tables (SURVEY.md section 7 "hard parts"): the arithmetic follows
cednerf/taichi_kernel/hash_encoder_half.py:12-35 (align_to, res_in_level_np, scale_in_level_np) and
:268-292 (offsets, hash_map_sizes, begin_fast_hash_level) of the reference.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

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
