#!/usr/bin/env python3
"""Finds the samples on which a half-precision mode of the field kernel and the oracle's mode differ; prints them and
saves their inputs (gpurun_out/r4_half/mismatch.npz) for the block-level replay (tools/probes/mfma_replay.py)."""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from ced_nerf_amd import synthetic as S  # noqa: E402
from ced_nerf_amd.model import DNGPradianceField  # noqa: E402
from oracle import oracle as O  # noqa: E402
from tools.check_half_exact import CASES, N, T, DEV  # noqa: E402

O.build()
out = {}
for ci, regime, prec in [(0, "init", "f16"), (0, "trained", "f16"), (2, "trained", "f16x2"), (2, "trained", "f32+h16x2"),
                         (3, "trained", "f16x2"), (5, "init", "f16x2")]:
    kw = CASES[ci]
    p = S.init_field_params([-1.5] * 3 + [1.5] * 3, 1.0 / 64 if regime == "trained" else 1e-4, 1024, 17, regime=regime, seed=7 + ci, **kw)
    rng = np.random.default_rng(11)
    n = 20000 + 37
    pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)
    t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32); t[2] = 0; t[3] = 1
    d = rng.normal(size=(n, 3)).astype(np.float32)
    want = O.OracleField(p, mlp_half=prec).forward(pos, t, d, want_geo=True)
    f = DNGPradianceField.from_params(p, DEV, mlp_precision=prec).eval()
    rgb, res = f(T(pos), T(t), T(d))
    g_rgb, g_geo, g_sig = N(rgb), N(res["base_mlp_out"]), N(res["density"])[:, 0]
    bad = np.nonzero((g_rgb != want["rgb"]).any(1) | (g_geo != want["base_mlp_out"]).any(1) | (g_sig != want["density"]))[0]
    print(f"case{ci} {regime} {prec}: {len(bad)} samples differ: {bad.tolist()}")
    for i in bad[:12]:
        print(f"   s{i}: pos {pos[i]} t {t[i]} d {d[i]} |d| {np.linalg.norm(d[i]):.6g}")
        print(f"        rgb gpu {g_rgb[i]} oracle {want['rgb'][i]}  sigma gpu {g_sig[i]:.9g} oracle {want['density'][i]:.9g}")
        gd = np.nonzero(g_geo[i] != want["base_mlp_out"][i])[0]
        for q in gd:
            print(f"        geo[{q}] gpu {g_geo[i, q]!r} oracle {want['base_mlp_out'][i, q]!r}")
    key = f"c{ci}_{regime}_{prec}"
    out[key + "_idx"] = bad; out[key + "_gpu_rgb"] = g_rgb[bad]; out[key + "_gpu_geo"] = g_geo[bad]
np.savez("gpurun_out/r4_half/mismatch.npz", **out)
