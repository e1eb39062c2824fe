#!/usr/bin/env python3
"""Round 4, on the MI355X: are the fp16-MFMA modes of the fused field kernel reproduced BIT FOR BIT by the oracle's
fp16-operand modes (oracle/mfma_f16_model.h)?  Field level (every model-flag case x two regimes x three modes), then
whole frames through render_image_test (schedule, counts, pixels).  Prints mismatch counts; exits non-zero on any.

    python tools/check_half_exact.py [--full]      # --full adds the 800x800 frames (oracle: minutes of host time)
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from ced_nerf_amd import synthetic as S  # noqa: E402
from ced_nerf_amd.model import DNGPradianceField  # noqa: E402
from ced_nerf_amd.nerfacc_api import OccGridEstimator  # noqa: E402
from ced_nerf_amd.utils import Rays, render_image_test  # noqa: E402
from oracle import oracle as O  # noqa: E402

DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)  # noqa: E731
N = lambda t: t.detach().cpu().numpy()  # noqa: E731
CASES = [dict(), dict(use_div_offsets=True), dict(use_time_embedding=True),
         dict(use_time_embedding=True, use_time_attenuation=True, use_div_offsets=True),
         dict(table_dtype=np.float16), dict(temporal_hash=True, table_dtype=np.float16, use_time_embedding=True)]


def nbad(a, b):
    a = np.ascontiguousarray(a, np.float32).reshape(-1); b = np.ascontiguousarray(b, np.float32).reshape(-1)
    return int((a.view(np.uint32) != b.view(np.uint32)).sum() - ((a == 0) & (b == 0) & (a.view(np.uint32) != b.view(np.uint32))).sum())


def main():
    O.build()
    bad_total = 0
    for ci, kw in enumerate(CASES):
        for regime in ("init", "trained"):
            p = S.init_field_params([-1.5] * 3 + [1.5] * 3, 1.0 / 64 if regime == "trained" else 1e-4, 1024, 17,
                                    regime=regime, seed=7 + ci, **kw)
            rng = np.random.default_rng(11)
            n = 20000 + 37
            pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)
            t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32); t[2] = 0; t[3] = 1
            d = rng.normal(size=(n, 3)).astype(np.float32)
            for prec in ("f16", "f16x2", "f32+h16x2"):
                want = O.OracleField(p, mlp_half=prec).forward(pos, t, d, want_geo=True)
                f = DNGPradianceField.from_params(p, DEV, mlp_precision=prec).eval()
                rgb, res = f(T(pos), T(t), T(d))
                b = (nbad(N(res["density"])[:, 0], want["density"]), nbad(N(res["base_mlp_out"]), want["base_mlp_out"]),
                     nbad(N(rgb), want["rgb"]))
                bad_total += sum(b)
                print(f"field case{ci} {regime:8s} {prec:10s}: differing sigma {b[0]} geo {b[1]} rgb {b[2]} of {n}"
                      f"   (max |rgb| err {np.abs(N(rgb) - want['rgb']).max():.2e})", flush=True)
    frames = [("dnerf", 80, 60, {}), ("hypernerf", 48, 64, {}), ("dnerf", 80, 60, {"table_dtype": np.float16})]
    if "--full" in sys.argv:
        frames += [("dnerf", 800, 800, {}), ("dnerf", 800, 800, {"table_dtype": np.float16})]
    for name, w, h, kw in frames:
        extra = dict(log2_hashmap_size=17) if w < 400 else {}
        sc = S.make_scene(name, w, h, "trained", **extra, **kw)
        cfg = sc["cfg"]
        oest = O.OracleEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"], sc["binaries"])
        est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
        est.set_binaries(T(sc["binaries"]))
        rays = Rays(origins=T(sc["origins"]), viewdirs=T(sc["viewdirs"]))
        rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
        plain = None
        for prec in ("f32", "f16", "f16x2", "f32+h16x2"):
            t0 = time.time()
            of = O.OracleField(sc["params"], mlp_half=prec)
            w_rgb, w_op, w_dp, w_total = O.render_image_test(1024, of, oest, sc["origins"], sc["viewdirs"],
                                                             timestamps=sc["timestamps"], **sc["render"])
            t_or = time.time() - t0
            if prec == "f32":
                plain = (w_rgb, w_op, w_dp, w_total)
            f = DNGPradianceField.from_params(sc["params"], DEV, mlp_precision=prec).eval()
            rgb, op, dp, total = render_image_test(1024, f, est, rays, timestamps=T(sc["timestamps"]), **rk)
            b = (nbad(N(rgb), w_rgb), nbad(N(op), w_op), nbad(N(dp), w_dp))
            bad_total += sum(b) + (total != w_total)
            print(f"frame {name} {w}x{h} {sorted(kw)} {prec:10s}: samples {total} vs oracle {w_total}; differing rgb {b[0]} "
                  f"opacity {b[1]} depth {b[2]}; vs PLAIN oracle: samples {total - plain[3]:+d}, rgb max {np.abs(N(rgb) - plain[0]).max():.2e} "
                  f"depth max {np.abs(N(dp) - plain[2]).max():.2e}  (oracle {t_or:.1f} s)", flush=True)
    print("TOTAL differing values:", bad_total)
    sys.exit(1 if bad_total else 0)


if __name__ == "__main__":
    main()
