#!/usr/bin/env python3
"""Every frame bench.py times (800x800 turntable, azimuth 30 + 12 f degrees, f = 0..47) rendered on a strided ray set by
the HIP path in an fp16-MFMA mode and by the CPU oracle's mode of the same name: sample totals and every pixel compared
bit for bit.  (bench.py itself checks frames 0 / 23 / 47; this is the whole set, once per round.)

    python tools/oracle_mode_all_frames.py [--mode f16x2] [--stride 3] [--frames 48] > profiles/rNN_oracle_mode_all_frames.txt
"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--mode", default="f16x2"); ap.add_argument("--stride", type=int, default=3); ap.add_argument("--frames", type=int, default=48)
args = ap.parse_args()
dev = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
O.build()
sc = S.make_scene("dnerf", 800, 800, "trained", azim_deg=30.0)
cfg = sc["cfg"]
of = O.OracleField(sc["params"], mlp_half=args.mode)
plain = O.OracleField(sc["params"])
oest = O.OracleEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"], sc["binaries"])
f = DNGPradianceField.from_params(sc["params"], dev, mlp_precision=args.mode).eval()
est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
ts = T(sc["timestamps"])
bad = 0
print(f"# mode {args.mode}, every {args.stride}th pixel in x and y of 800x800, max_samples 1024; per frame: samples gpu / oracle({args.mode}) / "
      f"oracle(plain fp32), pixels differing from oracle({args.mode}), max |rgb| vs plain oracle", flush=True)
for k in range(args.frames):
    c2w = S.look_at_c2w(cfg["radius"], 30.0, 30.0 + 12.0 * k, cfg["opengl"])
    o, d = S.make_camera_rays(800, 800, cfg["camera_angle_x"], c2w, cfg["opengl"])
    o = np.ascontiguousarray(o[::args.stride, ::args.stride]); d = np.ascontiguousarray(d[::args.stride, ::args.stride])
    t0 = time.time()
    w = O.render_image_test(1024, of, oest, o, d, timestamps=sc["timestamps"], **sc["render"])
    p = O.render_image_test(1024, plain, oest, o, d, timestamps=sc["timestamps"], **sc["render"])
    g = render_image_test(1024, f, est, Rays(T(o), T(d)), timestamps=ts, **rk)
    torch.cuda.synchronize()
    gg = [x.cpu().numpy() for x in g[:3]]
    ndiff = sum(int((a.view(np.uint32) != np.ascontiguousarray(b.reshape(a.shape)).view(np.uint32)).sum()) for a, b in zip(gg, w[:3]))
    ok = ndiff == 0 and int(g[3]) == int(w[3])
    bad += not ok
    print(f"frame {k:2d}: samples {int(g[3])} / {int(w[3])} / {int(p[3])}  differing values {ndiff}  rgb vs plain {np.abs(gg[0] - p[0]).max():.2e}"
          f"  {'bit-exact' if ok else 'MISMATCH'}  ({time.time() - t0:.0f} s)", flush=True)
print(f"# {args.frames - bad} of {args.frames} frames bit-identical to the oracle's {args.mode} mode")
sys.exit(1 if bad else 0)
