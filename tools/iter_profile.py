"""Per-iteration view of one 800x800 frame (render_image_test): alive rays, samples, field-kernel time and rate."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
scene = sys.argv[1] if len(sys.argv) > 1 else "dnerf"
W, H = {"dnerf": (800, 800), "hypernerf": (536, 960), "dynerf": (1352, 1014)}[scene]
sc = S.make_scene(scene, W, H, "trained"); cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev, mlp_precision=os.environ.get("PRECISION", "f32")).eval()
est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rays = Rays(T(sc["origins"]), T(sc["viewdirs"])); ts = T(sc["timestamps"])
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
tr = ops.FrameTracer(capacity=96, with_events=True)
for _ in range(3):
    out = render_image_test(1024, f, est, rays, timestamps=ts, tracer=tr, **rk)
torch.cuda.synchronize()
import time
t_a = time.perf_counter()
for _ in range(5):
    out = render_image_test(1024, f, est, rays, timestamps=ts, tracer=tr, **rk)
torch.cuda.synchronize()
frame_ms = (time.perf_counter() - t_a) / 5 * 1e3
its, ms = tr.iterations(), tr.field_ms()
tot = 0.0
for i, (it, m) in enumerate(zip(its, ms)):
    tiles = (it["n_new"] + 31) // 32
    print(f"iter {i:2d}: alive {it['n_alive']:7d} n_samples {it['n_samples']:3d} samples {it['n_new']:8d} tiles {tiles:7d} "
          f"({tiles / (256 * 12):6.2f} per wave at 256 wg) field {m*1e3:7.1f} us  {it['n_new'] / m / 1e6:7.2f} Gsamples/s")
    tot += m
print(f"{scene} {W}x{H}: frame {frame_ms:.3f} ms; total samples {out[3]}, field {tot:.3f} ms -> {out[3] / tot / 1e6:.2f} Gsamples/s inside "
      f"field kernels; outside the field kernel {frame_ms - tot:.3f} ms")
