"""One 800x800 frame alone: per field launch the HIP event pair, the kernel's own device stamps, and (when run under
rocprofv3 --kernel-trace) the dispatch's start/end from the trace -- three views of the same launches."""
import os, sys, glob, csv
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sc = S.make_scene("dnerf", 800, 800, "trained"); cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev).eval()
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
rays = Rays(T(sc["origins"]), T(sc["viewdirs"])); ts = T(sc["timestamps"])
tr = ops.FrameTracer(capacity=64, with_events=True); tr.enable_device_stamps(torch.device(dev))
for _ in range(3):
    render_image_test(1024, f, est, rays, timestamps=ts, tracer=tr, **rk)
torch.cuda.synchronize()
ev = tr.field_ms(); dv = tr.field_intervals_device()
print("khz", ops._lib.lib().ced_wall_clock_khz())
for i, (e, d) in enumerate(zip(ev, dv)):
    print(f"iter {i:2d}: events {e*1e3:8.1f} us   stamps {(d[1]-d[0])*1e3:8.1f} us   start-to-next-start {((dv[i+1][0]-d[0])*1e3 if i+1 < len(dv) else 0):8.1f} us")
