"""Diagnostic (library built with -DCED_MARCH_PROFILE): where the marching kernel's waves spend their cycles,
per iteration of one 800x800 frame.  usage: CED_NERF_LIB=build/libmprof.so python tools/debug_march_profile.py"""
import os, sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import _lib, ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
scene = sys.argv[1] if len(sys.argv) > 1 else "dnerf"
W, H = {"dnerf": (800, 800), "hypernerf": (536, 960), "dynerf": (1352, 1014)}[scene]
sc = S.make_scene(scene, W, H, "trained"); cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev).eval()
est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
rays = Rays(T(sc["origins"]), T(sc["viewdirs"])); ts = T(sc["timestamps"])
L = _lib.lib()
names = ["prologue", "segment select", "skip-march", "DDA set-up", "brick probes", "look-ahead DDA", "occupancy wait + emission", "reserve + copy-out"]
prev = None
for max_samples in ((1, 2, 4, 7, 11, 17, 26, 41, 71, 132, 1024) if scene == "dnerf" else (4, 34, 1024)):      # cumulative: iteration k is the difference
    buf = (C.c_ulonglong * 16)()
    L.ced_debug_march_profile(None, 1)
    render_image_test(max_samples, f, est, rays, timestamps=ts, **rk)
    torch.cuda.synchronize()
    L.ced_debug_march_profile(buf, 0)
    v = np.array(list(buf), np.float64)
    print(f"max_samples {max_samples:5d}: waves {int(v[8]):6d}  total Mcycles {v[:8].sum()/1e6:8.1f}  " +
          "  ".join(f"{n} {100*x/max(v[:8].sum(),1):.0f}%" for n, x in zip(names, v[:8])))
    if prev is not None:
        dv = v - prev
        tot = max(dv[:8].sum(), 1)
        print(f"      this iteration alone: waves {int(dv[8]):6d}, Mcycles {tot/1e6:7.1f}, per wave {tot/max(dv[8],1)/1e3:6.1f} k: " +
              "  ".join(f"{n} {100*x/tot:.0f}%" for n, x in zip(names, dv[:8])))
    prev = v
