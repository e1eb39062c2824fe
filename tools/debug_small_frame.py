"""Smallest end-to-end call of the native frame renderer (debugging aid): one 64x48 frame."""
import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ced_nerf_amd import synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
sc = S.make_scene("dnerf", 64, 48, "trained", log2_hashmap_size=15)
cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], DEV).eval()
est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
est.set_binaries(T(sc["binaries"]))
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
print("calling", flush=True)
out = render_image_test(int(sys.argv[1]) if len(sys.argv) > 1 else 64, f, est, Rays(T(sc["origins"]), T(sc["viewdirs"])), timestamps=T(sc["timestamps"]), **rk)
torch.cuda.synchronize()
print("ok", out[3], float(out[0].mean()), flush=True)
