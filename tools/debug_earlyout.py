import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import _lib, ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
dev="cuda:0"; T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
name=sys.argv[1] if len(sys.argv)>1 else "dynerf"
sc=S.make_scene(name,64,48,"trained",log2_hashmap_size=17); cfg=sc["cfg"]
f=DNGPradianceField.from_params(sc["params"],dev).eval()
est=OccGridEstimator(cfg["aabb"],128,cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk=dict(sc["render"]); rk["render_bkgd"]=T(rk["render_bkgd"])
rays=Rays(T(sc["origins"]),T(sc["viewdirs"]))
res={}
for eo in (0,1):
    _lib.check(_lib.lib().ced_set_option(b"march_early_out", eo))
    tr=ops.FrameTracer(1100, with_events=False)
    out=render_image_test(1024,f,est,rays,timestamps=T(sc["timestamps"]),tracer=tr,**rk)
    res[eo]=(out,tr.iterations())
    print(eo, out[3], tr.iterations()[:6])
print("binaries per level occupied:", sc["binaries"].reshape(sc["binaries"].shape[0],-1).sum(1))
