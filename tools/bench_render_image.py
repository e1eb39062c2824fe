"""Times the render_image (sampling -> visibility filter -> rendering) path on the 800x800 frame."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image, render_image_test
dev="cuda:0"; T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
name=sys.argv[1] if len(sys.argv)>1 else "dnerf"
W,H={"dnerf":(800,800),"hypernerf":(536,960),"dynerf":(1352,1014)}[name]
dtype=np.float16 if (len(sys.argv)>2 and sys.argv[2]=="f16") else np.float32
sc=S.make_scene(name,W,H,"trained",table_dtype=dtype); cfg=sc["cfg"]
prec=os.environ.get("PRECISION","f32")
f=DNGPradianceField.from_params(sc["params"],dev,mlp_precision=prec).eval()
est=OccGridEstimator(cfg["aabb"],128,cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk=dict(sc["render"]); rk["render_bkgd"]=T(rk["render_bkgd"])
rays=Rays(T(sc["origins"]),T(sc["viewdirs"])); ts=T(sc["timestamps"])
for fn,nm in ((lambda: render_image(f,est,rays,timestamps=ts,**rk),"render_image"),(lambda: render_image_test(1024,f,est,rays,timestamps=ts,**rk),"render_image_test")):
    for _ in range(2): out=fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(5): out=fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/5
    print(f"{name} {W}x{H} {'fp16' if dtype==np.float16 else 'fp32'}-table mlp={prec} {nm}: {dt*1e3:.2f} ms/frame, samples={out[3]}, {out[3]/dt/1e9:.3f} Gsamples/s, {W*H/dt/1e6:.1f} Mrays/s")
