import sys
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from ced_nerf_amd import cameras, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import render_image_test
from ced_nerf_amd.video import render_video
DEV='cuda:0'
T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
W,H=96,72
sc=S.make_scene("dnerf",W,H,"trained",log2_hashmap_size=15)
cfg=sc["cfg"]
f=DNGPradianceField.from_params(sc["params"],DEV).eval()
est=OccGridEstimator(cfg["aabb"],cfg["grid_resolution"],cfg["grid_levels"]).to(DEV); est.set_binaries(T(sc["binaries"]))
rk=dict(sc["render"]); rk["render_bkgd"]=T(rk["render_bkgd"])
focal=0.5*W/np.tan(0.5*cfg["camera_angle_x"])
K=np.array([[focal,0,W/2.0],[0,focal,H/2.0],[0,0,1]],np.float32)
n_frames=5
poses=[S.look_at_c2w(cfg["radius"],30.0,15.0+20.0*k,cfg["opengl"]) for k in range(n_frames)]
times=[torch.tensor([[k/(n_frames-1.0)]],device=DEV) for k in range(n_frames)]
rays_of=lambda i: cameras.pinhole_rays(K,poses[i],W,H,cfg["opengl"],device=DEV)
alone=[render_image_test(1024,f,est,rays_of(i),timestamps=times[i],**rk) for i in range(n_frames)]
for rep in range(30):
    frames=render_video(f,est,rays_of,lambda i: times[i],n_frames,render_kwargs=rk,frames_in_flight=2,frames_per_call=2,keep_float=True)
    torch.cuda.synchronize()
    for i,(fr,want) in enumerate(zip(frames,alone)):
        for nm,a,b in (("rgb",fr["rgb_f32"],want[0]),("op",fr["opacity_f32"],want[1]),("dp",fr["depth_f32"],want[2])):
            if not torch.equal(a,b):
                d=(a-b).abs()
                print(f"rep {rep} frame {i} {nm}: {int((a!=b).sum())} values differ, max {float(d.max()):.3e}, first idx {torch.nonzero((a!=b).reshape(-1))[:5].reshape(-1).tolist()}", flush=True)
print("done")
