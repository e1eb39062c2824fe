"""Experiment: K independent 800x800 frames in flight on K streams / K host threads."""
import os, sys, time, threading
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import _lib, ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
dev="cuda:0"; T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
K=int(sys.argv[1]) if len(sys.argv)>1 else 2
variant=int(sys.argv[2]) if len(sys.argv)>2 else 2
_lib.check(_lib.lib().ced_set_option(b"field_variant", variant))
scs=[S.make_scene("dnerf",800,800,"trained",azim_deg=30.0+12.0*f) for f in range(K)]
cfg=scs[0]["cfg"]
f=DNGPradianceField.from_params(scs[0]["params"],dev).eval()
est=OccGridEstimator(cfg["aabb"],128,1).to(dev); est.set_binaries(T(scs[0]["binaries"]))
rk=dict(scs[0]["render"]); rk["render_bkgd"]=T(rk["render_bkgd"])
rays=[Rays(T(s["origins"]),T(s["viewdirs"])) for s in scs]; ts=T(scs[0]["timestamps"])
f._descriptor()
streams=[torch.cuda.Stream() for _ in range(K)]
results=[None]*K
def work(i, reps):
    with torch.cuda.stream(streams[i]):
        for _ in range(reps):
            results[i]=render_image_test(1024,f,est,rays[i],timestamps=ts,**rk)
def run(reps):
    th=[threading.Thread(target=work,args=(i,reps)) for i in range(K)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
run(3)
t0=time.perf_counter(); reps=10; run(reps); dt=time.perf_counter()-t0
tot=sum(r[3] for r in results)*reps
print(f"K={K} variant={variant}: {dt/reps*1e3:.3f} ms per {K} frames -> {dt/reps/K*1e3:.3f} ms/frame, {tot/dt/1e9:.3f} Gsamples/s, {K*640000*reps/dt/1e6:.1f} Mrays/s")
