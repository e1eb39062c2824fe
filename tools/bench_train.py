"""Timing of train.train_step on the 800x800 D-NeRF-shaped scene (random ray batches of one view)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import synthetic as S
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.train import TrainableField, train_step
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sc = S.make_scene("dnerf", 800, 800, "trained"); cfg = sc["cfg"]
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
field = TrainableField(sc["params"], dev)
# the reference trains with apex FusedAdam (train_real.py:269): torch's fused Adam is the counterpart here.  LR=0 keeps the
# parameters (hence the sample counts) the same from step to step and run to run: a stable timing workload.
opt = torch.optim.Adam(field.parameters(), lr=float(os.environ.get("LR", "0")), eps=1e-15,
                       fused=os.environ.get("ADAM_FUSED", "1") == "1")
o = T(sc["origins"]).reshape(-1, 3); d = T(sc["viewdirs"]).reshape(-1, 3); ts = T(sc["timestamps"])
bk = T(sc["render"]["render_bkgd"])
g = torch.Generator(device=dev).manual_seed(0)
for n_rays in [int(v) for v in os.environ.get("N_RAYS", "16384,65536,262144").split(",")]:
    target = torch.rand(n_rays, 3, device=dev, generator=g)
    samples = []
    for it in range(13):
        if it == 3:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        idx = torch.randint(0, o.shape[0], (n_rays,), device=dev, generator=g)
        out = train_step(field, est, opt, o[idx].contiguous(), d[idx].contiguous(), ts, target, cfg["render_step_size"],
                         near_plane=cfg["near_plane"], far_plane=cfg["far_plane"], render_bkgd=bk,
                         overlap_table_grad=os.environ.get("OVERLAP", "1") == "1")
        samples.append(out["n_samples"])
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"train_step {n_rays} rays: {dt*1e3:.1f} ms/step, {np.mean(samples[3:]):.0f} samples kept/step, "
          f"{n_rays/dt/1e6:.2f} Mrays/s, {np.mean(samples[3:])/dt/1e6:.1f} Msamples/s")
