"""Fused field kernel: launch time against the number of samples (the first n ray-ordered samples of the 800x800
D-NeRF-shaped frame's one-shot march) -- what a launch costs beyond its samples at the kernel's marginal rate.
PRECISION selects the arithmetic (default f32+h16x2)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator, march_packed
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sc = S.make_scene("dnerf", 800, 800, "trained"); cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev, mlp_precision=os.environ.get("PRECISION", "f32+h16x2")).eval()
o = T(sc["origins"]).reshape(-1, 3); d = T(sc["viewdirs"]).reshape(-1, 3)
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
n = o.shape[0]
near = torch.full((n,), cfg["near_plane"], device=dev); far = torch.full((n,), cfg["far_plane"], device=dev)
t0, t1, ri, packed, _ = march_packed(o, d, est.binaries, est.aabbs, near, far, cfg["render_step_size"], cfg["cone_angle"])
ts = T(sc["timestamps"]).reshape(-1)
rows = []
for k in (32, 3072 * 32, 3072 * 32 * 2, 3072 * 32 * 4, 500000, 3072 * 32 * 5, 3072 * 32 * 6, 1000000, 2000000, 4000000, 8000000):
    a, b, c = t0[:k].contiguous(), t1[:k].contiguous(), ri[:k].contiguous()
    for _ in range(3):
        ops.field_forward_rays(f._descriptor(), o, d, c, a, b, ts, False, True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.field_forward_rays(f._descriptor(), o, d, c, a, b, ts, False, True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    rows.append((k, us))
    print(f"n = {k:8d} ({k / (3072 * 32):6.2f} tiles per wave): {us:8.1f} us  {k / us / 1e3:6.3f} Gsamples/s")
(k1, u1), (k2, u2) = rows[-3], rows[-1]
rate = (k2 - k1) / (u2 - u1)
print(f"marginal rate {rate / 1e3:.3f} Gsamples/s; fixed cost by that rate: " + ", ".join(f"{k}: {u - k / rate:.1f} us" for k, u in rows))
