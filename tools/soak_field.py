"""Soak test: run-to-run reproducibility of the fused field kernels over many 15 M-sample launches per mode."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import march_packed, OccGridEstimator
dev = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
reps = int(os.environ.get("REPS", "40"))
for scene, kw in (("dnerf", {}), ("hypernerf", {})):
    sc = S.make_scene(scene, 800 if scene == "dnerf" else 536, 800 if scene == "dnerf" else 960, "trained", **kw)
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], dev).eval()
    o = T(sc["origins"]).reshape(-1, 3); d = T(sc["viewdirs"]).reshape(-1, 3)
    est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
    n = o.shape[0]
    near = torch.full((n,), cfg["near_plane"], device=dev); far = torch.full((n,), cfg["far_plane"], device=dev)
    t0, t1, ri, _, _ = march_packed(o, d, est.binaries, est.aabbs, near, far, cfg["render_step_size"], cfg["cone_angle"])
    if t0.shape[0] > 16_000_000:
        t0, t1, ri = t0[:16_000_000].contiguous(), t1[:16_000_000].contiguous(), ri[:16_000_000].contiguous()
    ts = T(sc["timestamps"]).reshape(-1)
    for prec in ("f32", "f16x2", "f16"):
        f.set_mlp_precision(prec)
        ref = ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, True)
        ref = (ref[0].clone(), ref[1].clone())
        bad = 0
        t_a = time.time()
        for _ in range(reps):
            out = ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, True)
            if not (torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])):
                bad += 1
        torch.cuda.synchronize()
        print(f"{scene} {prec}: {reps} launches of {t0.shape[0]} samples, {bad} differ from the first ({time.time()-t_a:.1f} s)", flush=True)
