"""Soak test: frames in flight (PipelinedRenderer, half-chip field launches) reproduce the one-at-a-time images."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import synthetic as S
from ced_nerf_amd.dist import PipelinedRenderer, ShardedRenderer
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
reps = int(os.environ.get("REPS", "100"))
sc = S.make_scene("dnerf", 800, 800, "trained"); cfg = sc["cfg"]
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"]); ts = T(sc["timestamps"])
for prec in ("f32", "f16x2", "f16"):
    f = DNGPradianceField.from_params(sc["params"], dev, mlp_precision=prec).eval()
    lanes = []
    for k in range(3):
        c2w = S.look_at_c2w(cfg["radius"], 30.0, 30.0 + 12.0 * k, cfg["opengl"])
        o, d = S.make_camera_rays(800, 800, cfg["camera_angle_x"], c2w, cfg["opengl"])
        r = ShardedRenderer(f, est, 1, 0, torch.device(dev), max_samples=1024, render_kwargs=rk, tile_order=True)
        r.set_rays(T(o)[None], T(d)[None]); lanes.append(r)
    PipelinedRenderer.restore_field_blocks()
    ref = [l.render(ts) for l in lanes]                   # one at a time, full-chip launches
    pipe = PipelinedRenderer(lanes)
    bad = 0
    for _ in range(reps):
        outs = pipe.render(ts)
        torch.cuda.synchronize()
        for a, b in zip(outs, ref):
            if not (torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["depth"], b["depth"]) and a["total_samples"] == b["total_samples"]):
                bad += 1
    print(f"{prec}: {reps} pipelined steps x 3 frames, {bad} frames differ from the one-at-a-time render", flush=True)
