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
units = int(os.environ.get("UNITS", "1"))       # frames per native call (ced_render_frames_test)
sc = S.make_scene("dnerf", 800, 800, "trained"); cfg = sc["cfg"]
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"]); ts = T(sc["timestamps"])
for prec in os.environ.get("MODES", "f32+h16x2,f32,f16x2,f16").split(","):
    f = DNGPradianceField.from_params(sc["params"], dev, mlp_precision=prec).eval()
    lanes, singles = [], []
    for k in range(3):
        os_, ds_ = [], []
        for u in range(units):
            c2w = S.look_at_c2w(cfg["radius"], 30.0, 30.0 + 12.0 * (k * units + u), cfg["opengl"])
            o, d = S.make_camera_rays(800, 800, cfg["camera_angle_x"], c2w, cfg["opengl"])
            os_.append(T(o)); ds_.append(T(d))
            one = ShardedRenderer(f, est, 1, 0, torch.device(dev), max_samples=1024, render_kwargs=rk, tile_order=True)
            one.set_rays(T(o)[None], T(d)[None]); singles.append(one)
        r = ShardedRenderer(f, est, 1, 0, torch.device(dev), max_samples=1024, render_kwargs=rk, tile_order=True, units=units)
        r.set_rays(torch.stack(os_), torch.stack(ds_)); lanes.append(r)
    alone = [l.render(ts) for l in singles]              # one frame at a time, full-chip launches
    ref = [dict(rgb=torch.cat([alone[k * units + u]["rgb"] for u in range(units)]),
                depth=torch.cat([alone[k * units + u]["depth"] for u in range(units)]),
                total_samples=sum(alone[k * units + u]["total_samples"] for u in range(units))) for k in range(3)]
    pipe = PipelinedRenderer(lanes)
    bad = 0
    for _ in range(reps):
        outs = pipe.render(ts)
        torch.cuda.synchronize()
        for a, b in zip(outs, ref):
            if not (torch.equal(a["rgb"], b["rgb"]) and torch.equal(a["depth"], b["depth"]) and a["total_samples"] == b["total_samples"]):
                bad += 1
    print(f"{prec}: {reps} pipelined steps x 3 calls x {units} frame(s), {bad} calls differ from the one-frame-at-a-time render", flush=True)
