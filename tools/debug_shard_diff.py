"""Where do frames rendered in shards (schedule-local) differ from the same frames rendered whole?  Emulates the
ranks of a weak-scaling step one after the other on one GPU and lists the worst pixels."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import synthetic as S, dist as cdist
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
world, units = 2, 3
sc = S.make_scene("dnerf", W, W, "trained", azim_deg=30.0); cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev).eval()
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"]); ts = T(sc["timestamps"])
frames = []
for k in range(units * world):
    c2w = S.look_at_c2w(cfg["radius"], 30.0, 30.0 + 12.0 * k, cfg["opengl"])
    o, d = S.make_camera_rays(W, W, cfg["camera_angle_x"], c2w, cfg["opengl"])
    frames.append((T(o), T(d)))
O_ = torch.stack([a for a, _ in frames]); D_ = torch.stack([b for _, b in frames])
n = units * world * W * W
img = torch.zeros((n + 1, 5), device=dev)
for r in range(world):
    sr = cdist.ShardedRenderer(f, est, world, r, torch.device(dev), max_samples=1024, render_kwargs=rk, units=units)
    sr.set_rays(O_, D_)
    rgb, op, dp, ns = sr.render_local(ts)
    dest = sr.gather_index.view(world, -1)[r][:sr.n_pad]
    img[dest] = torch.cat([rgb.view(-1, 3), op.view(-1, 1), dp.view(-1, 1)], 1)
img = img[:n].view(units * world, W, W, 5)
worst = 0
for k in range(units * world):
    s = render_image_test(1024, f, est, Rays(*frames[k]), timestamps=ts, **rk)
    whole = torch.cat([s[0], s[1], s[2]], -1)
    diff = (img[k] - whole).abs()
    m = diff.max().item(); worst = max(worst, m)
    idx = torch.nonzero(diff.max(-1).values > 1e-4)
    print(f"frame {k}: max diff rgb {diff[..., :3].max().item():.3e} op {diff[..., 3].max().item():.3e} depth {diff[..., 4].max().item():.3e}; pixels > 1e-4: {idx.shape[0]}")
    for y, x in idx[:5].tolist():
        print("   pixel", y, x, "shard", [round(v, 6) for v in img[k, y, x].tolist()], "whole", [round(v, 6) for v in whole[y, x].tolist()])
print("worst", worst)
