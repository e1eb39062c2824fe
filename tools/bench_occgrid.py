"""Timing of the occupancy-grid refresh (nerfacc OccGridEstimator.update_every_n_steps, train_real.py:324-336) on the
D-NeRF- and DyNeRF-shaped configurations: the density of the sampled cells comes from the fused field kernel."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import synthetic as S
from ced_nerf_amd.model import DNGPradianceField, make_occ_eval_fn
from ced_nerf_amd.nerfacc_api import OccGridEstimator
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for scene in (sys.argv[1:] or ["dnerf", "dynerf"]):
    sc = S.make_scene(scene, 64, 64, "trained"); cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], dev).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
    est.train()
    fn = make_occ_eval_fn(f, T(sc["timestamps"]), cfg["render_step_size"])
    for step in range(0, 16 * 4, 16):
        est.update_every_n_steps(step=step + 256, occ_eval_fn=fn, occ_thre=1e-2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for k in range(n):
        est.update_every_n_steps(step=1024 + 16 * k, occ_eval_fn=fn, occ_thre=1e-2)
    torch.cuda.synchronize()
    print(f"{scene}: occupancy-grid update {((time.perf_counter() - t0) / n) * 1e3:.2f} ms ({cfg['grid_levels']} level(s) of {cfg['grid_resolution']}^3)")
