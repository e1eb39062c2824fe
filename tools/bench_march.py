"""Latency probe of the marching kernel: a few rays through an EMPTY grid of varying resolution."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops
from ced_nerf_amd.nerfacc_api import ray_aabb_intersect, sort_intersections
dev = "cuda:0"
def run(n_rays, res, step, cone=0.0, occupied=False):
    rng = np.random.default_rng(0)
    o = np.tile(np.array([[-4.0, -3.7, -3.9]], np.float32), (n_rays, 1)) + rng.normal(size=(n_rays, 3)).astype(np.float32) * 0.01
    d = -o / np.linalg.norm(o, axis=1, keepdims=True)
    o = torch.from_numpy(o).to(dev); d = torch.from_numpy(d.astype(np.float32)).to(dev)
    b = torch.zeros((1, res, res, res), dtype=torch.bool, device=dev)
    if occupied:
        b[:] = True
    aabbs = torch.tensor([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], device=dev)
    tmin, tmax, hits = ray_aabb_intersect(o, d, aabbs)
    ts, ti = sort_intersections(tmin, tmax)
    near = torch.zeros(n_rays, device=dev); far = torch.full((n_rays,), 1e10, device=dev)
    counts = torch.empty(n_rays, dtype=torch.int64, device=dev)
    args = (o, d, b, aabbs, near, far, step, cone, 0, None, ts, ti, hits)
    for _ in range(3):
        ops.traverse_grids_raw(*args, 0, counts=counts)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.traverse_grids_raw(*args, 0, counts=counts)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3, int(counts.sum())
for res in (16, 32, 64, 128, 256):
    us, c = run(64, res, 5e-3)
    print(f"empty grid res={res:4d} 64 rays cone=0     : {us:8.1f} us  samples={c}")
for res in (32, 128):
    us, c = run(64, res, 1e-3, cone=0.004)
    print(f"empty grid res={res:4d} 64 rays cone=0.004 : {us:8.1f} us")
for res in (32, 128):
    us, c = run(64, res, 5e-3, occupied=True)
    print(f"full  grid res={res:4d} 64 rays            : {us:8.1f} us  samples={c}")
us, c = run(640000, 128, 5e-3)
print(f"empty grid res=128 640k rays : {us:8.1f} us")
us, c = run(64, 128, 5e-3)
