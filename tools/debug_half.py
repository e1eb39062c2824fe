"""Debug aid: where do the half-precision field kernels deviate from the fp32 kernel?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import _lib, ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import march_packed, OccGridEstimator
dev = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sc = S.make_scene("dnerf", 800, 800, "trained")
cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev).eval()
o = T(sc["origins"]).reshape(-1, 3); d = T(sc["viewdirs"]).reshape(-1, 3)
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
n = o.shape[0]
near = torch.full((n,), cfg["near_plane"], device=dev); far = torch.full((n,), cfg["far_plane"], device=dev)
t0, t1, ri, packed, _ = march_packed(o, d, est.binaries, est.aabbs, near, far, cfg["render_step_size"], cfg["cone_angle"])
ts = T(sc["timestamps"]).reshape(-1)
N = t0.shape[0]
pos = o[ri] + d[ri] * (t0 + t1)[:, None] / 2.0
dirs = d[ri].contiguous(); tt = ts[0].expand(N).contiguous()
def run(prec, n_use, rays):
    f.set_mlp_precision(prec)
    if rays:
        r, s = ops.field_forward_rays(f._descriptor(), o, d, ri[:n_use].contiguous(), t0[:n_use].contiguous(), t1[:n_use].contiguous(), ts, False, True)
    else:
        r, s, _ = ops.field_forward(f._descriptor(), pos[:n_use].contiguous(), tt[:n_use].contiguous(), dirs[:n_use].contiguous())
    torch.cuda.synchronize()
    return r.clone(), s.clone()
for n_use in (5037, 200000, 2000000, N):
    for rays in (False, True):
        ref = run("f32", n_use, rays)
        for prec in os.environ.get("PRECS", "f16x2").split(","):
            a = run(prec, n_use, rays); b = run(prec, n_use, rays)
            err = (a[0] - ref[0]).abs().max(dim=1).values
            nd = ((a[0] != b[0]).any(dim=1) | (a[1] != b[1])).nonzero().flatten()
            print(f"   run-to-run differing samples {nd.numel()} first {nd[:10].tolist()} %32 hist {torch.bincount(nd % 32, minlength=32).tolist() if nd.numel() else ''}")
            sbad = (a[1] != ref[1]).nonzero().flatten()
            print(f"   sigma mismatches {sbad.numel()} first {sbad[:8].tolist()} %32 hist {torch.bincount(sbad % 32, minlength=32).tolist() if sbad.numel() else ''}")
            bad = (err > 2e-3).nonzero().flatten()
            print(f"n={n_use} rays={rays} {prec}: max err {err.max().item():.3e}, bad {bad.numel()}, deterministic {torch.equal(a[0], b[0])}",
                  "first bad", bad[:12].tolist(), "bad%32 hist", torch.bincount(bad % 32, minlength=32).tolist() if bad.numel() else "")
