"""Micro-benchmark of the fused field kernel on a realistic, ray-coherent sample stream
(all samples marched through the occupancy grid of the 800x800 D-NeRF-shaped frame)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import _lib, ops, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import march_packed

dev = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
scene = sys.argv[1] if len(sys.argv) > 1 else "dnerf"
dtype = np.float16 if (len(sys.argv) > 2 and sys.argv[2] == "f16") else np.float32
W = H = 800 if scene == "dnerf" else 600
log2T = int(os.environ.get("LOG2T", "21"))
sc = S.make_scene(scene, W, H, "trained", table_dtype=dtype, log2_hashmap_size=log2T)
print("log2T", log2T, "table MB", sc["params"]["hash"]["table"].nbytes / 1e6)
cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev).eval()
o = T(sc["origins"]).reshape(-1, 3); d = T(sc["viewdirs"]).reshape(-1, 3)
from ced_nerf_amd.nerfacc_api import OccGridEstimator
est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
n = o.shape[0]
near = torch.full((n,), cfg["near_plane"], device=dev); far = torch.full((n,), cfg["far_plane"], device=dev)
t0, t1, ri, packed, _ = march_packed(o, d, est.binaries, est.aabbs, near, far, cfg["render_step_size"], cfg["cone_angle"])
ts = T(sc["timestamps"]).reshape(-1)
N = t0.shape[0]
print("samples", N)
prec = os.environ.get("PRECISION", "f32")
ref = None
if prec != "f32":
    ref = ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, True)
    ref = (ref[0].clone(), ref[1].clone())
    f.set_mlp_precision(prec)
res = {}
for hv in [int(v) for v in os.environ.get("HALF_VARIANTS", "0").split(",")]:
    if prec == "f32":
        continue
    _lib.check(_lib.lib().ced_set_option(b"half_variant", hv))
    for _ in range(2):
        out = ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, True)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    drgb = (out[0] - ref[0]).abs().max().item()
    rel = ((out[1] - ref[1]).abs() / ref[1].abs().clamp_min(1e-6)).max().item()
    dsig = (out[1] - ref[1]).abs().max().item()
    print(f"precision {prec} half_variant {hv}: {ms:.3f} ms -> {N/ms*1e3/1e9:.3f} Gsamples/s; vs f32: max|drgb| {drgb:.3e}, "
          f"max|dsigma| {dsig:.3e} (max sigma {ref[1].max().item():.3e}), max rel dsigma {rel:.3e}")
if prec != "f32":
    sys.exit(0)
for variant in [int(v) for v in os.environ.get('VARIANTS', '2').split(',')]:
    _lib.check(_lib.lib().ced_set_option(b"field_variant", variant))
    for stg in [int(v) for v in os.environ.get('STAGGER', '0').split(',')]:
        _lib.check(_lib.lib().ced_set_option(b"field_stagger", stg))
        for want_rgb in (True,):
            for _ in range(2):
                ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, want_rgb)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, want_rgb)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"variant {variant} stagger {stg} rgb={want_rgb}: {ms:.3f} ms -> {N/ms*1e3/1e9:.3f} Gsamples/s, {N*38e3/ms*1e3/1e12:.1f} TFLOP/s(alg)")
