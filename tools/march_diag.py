"""Where the marching kernel of the frame renderer spends its waves (diagnostic build, -DCED_MARCH_DIAG):
   SRCS=frame tools/build_variant.sh diag -DCED_MARCH_DIAG
   CED_NERF_LIB=build/variants/libcednerf_hip.diag.so python tools/march_diag.py [dnerf|hypernerf|dynerf]
Per phase of the walk: passes of a wave through it, lanes active in those passes, cycles until the wave's next tick.
First the first iteration alone (max_samples = min_samples), then the whole frame."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import _lib, synthetic as S
from ced_nerf_amd.model import DNGPradianceField
from ced_nerf_amd.nerfacc_api import OccGridEstimator
from ced_nerf_amd.utils import Rays, render_image_test
dev = "cuda:0"; T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
scene = sys.argv[1] if len(sys.argv) > 1 else "dnerf"
W, H = {"dnerf": (800, 800), "hypernerf": (536, 960), "dynerf": (1352, 1014)}[scene]
sc = S.make_scene(scene, W, H, "trained"); cfg = sc["cfg"]
f = DNGPradianceField.from_params(sc["params"], dev).eval()
est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev); est.set_binaries(T(sc["binaries"]))
from ced_nerf_amd.dist import tile_cyclic_assignment
o, d = sc["origins"].reshape(-1, 3), sc["viewdirs"].reshape(-1, 3)
if os.environ.get("TILE_ORDER", "1") == "1":          # the bench's ray order: 8x8-pixel tiles
    perm = tile_cyclic_assignment(1, H, W, 1)[1][0]
    o, d = o[perm], d[perm]
rays = Rays(T(o), T(d)); ts = T(sc["timestamps"])
rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
L = _lib.lib()
read = L.ced_diag_march_read
read.restype = C.c_int; read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
NAMES = ["ray set-up", "segment set-up", "field probe", "re-entry", "walk batch", "emission", "reserve+regen", "tail"]
waves = L.ced_diag_march_waves
waves.restype = C.c_int; waves.argtypes = [C.POINTER(C.c_uint), C.c_int, C.c_int]
def wave_report(skip=0):
    cap = 1 << 17
    buf = (C.c_uint * (cap * 24))()
    n = waves(buf, cap, 1)
    w = np.frombuffer(buf, dtype=np.uint32)[: n * 24].reshape(n, 8, 3).astype(np.float64)[skip:]
    n -= skip
    tot = w.sum(axis=0)
    print("  phases of these waves: " + "  ".join(f"{NAMES[p]} {tot[p,0]:.0f} passes x {tot[p,1]/max(tot[p,0],1):.0f} lanes, {tot[p,2]/tot[:,2].sum()*100:.0f} % of cycles" for p in range(7) if tot[p,0]))
    cyc = w[:, :, 2].sum(axis=1)
    order = np.argsort(-cyc)
    print(f"  {n} wave records; cycles per wave: median {np.median(cyc):.0f}, p90 {np.percentile(cyc, 90):.0f}, p99 {np.percentile(cyc, 99):.0f}, max {cyc.max():.0f}")
    heavy = cyc > 4 * np.median(cyc)
    print(f"  waves above 4x the median: {heavy.sum()} holding {cyc[heavy].sum() / cyc.sum() * 100:.0f} % of all wave cycles")
    print("  slowest waves (passes per phase: set-up seg probe re-entry walk emit | lanes per pass | cycles per phase in k):")
    for i in order[:8]:
        r = w[i]
        print("   ", " ".join(f"{int(v):4d}" for v in r[:6, 0]), "|", " ".join(f"{(r[p,1] / max(r[p,0],1)):4.0f}" for p in range(6)),
              "|", " ".join(f"{r[p,2]/1e3:6.0f}" for p in range(7)), f"| total {cyc[i]/1e3:.0f}k")
def report(title, max_samples, skip=0):
    render_image_test(max_samples, f, est, rays, timestamps=ts, **rk)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 24)()
    assert read(buf, 1) == 0
    waves(None, 0, 1)
    out = render_image_test(max_samples, f, est, rays, timestamps=ts, **rk)
    assert read(buf, 1) == 0
    v = np.array(list(buf), dtype=np.float64).reshape(8, 3)
    cyc = v[:, 2].sum()
    print(f"== {scene} {W}x{H} {title}: samples {out[3]}")
    for p in range(8):
        if v[p, 0] == 0: continue
        print(f"  {NAMES[p]:15s} passes {v[p,0]:12.0f}  lanes/pass {v[p,1]/v[p,0]:5.1f} ({v[p,1]/v[p,0]/64*100:4.0f} %)  "
              f"cycles {v[p,2]/cyc*100:5.1f} %  cycles/pass {v[p,2]/v[p,0]:8.0f}")
    util = (v[:, 2] * (v[:, 1] / np.maximum(v[:, 0], 1) / 64)).sum() / cyc
    print(f"  cycle-weighted lane utilisation {util*100:.0f} %")
    wave_report(skip)
ms = 1 if cfg["cone_angle"] == 0 else 4
report("first iteration", ms)
report("second iteration alone (rows after the first launch's)", ms + 1, skip=(W * H + 127) // 128 * 2 if cfg["grid_levels"] == 1 else 0)
report("whole frame", 1024)
