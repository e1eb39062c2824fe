"""Standalone throughput of the fused field kernel on a TEMPORAL hash table (hash_encoder_inter.py layout), random points."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import synthetic as S
from ced_nerf_amd.model import DNGPradianceField
dev = "cuda:0"
n = int(os.environ.get("N", str(1 << 22)))
for te in (False, True):
    kw = dict(use_time_embedding=True, use_time_attenuation=True, use_div_offsets=True) if te else {}
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-3, 1024, 19, regime="trained", temporal_hash=True, **kw)
    g = torch.Generator(device=dev).manual_seed(0)
    base = torch.rand(n // 8, 1, 3, device=dev, generator=g) * 1.8 - 0.9
    step = torch.randn(n // 8, 1, 3, device=dev, generator=g) * 0.002
    pos = (base + step * torch.arange(8, device=dev).view(1, 8, 1)).reshape(-1, 3).contiguous()
    t = torch.rand(n, 1, device=dev, generator=g)
    d = torch.randn(n, 3, device=dev, generator=g)
    for prec in ("f16x2", "f32"):
        f = DNGPradianceField.from_params(p, dev, mlp_precision=prec).eval()
        for _ in range(2):
            f(pos, t, d)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            f(pos, t, d)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"temporal table, time embedding {te}, {prec}: {ms:.3f} ms -> {n / ms * 1e3 / 1e9:.3f} Gsamples/s")
