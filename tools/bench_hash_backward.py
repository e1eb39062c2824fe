"""Micro-benchmark of ced_hash_encode_backward on the marched samples' positions of the 800x800 frame."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops, synthetic as S
dev = "cuda:0"
log2T = int(os.environ.get("LOG2T", "21"))
p = S.init_field_params([-1.5] * 3 + [1.5] * 3, 1e-4, 1024, log2T, regime="trained")
table = torch.from_numpy(p["hash"]["table"]).to(dev)
desc, tabs = ops.make_hash_desc(table, 16, 1024, 16, log2T, False)
n = int(os.environ.get("N", str(1 << 22)))
g = torch.Generator(device=dev).manual_seed(0)
# ray-coherent points: short runs along random directions (as marched samples are)
base = torch.rand(n // 8, 1, 3, device=dev, generator=g) * 0.9 + 0.05
step = torch.randn(n // 8, 1, 3, device=dev, generator=g) * 0.002
x = (base + step * torch.arange(8, device=dev).view(1, 8, 1)).reshape(-1, 3).clamp(0, 1).contiguous()
dy = torch.randn(x.shape[0], 32, device=dev, generator=g)
grad = torch.zeros((int(desc.total_entries), 2), device=dev)
for want_dx in (True, False):
    for _ in range(2):
        ops.hash_encode_backward(desc, x, dy, grad_table=grad, want_dx=want_dx)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.hash_encode_backward(desc, x, dy, grad_table=grad, want_dx=want_dx)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    ns = x.shape[0]
    print(f"hash backward T=2^{log2T} n={ns} want_dx={want_dx}: {ms:.3f} ms -> {ns/ms*1e3/1e9:.3f} Gsamples/s, "
          f"{ns*256/ms*1e3/1e9:.1f} G atomic adds/s, {ns*(256*4 + 12 + 128)/ms*1e3/1e9:.0f} GB/s algorithmic (1024 B atomics + 140 B in)")
