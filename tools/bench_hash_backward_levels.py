"""Per-level cost of the table-gradient kernel (ced_hash_encode_backward, want_dx=False) on ray-coherent positions:
one launch per level (a descriptor that holds just that level), so that contention on the small dense levels and the
request-bound fine levels show separately.  N (default 1.59 M = the 262 k-ray training step's kept samples)."""
import copy, ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops, synthetic as S
dev = "cuda:0"
p = S.init_field_params([-1.5] * 3 + [1.5] * 3, 1e-4, 1024, 21, regime="trained")
table = torch.from_numpy(p["hash"]["table"]).to(dev)
desc, tabs = ops.make_hash_desc(table, 16, 1024, 16, 21, False)
n = int(os.environ.get("N", "1590000")) // 8 * 8
g = torch.Generator(device=dev).manual_seed(0)
base = torch.rand(n // 8, 1, 3, device=dev, generator=g) * 0.9 + 0.05
step = torch.randn(n // 8, 1, 3, device=dev, generator=g) * 0.002
x = (base + step * torch.arange(8, device=dev).view(1, 8, 1)).reshape(-1, 3).clamp(0, 1).contiguous()
grad = torch.zeros((int(desc.total_entries), 2), device=dev)


def timed(d, dy, reps=5):
    for _ in range(2):
        ops.hash_encode_backward(d, x, dy, grad_table=grad, want_dx=False)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.hash_encode_backward(d, x, dy, grad_table=grad, want_dx=False)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


dy = torch.randn(n, 32, device=dev, generator=g)
all_ms = timed(desc, dy)
print(f"all 16 levels, n = {n}: {all_ms:.3f} ms  ({n * 256 * 4 / all_ms / 1e9:.3f} TB/s of added bytes)")
tot = 0.0
for l in range(16):
    d1 = type(desc)()
    C.memmove(C.byref(d1), C.byref(desc), C.sizeof(desc))
    d1.n_levels = 1
    for nm in ("scale", "res", "offset", "size", "hashed"):
        getattr(d1, nm)[0] = getattr(desc, nm)[l]
    ms = timed(d1, dy[:, 2 * l:2 * l + 2].contiguous())
    tot += ms
    print(f"level {l:2d}: res {int(desc.res[l]):5d} size {int(desc.size[l]):8d} {'hashed' if desc.hashed[l] else 'dense '}  {ms:.3f} ms"
          f"  ({n * 16 * 4 / ms / 1e9:.3f} TB/s)")
print(f"sum of per-level launches {tot:.3f} ms")
