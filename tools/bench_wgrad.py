"""Timing of ced_weight_grad (dW = dy^T x over the sample stream) against the library GEMM, per layer shape."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
for n in (330_000, 1_200_000, 4_200_000):
    for n_out, n_in in ((64, 64), (64, 32), (16, 64), (6, 64), (3, 64), (64, 19)):
        x = torch.randn(n, n_in, device=dev, generator=g); dy = torch.randn(n, n_out, device=dev, generator=g)
        res = {}
        for name, fn in (("hip", lambda: ops.weight_grad(x, dy)), ("library", lambda: dy.t() @ x)):
            for _ in range(3):
                fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / 10
        gb = n * (n_in + n_out) * 4 / 1e9
        print(f"n={n:8d} dW[{n_out:2d}x{n_in:2d}]: hip {res['hip']*1e3:7.1f} us ({gb/res['hip']*1e3:6.0f} GB/s algorithmic), "
              f"library {res['library']*1e3:7.1f} us")
