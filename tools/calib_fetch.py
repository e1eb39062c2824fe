"""FETCH_SIZE calibration for 8-byte gathers: the standalone hash encode on uniformly random points (no locality
between lanes), to be run under `rocprofv3 --pmc FETCH_SIZE`.  Prints the expected line counts."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ced_nerf_amd import ops, synthetic as S
from ced_nerf_amd.hashgrid import level_tables
dev = "cuda:0"
p = S.init_field_params([-1.5] * 3 + [1.5] * 3, 1e-4, 1024, 21, regime="trained")
table = torch.from_numpy(p["hash"]["table"]).to(dev)
desc, _ = ops.make_hash_desc(table, 16, 1024, 16, 21, False)
n = 1 << 22
x = torch.rand(n, 3, device=dev)
for _ in range(3):
    ops.hash_encode(desc, x)
torch.cuda.synchronize()
tabs = level_tables(16, 1024, 16, 21)
lvl_bytes = [int(s) * 8 for s in tabs["size"]]
print("points", n, "table MB", sum(lvl_bytes) / 1e6, "level MB", [round(b / 1e6, 2) for b in lvl_bytes])
