"""Host-side cost of the oracle's field in its arithmetic modes (test infrastructure): how long the parity tests wait."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from ced_nerf_amd import synthetic as S
O.build()
p = S.init_field_params([-1.5]*3+[1.5]*3, 1/64, 1024, 17, regime="trained", seed=7)
rng=np.random.default_rng(1)
n=40000
pos=rng.uniform(-1.6,1.6,size=(n,3)).astype(np.float32); t=rng.uniform(0,1,size=(n,1)).astype(np.float32); d=rng.normal(size=(n,3)).astype(np.float32)
for mode in ("f32","f16","f16x2"):
    of=O.OracleField(p,mlp_half=mode)
    of.forward(pos[:100],t[:100],d[:100])
    t0=time.time(); of.forward(pos,t,d); dt=time.time()-t0
    print(mode, '%.2f s for %d samples, OMP_NUM_THREADS=%s' % (dt, n, os.environ.get('OMP_NUM_THREADS', 'unset')))
