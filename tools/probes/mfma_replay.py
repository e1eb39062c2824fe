#!/usr/bin/env python3
"""Replays, on the hardware, every block of eight products the oracle evaluates for chosen samples of the field.

    python tools/probes/mfma_replay.py <mismatch.npz>      # on the MI355X: needs tools/probes/mfma_f16_order (built)

For each sample listed in mismatch.npz (tools/debug_half_mismatch.py) the oracle's fp16-operand mode is re-run on its
slow path with the block recorder on; every recorded (accumulator, a[8], b[8]) goes through ONE v_mfma_f32_16x16x16_f16
(diagonal of a probe tile, k = 8..15 zero) and the hardware's result is compared with the model's.  A difference here is
a case the model (oracle/mfma_f16_model.h) gets wrong; none means the kernel and the oracle differ OUTSIDE the matrix
instruction.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from ced_nerf_amd import synthetic as S  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = [dict(), dict(use_div_offsets=True), dict(use_time_embedding=True),
         dict(use_time_embedding=True, use_time_attenuation=True, use_div_offsets=True),
         dict(table_dtype=np.float16), dict(temporal_hash=True, table_dtype=np.float16, use_time_embedding=True)]


class Rec(C.Structure):
    _fields_ = [("acc_in", C.c_float), ("n", C.c_int32), ("a", C.c_float * 8), ("b", C.c_float * 8), ("out", C.c_float)]


def record(p, prec, pos, t, d):
    of = O.OracleField(p, mlp_half=prec, prepare=False)
    cap = 200000
    buf = (Rec * cap)()
    O.lib().ced_o_record_blocks(buf, C.c_int64(cap))
    of.forward(pos[None], t[None], d[None], want_geo=True)
    n = O.lib().ced_o_recorded_blocks()
    O.lib().ced_o_record_blocks(None, C.c_int64(0))
    return [buf[i] for i in range(n)]


def main(path):
    O.build()
    O.lib().ced_o_recorded_blocks.restype = C.c_int64
    mm = np.load(path)
    recs = []
    for key in [k[:-4] for k in mm.files if k.endswith("_idx")]:
        ci, regime, prec = key.split("_", 2)
        ci = int(ci[1:])
        p = S.init_field_params([-1.5] * 3 + [1.5] * 3, 1.0 / 64 if regime == "trained" else 1e-4, 1024, 17, regime=regime,
                                seed=7 + ci, **CASES[ci])
        rng = np.random.default_rng(11)
        n = 20000 + 37
        pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)
        t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32); t[2] = 0; t[3] = 1
        d = rng.normal(size=(n, 3)).astype(np.float32)
        for i in mm[key + "_idx"]:
            r = record(p, prec, pos[i], t[i], d[i])
            print(f"{key} sample {i}: {len(r)} blocks recorded")
            recs += [(key, int(i), q, x) for q, x in enumerate(r)]
    nt = (len(recs) + 15) // 16
    A = np.zeros((nt, 16, 32), np.float16); B = np.zeros((nt, 32, 16), np.float16); Cc = np.zeros((nt, 16, 16), np.float32)
    for k, (_, _, _, x) in enumerate(recs):
        tt, i = divmod(k, 16)
        A[tt, i, :8] = np.array(x.a[:], np.float32).astype(np.float16)
        B[tt, :8, i] = np.array(x.b[:], np.float32).astype(np.float16)
        Cc[tt, i, i] = x.acc_in
    with open("/tmp/replay_in.bin", "wb") as f:
        f.write(np.int32(nt).tobytes())
        for tt in range(nt):
            f.write(A[tt].view(np.uint16).tobytes()); f.write(B[tt].view(np.uint16).tobytes()); f.write(Cc[tt].tobytes())
    subprocess.check_call([os.path.join(HERE, "mfma_f16_order"), "/tmp/replay_in.bin", "/tmp/replay_out.bin"])
    raw = np.fromfile("/tmp/replay_out.bin", dtype=np.uint8)
    D = raw[4:].view(np.float32).reshape(nt, 4, 16, 16)
    bad = 0
    for k, (key, si, q, x) in enumerate(recs):
        tt, i = divmod(k, 16)
        hw = D[tt, 0, i, i]
        if np.float32(hw).view(np.uint32) != np.float32(x.out).view(np.uint32) and not (hw == 0 and x.out == 0):
            bad += 1
            print(f"MODEL != HARDWARE: {key} sample {si} block {q}: acc_in {x.acc_in!r} hw {hw!r} model {x.out!r}")
            print("    a", [float(v).hex() for v in x.a[:x.n]])
            print("    b", [float(v).hex() for v in x.b[:x.n]])
            print("    acc_in", float(x.acc_in).hex(), "hw", float(hw).hex(), "model", float(x.out).hex())
    print(f"{len(recs)} blocks replayed, {bad} differ between the hardware and the model")


if __name__ == "__main__":
    main(sys.argv[1])
