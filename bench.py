"""Benchmark of the rendering hot path on MI355X (contract: see the task brief / DESIGN.md section Measurement).

One step = one full-frame `render_image_test` (cednerf/utils.py:153-318) of the BASELINE.json
config-2 workload: 800x800 D-NeRF "lego"-shaped synthetic scene, hash L=16 F=2 T=2^21 fp32 table,
64-wide MLPs, "trained-like" parameters, max_samples=1024.  With N GPUs every step renders N such
frames (consecutive camera azimuths of a video render); their rays are dealt tile-cyclically over
the ranks and the pixels are all-gathered over RCCL, so per-GPU work is fixed (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_SAMPLE_F32 = 1072.0     # 16*8*2*4 B table + 28 B in + 20 B out
ALG_BYTES_PER_SAMPLE_F16 = 560.0
ALG_FLOPS_PER_SAMPLE = 38.0e3         # unpadded MLP flops per sample
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--scene", default="dnerf", choices=["dnerf", "hypernerf", "dynerf"])
    ap.add_argument("--table-dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "r01_pmc_field.json"),
                    help="per-launch HBM traffic of the field kernel from the rocprofv3 --pmc passes (tools/pmc_summary.py)")
    ap.add_argument("--regime", default="trained")
    ap.add_argument("--max-samples", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-stride", type=int, default=1, help="pixel stride of the CPU-baseline ray sample")
    ap.add_argument("--cpu-passes", type=int, default=3, help="how many times the CPU baseline renders its sample")
    ap.add_argument("--torch-stride", type=int, default=4,
                    help="pixel stride of the ray sample the PyTorch-fp32 CPU path renders (0 = skip)")
    return ap.parse_args()


def cpu_baseline(sc, args):
    """The CPU oracle (a scalar C port, OpenMP over rays) on a strided sample of the same frame."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))          # before the OpenMP runtime loads
    from oracle import oracle as O
    O.build()
    cfg = sc["cfg"]
    of = O.OracleField(sc["params"])
    oest = O.OracleEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"], sc["binaries"])
    s = args.cpu_stride
    o = np.ascontiguousarray(sc["origins"][::s, ::s]); d = np.ascontiguousarray(sc["viewdirs"][::s, ::s])
    t0 = time.perf_counter()
    for _ in range(args.cpu_passes):
        out = O.render_image_test(args.max_samples, of, oest, o, d, timestamps=sc["timestamps"], **sc["render"])
    dt = (time.perf_counter() - t0) / args.cpu_passes
    cores = int(os.environ["OMP_NUM_THREADS"])
    return {"value": out[3] / dt, "unit": "samples/s", "cores": cores, "kind": "port",
            "rays_per_sec": o.shape[0] * o.shape[1] / dt,
            "sample": f"{args.cpu_passes} passes over every {s}th pixel in x and y of the same "
                      f"{args.width}x{args.height} frame ({o.shape[0] * o.shape[1]} rays, {out[3]} samples, "
                      f"{dt:.1f} s per pass)"}


def cpu_baseline_pytorch(sc, args):
    """The pure-PyTorch fp32 restatement (oracle/torch_oracle.py: torch field + compositing, native
    marching as in the reference) on the host cores, on a strided sample of the same frame."""
    import torch as _t
    from oracle import oracle as O, torch_oracle as TO
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    _t.set_num_threads(min(cores, 64))
    cfg = sc["cfg"]
    tf = TO.TorchField(sc["params"])
    oest = O.OracleEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"], sc["binaries"])
    s = args.torch_stride
    o = np.ascontiguousarray(sc["origins"][::s, ::s]); d = np.ascontiguousarray(sc["viewdirs"][::s, ::s])
    t0 = time.perf_counter()
    out = TO.render_image_test(args.max_samples, tf, oest, o, d, timestamps=sc["timestamps"], **sc["render"])
    dt = time.perf_counter() - t0
    return {"value": out[3] / dt, "unit": "samples/s", "cores": _t.get_num_threads(), "kind": "port (PyTorch fp32)",
            "rays_per_sec": o.shape[0] * o.shape[1] / dt,
            "sample": f"every {s}th pixel in x and y of the same frame ({o.shape[0] * o.shape[1]} rays, {out[3]} samples, "
                      f"{dt:.1f} s)"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from ced_nerf_amd import _lib, synthetic as S
    from ced_nerf_amd import dist as cdist
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays
    _lib.lib()

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # one frame per GPU per step: consecutive azimuths of a turntable video
    tdt = np.float16 if args.table_dtype == "f16" else np.float32
    frames = [S.make_scene(args.scene, args.width, args.height, args.regime, azim_deg=30.0 + 12.0 * f, table_dtype=tdt)
              for f in range(world)]
    sc = frames[0]
    cfg = sc["cfg"]
    field = DNGPradianceField.from_params(sc["params"], dev).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev)
    est.set_binaries(T(sc["binaries"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    origins = torch.stack([T(f["origins"]) for f in frames])        # [F,H,W,3]
    viewdirs = torch.stack([T(f["viewdirs"]) for f in frames])
    ts = T(sc["timestamps"])
    from ced_nerf_amd import ops
    renderer = cdist.ShardedRenderer(field, est, world, rank, dev, max_samples=args.max_samples, render_kwargs=rk)
    renderer.set_rays(origins, viewdirs)
    tracer = ops.FrameTracer(capacity=128, with_events=True)     # HIP events around every field launch
    renderer.tracer = tracer
    field_ms, field_launches, field_samples = [0.0], [0], [0]

    def step():
        out = renderer.render(ts)
        # the frame call returns after its last per-iteration sync, so these events have completed
        ms = tracer.field_ms()
        field_ms[0] += sum(ms); field_launches[0] += len(ms)
        field_samples[0] += sum(it["n_new"] for it in tracer.iterations())
        return out

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    field_ms[0], field_launches[0], field_samples[0] = 0.0, 0, 0
    t0 = time.perf_counter()
    samples_local = 0
    for _ in range(args.steps):
        out = step()
        samples_local += out["local_samples"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    prof = {"field": {"ms": field_ms[0], "launches": field_launches[0], "units": float(field_samples[0])}}
    tt = torch.tensor([dt, float(samples_local)], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); samples_total = float(tsum[1])
    else:
        samples_total = float(samples_local)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    n_rays_step = world * args.width * args.height
    fp16 = sc["params"]["hash"]["table"].dtype == np.float16
    line = {
        "metric": "samples_per_sec (render_image_test, 800x800 D-NeRF lego-shaped synthetic)",
        "value": samples_total / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "rays_per_sec": n_rays_step * args.steps / dt,
        "samples_per_ray": samples_total / (n_rays_step * args.steps),
        "config": {"workload": f"{args.scene} {args.width}x{args.height} render_image_test max_samples={args.max_samples}, "
                               f"hash L=16 F=2 T=2^21 {'fp16' if fp16 else 'fp32'} table, 64-wide MLPs, "
                               f"{args.regime} params, occupancy 128^3 x{cfg['grid_levels']}",
                   "frames_per_step": world, "rays_per_step": n_rays_step,
                   "parallelism": f"rays tile-cyclic over {world} GPU(s) + all-gather of pixels"},
    }
    fk = prof.get("field", None)
    if fk and fk["launches"] > 0:
        avg_ms = fk["ms"] / fk["launches"]
        samples_per_launch = fk["units"] / fk["launches"]
        tflops = samples_per_launch * ALG_FLOPS_PER_SAMPLE / (avg_ms * 1e-3) / 1e12
        gbs = samples_per_launch * (ALG_BYTES_PER_SAMPLE_F16 if fp16 else ALG_BYTES_PER_SAMPLE_F32) / (avg_ms * 1e-3) / 1e9
        traffic = None
        if args.scene == "dnerf" and not fp16 and os.path.exists(args.pmc_json):
            try:        # HBM bytes per launch from the committed PMC passes of this same workload
                pj = json.load(open(args.pmc_json))
                k = [v for n, v in pj.items() if n.startswith("void ced::field_kernel")][0]
                traffic = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
            except Exception:
                traffic = None
        line["roofline"] = {"kernel": "field_kernel (fused DNGPradianceField forward)", "bound": "mfma",
                            "achieved": tflops, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                            "frac": tflops / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
                            "avg_launch_ms": avg_ms, "launches": fk["launches"],
                            "samples_per_launch": samples_per_launch,
                            "field_share_of_step": fk["ms"] / (1e3 * dt)}
        line["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": gbs / PEAK_HBM_GBS, "traffic": traffic}
        line["kernel_ms_per_step"] = {k: v["ms"] / args.steps for k, v in prof.items()}
    if not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(sc, args)
        if args.torch_stride > 0:
            line["cpu_baseline_pytorch"] = cpu_baseline_pytorch(sc, args)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
