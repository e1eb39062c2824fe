"""Benchmark of the rendering hot path on MI355X (contract: see the task brief / DESIGN.md section Measurement).

One frame = one full-frame `render_image_test` (cednerf/utils.py:153-318) of the BASELINE.json
config-2 workload: 800x800 D-NeRF "lego"-shaped synthetic scene, hash L=16 F=2 T=2^21 fp32 table,
64-wide MLPs, "trained-like" parameters, max_samples=1024.  A step renders 3 calls in flight x 16 frames
per call (consecutive camera azimuths of a video render), every frame on its own render_image_test
schedule.  With N GPUs every frame's rays are dealt tile-cyclically over the ranks under the ONE
image-global schedule (survivor counts all-reduced per iteration) and the pixels are all-gathered over
RCCL.  Default WEAK scaling: a call holds 16 x N frames (at most 64), so a rank's launches keep their size as N grows;
--scaling strong keeps the step's frames fixed instead (the other mode is always reported beside the main figure).

Default arithmetic since round 4: f16x2 -- all three MLPs on fp16 MFMA with every operand split into two fp16 numbers --
because it now meets both bars: sample counts / opacity / depth / rgb BIT-IDENTICAL to the CPU oracle's mode of the same
name (the oracle restates the matrix instruction, oracle/mfma_f16_model.h) and within 1e-4 of the plain fp32 oracle
with the same sample counts.
"""
import argparse
import json
import os
import sys
import time

# HIP multiplexes a process's streams onto 4 hardware queues by default, and streams that share a queue serialise.
# A rank drives 3 call streams + the gather stream + the default stream (+ RCCL's own): give every one its own queue (read by
# the HIP runtime when it initialises, i.e. before the first torch.cuda call; measured: 4 lanes lose 15-20 % without it).
# A matter of speed only: since round 4 every collective of a rank is issued by ONE thread on ONE communicator in an order
# that cannot differ between ranks (ced_nerf_amd/dist.py: PipelinedRenderer), so liveness no longer depends on queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALG_BYTES_PER_SAMPLE_F32 = 1072.0     # 16*8*2*4 B table + 28 B in + 20 B out
ALG_BYTES_PER_SAMPLE_F16 = 560.0
ALG_FLOPS_PER_SAMPLE = 38.0e3         # unpadded MLP flops per sample
ALG_FLOPS_SIGMA_CHAIN = 27.0e3        # of which xyz_wrap 20.9 k + mlp_base 6.1 k (what counts / opacity / depth depend on)
ALG_FLOPS_HEAD = 11.0e3               # and mlp_head 11.0 k (feeds rgb only)
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: HBM3E 8 TB/s
PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: f32-input MFMA dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0         # MI355X_MICROARCH.md: f16/bf16 MFMA dense peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--scene", default="dnerf", choices=["dnerf", "hypernerf", "dynerf"])
    ap.add_argument("--table-dtype", default="f32", choices=["f32", "f16"])
    ap.add_argument("--also", default="f32+h16x2,f32,f16",
                    help="comma list of further --mlp-precision modes to time briefly after the main measurement "
                         "(reported under other_mlp_precisions; empty string: none)")
    ap.add_argument("--mlp-precision", default="f16x2", choices=["f32+h16x2", "f32", "f16x2", "f16"],
                    help="arithmetic of the three MLPs (include/cednerf_hip.h CED_MLP_*); every mode is bit-identical to the "
                         "CPU oracle's mode of the same name (tests).  f16x2 (default) = all three on fp16 MFMA, every operand "
                         "split into two fp16 numbers, fp32 accumulate: within 1e-4 of the plain fp32 oracle, same sample "
                         "counts; f32+h16x2 = the chain that decides counts / opacity / depth on exact fp32 MFMA (those three "
                         "bit-identical to the PLAIN oracle), only mlp_head split; f32 = all exact fp32 MFMA; f16 = fp16 "
                         "operands, the reference's tcnn class (BASELINE config 5 with --table-dtype f16)")
    ap.add_argument("--comm-timeout", type=float, default=float(os.environ.get("CED_COMM_TIMEOUT_S", "120")),
                    help="deadline (s) of every wait of the multi-rank exchange and of the process group's collectives: a "
                         "stuck run prints which lane / step / iteration it waits for and exits with status 3")
    ap.add_argument("--pmc-json", default=None,
                    help="per-launch HBM traffic of the field kernel from the rocprofv3 --pmc passes (tools/pmc_summary.py)")
    ap.add_argument("--regime", default="trained")
    ap.add_argument("--max-samples", type=int, default=1024)
    ap.add_argument("--frames-per-call", type=int, default=int(os.environ.get("CED_FRAMES_PER_CALL", "0")),
                    help="frames rendered by one native call (ced_render_frames_test): they share the launches of an "
                         "iteration, each on its own schedule; 1 = ced_render_image_test per frame; 0 (default) = about "
                         "10 M rays per call, 8..32 frames (800x800: 16; 400x400: 32; 1352x1014: 8): the smaller the "
                         "frame, the more of them it takes to fill a launch (400x400: 8 -> 32 frames per call +12 %, "
                         "800x800: 8 -> 16 +1.8 %, profiles/r03_sweep_frames_per_call.txt)")
    ap.add_argument("--frames-in-flight", type=int, default=3,
                    help="independent frames rendered concurrently per GPU (own stream + host thread each)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): per-GPU work fixed -- a call holds frames_per_call x N frames (at most 64), every "
                         "frame's rays dealt tile-cyclically over the N GPUs, so a rank's launches keep their single-GPU size "
                         "as N grows; strong: a step renders frames_in_flight x frames_per_call frames IN TOTAL whatever N "
                         "(value(N) / value(1) is then the speed-up on a fixed job; a rank's launches shrink with N).  Either "
                         "way a unit of a call is one rank's share of ONE frame and every frame runs the image-global "
                         "render_image_test schedule (survivor counts all-reduced per iteration), so the gathered frames are "
                         "bit-identical to single-GPU frames.  With several GPUs the other mode is measured briefly afterwards "
                         "and reported beside the main figure (`other_scaling`)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="after the contractual K-step window, further K-step windows are timed until this much time has "
                         "been measured in all (at least 5, at most 24 windows): median / p10 / p90 in `windows`; "
                         "0: only the contractual window (profiling runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-frame", action="store_true",
                    help="skip the one-frame-alone latency measurement after the timed region (profiling runs: every "
                         "field launch of the process is then one of the timed kind)")
    ap.add_argument("--cpu-stride", type=int, default=1, help="pixel stride of the CPU-baseline ray sample")
    ap.add_argument("--oracle-mode-frames", default="0,23,47",
                    help="timed frames checked bit for bit against the oracle's mode (frame 0 at --oracle-mode-stride, the others "
                         "at twice that stride); profiles/r04_oracle_mode_all_frames.txt holds all 48")
    ap.add_argument("--oracle-mode-stride", type=int, default=2,
                    help="pixel stride of the ray set on which the timed mode is compared BIT FOR BIT with the oracle's mode of "
                         "the same name (parity_vs_oracle_mode; 0 = skip, 1 = the whole frame: minutes of host time)")
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
            "implementation": "C port (oracle/cednerf_oracle.c: scalar C, OpenMP over rays, plain fp32 arithmetic)",
            "rays_per_sec": o.shape[0] * o.shape[1] / dt,
            "sample": f"{args.cpu_passes} passes over every {s}th pixel in x and y of the same "
                      f"{args.width}x{args.height} frame ({o.shape[0] * o.shape[1]} rays, {out[3]} samples, "
                      f"{dt:.1f} s per pass)"}, out


def parity_vs_oracle(oracle_out, gpu_single, gpu_timed_frame0, mode):
    """Frame 0 of the timed set against the pixels the cpu_baseline pass of the oracle computed for the same rays
    (the oracle is the checker here, never the thing measured).  gpu_single: (rgb, opacity, depth, total) of
    render_image_test on that frame alone; gpu_timed_frame0: the same frame as the timed pipeline produced it."""
    w_rgb, w_op, w_dp, w_total = oracle_out[:4]
    g = [t.detach().cpu().numpy() for t in gpu_single[:3]]
    want = [w_rgb, w_op, w_dp]
    diffs = [float(np.abs(a.reshape(b.shape).astype(np.float64) - b).max()) for a, b in zip(g, want)]
    bitexact = all(np.array_equal(a.reshape(b.shape).view(np.uint32), np.ascontiguousarray(b).view(np.uint32))
                   for a, b in zip(g, want))
    mse = float(np.mean((g[0].reshape(w_rgb.shape).astype(np.float64) - w_rgb) ** 2))
    out = {"frame": 0, "mlp_precision": mode, "rgb_max_abs": diffs[0], "opacity_max_abs": diffs[1],
           "depth_max_abs": diffs[2], "samples_gpu": int(gpu_single[3]), "samples_oracle": int(w_total),
           "samples_equal": int(gpu_single[3]) == int(w_total), "bitexact": bool(bitexact),
           "psnr_db": None if mse == 0.0 else float(-10.0 * np.log10(mse))}
    if gpu_timed_frame0 is not None:
        out["timed_frame_equals_single_render"] = all(
            bool(torch.equal(a.reshape(b.shape), b)) for a, b in zip(gpu_timed_frame0, gpu_single[:3]))
    return out


def parity_vs_oracle_mode(sc, args, field, est, rk, ts, T, mode, stride=2):
    """The timed arithmetic mode against the oracle's mode of the SAME name, bit for bit: every `stride`-th pixel in x and
    y of frame 0 rendered as an image of its own (render_image_test's schedule is image-global, so both sides render the
    same strided ray set) by the HIP path and by the oracle (which restates the matrix instruction: minutes of host time
    for a full frame, hence the stride)."""
    from oracle import oracle as O
    from ced_nerf_amd.utils import Rays, render_image_test
    cfg = sc["cfg"]
    of = O.OracleField(sc["params"], mlp_half=mode)
    oest = O.OracleEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"], sc["binaries"])
    o = np.ascontiguousarray(sc["origins"][::stride, ::stride]); d = np.ascontiguousarray(sc["viewdirs"][::stride, ::stride])
    t0 = time.perf_counter()
    want = O.render_image_test(args.max_samples, of, oest, o, d, timestamps=sc["timestamps"], **sc["render"])
    dt = time.perf_counter() - t0
    got = render_image_test(args.max_samples, field, est, Rays(T(o), T(d)), timestamps=ts, **rk)
    torch.cuda.synchronize()
    g = [t.detach().cpu().numpy() for t in got[:3]]
    same = [bool(np.array_equal(np.ascontiguousarray(a).view(np.uint32), np.ascontiguousarray(b.reshape(a.shape)).view(np.uint32)))
            for a, b in zip(g, want[:3])]
    return {"oracle_mode": mode, "rays": int(o.shape[0] * o.shape[1]), "pixel_stride": stride,
            "samples_gpu": int(got[3]), "samples_oracle": int(want[3]), "samples_equal": int(got[3]) == int(want[3]),
            "rgb_bitexact": same[0], "opacity_bitexact": same[1], "depth_bitexact": same[2], "bitexact": all(same),
            "oracle_seconds": dt}


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
    return {"value": out[3] / dt, "unit": "samples/s", "cores": _t.get_num_threads(), "kind": "port",
            "implementation": "PyTorch fp32 (oracle/torch_oracle.py: the pure-PyTorch path north_star names, torch ops on host cores)",
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
    # one rank per GPU; CED_BENCH_BACKEND=gloo is the rehearsal mode (several ranks may then share one card, and
    # the pixel gather is staged through the host: ced_nerf_amd/dist.py)
    backend = os.environ.get("CED_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        import datetime
        pg_timeout = datetime.timedelta(seconds=max(10.0, args.comm_timeout))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=pg_timeout)
        else:
            dist.init_process_group(backend, timeout=pg_timeout)

    from ced_nerf_amd import _lib, synthetic as S
    from ced_nerf_amd import dist as cdist
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays
    _lib.lib()
    if os.environ.get("CED_FIELD_SPREAD_TILES") is not None:
        _lib.check(_lib.lib().ced_set_option(b"field_spread_tiles", int(os.environ["CED_FIELD_SPREAD_TILES"])))

    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    # A step renders `lanes` x `world` frames of a turntable video (consecutive azimuths): every lane is
    # one frame per GPU, its rays dealt tile-cyclically over the ranks; the lanes run concurrently.
    lanes = max(1, args.frames_in_flight)
    if args.frames_per_call <= 0:
        args.frames_per_call = max(8, min(32, int(round(10.24e6 / float(args.width * args.height)))))
    # A unit of a native call is this rank's share of ONE frame in both modes (the frame's loop is the whole image's,
    # cednerf/utils.py:231-235); a call holds at most 64 frames.  strong: frames_per_call frames per call whatever the
    # number of ranks (total work fixed: each rank renders 1/world of every frame); weak: frames_per_call * world frames
    # (per-GPU work fixed: the launches keep their single-GPU size).
    frames_per_call_of = lambda scaling: (max(1, min(64 // world, args.frames_per_call)) * world if scaling == "weak"
                                          else max(1, min(64, args.frames_per_call)))
    per_call = max(1, min(64 // world, args.frames_per_call))        # units of a one-rank call
    fpc_main = frames_per_call_of(args.scaling)
    tdt = np.float16 if args.table_dtype == "f16" else np.float32
    n_frames = lanes * fpc_main
    sc = S.make_scene(args.scene, args.width, args.height, args.regime, azim_deg=30.0, table_dtype=tdt)
    cfg = sc["cfg"]

    def frame_rays(f):            # same scene and field, camera azimuth 30 + 12 f degrees
        if f == 0:
            return {"origins": sc["origins"], "viewdirs": sc["viewdirs"]}
        c2w = S.look_at_c2w(cfg["radius"], 30.0, 30.0 + 12.0 * f, cfg["opengl"])
        o, d = S.make_camera_rays(args.width, args.height, cfg["camera_angle_x"], c2w, cfg["opengl"])
        return {"origins": o, "viewdirs": d}

    n_frames_max = lanes * max(frames_per_call_of("weak"), frames_per_call_of("strong"))
    frames = [frame_rays(f) for f in range(n_frames_max if world > 1 else n_frames)]
    field = DNGPradianceField.from_params(sc["params"], dev, mlp_precision=args.mlp_precision).eval()
    field._descriptor()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev)
    est.set_binaries(T(sc["binaries"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    ts = T(sc["timestamps"])
    from ced_nerf_amd import ops
    # Every call in flight needs one small all-reduce per iteration and one pixel all-gather per call.  All of them go
    # through ONE process group (the default one) and are issued by ONE thread, the one that calls render_steps, in an
    # order fixed by the lanes' own message sequences (dist.PipelinedRenderer): no rank can see another order.
    if world > 1:
        warm = torch.zeros(1, device=dev, dtype=torch.int64)
        dist.all_reduce(warm)                   # the communicator is created here, by all ranks together
        torch.cuda.synchronize()

    def make_lanes(fc_):
        out = []
        for l in range(lanes):
            fr = frames[l * fc_:(l + 1) * fc_]
            r_ = cdist.ShardedRenderer(field, est, world, rank, dev, max_samples=args.max_samples, render_kwargs=rk,
                                       tile_order=os.environ.get("CED_TILE_ORDER", "1") != "0", units=per_call)
            r_.set_rays(torch.stack([T(f["origins"]) for f in fr]), torch.stack([T(f["viewdirs"]) for f in fr]))
            out.append(r_)
        return out

    lane_renderers, tracers = [], []
    for l, r in enumerate(make_lanes(fpc_main)):
        # HIP events around every field launch; one event set per timed step so nothing is read back
        # (hipEventElapsedTime) inside the timed region
        tracers.append([ops.FrameTracer(capacity=96, with_events=True) for _ in range(min(args.steps, 24))])
        if os.environ.get("CED_BENCH_NO_STAMPS", "0") != "1":
            for tr_ in tracers[-1]:
                tr_.enable_device_stamps(dev)
        r.tracer = tracers[-1][0]
        lane_renderers.append(r)
    # multi-rank: the pixel all-gather of one step overlaps the next step's kernels (own stream, no read-back)
    renderer = cdist.PipelinedRenderer(lane_renderers, async_gather=world > 1, comm_timeout_s=args.comm_timeout,
                                       field_max_blocks=int(os.environ.get("CED_FIELD_MAX_BLOCKS", "256")))
    field_ms, field_launches, field_samples = [0.0], [0], [0]
    step_no = [0]

    def step():
        for l, r in enumerate(lane_renderers):
            r.tracer = tracers[l][step_no[0] % len(tracers[l])]
        step_no[0] += 1
        outs = renderer.render(ts)
        return {"local_samples": sum(o["local_samples"] for o in outs)}

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    step_no[0] = 0
    ref_event = torch.cuda.Event(enable_timing=True)
    ref_event.record()
    t0 = time.perf_counter()
    samples_local = 0
    last_row = None
    if os.environ.get("CED_BENCH_JOIN_STEPS", "0") != "0":
        for _ in range(args.steps):
            out = step()
            samples_local += out["local_samples"]
    else:
        # the K steps as a stream: lanes do not wait for each other between steps (PipelinedRenderer.render_steps)
        def before_frame(l, s_):
            # the first min(steps, 24) steps of every lane are traced (the lanes start together, so their traced
            # windows coincide); later steps run untraced
            lane_renderers[l].tracer = tracers[l][s_] if s_ < len(tracers[l]) else None
        rows = renderer.render_steps(ts, args.steps, before_frame=before_frame)
        samples_local = sum(o["local_samples"] for row in rows for o in row)
        last_row = rows[-1]
        del rows
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0

    def reduce_window(dt_w, samples_w):
        tt_w = torch.tensor([dt_w, float(samples_w)], device=dev, dtype=torch.float64)
        if world > 1:
            tmax_w = tt_w.clone(); dist.all_reduce(tmax_w, op=dist.ReduceOp.MAX)
            tsum_w = tt_w.clone(); dist.all_reduce(tsum_w, op=dist.ReduceOp.SUM)
            return float(tmax_w[0]), float(tsum_w[1])
        return dt_w, float(samples_w)

    # further windows of the same K steps (untraced), each bracketed like the first: run-to-run spread of the number
    window_rates = []
    dt_first, samples_first = reduce_window(dt, samples_local)
    window_rates.append(samples_first / dt_first)
    timed = dt_first
    n_extra = 0
    while args.min_seconds > 0 and (timed < args.min_seconds or n_extra < 4) and n_extra < 23:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        rows_w = renderer.render_steps(ts, args.steps, before_frame=lambda l, s_: setattr(lane_renderers[l], "tracer", None))
        s_w = sum(o["local_samples"] for row in rows_w for o in row)
        del rows_w
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        d_w, s_tot = reduce_window(time.perf_counter() - t_w, s_w)
        window_rates.append(s_tot / d_w)
        timed += d_w
        n_extra += 1
    intervals, intervals_dev = [], []
    for l in range(lanes):                      # the last min(steps, 24) steps' field launches
        for tr in tracers[l][:min(args.steps, len(tracers[l]))]:
            iv = tr.field_intervals(ref_event)
            intervals += iv
            if tr.stamps is not None:
                intervals_dev += [x for x in tr.field_intervals_device() if x is not None]
            field_ms[0] += sum(e - b for b, e in iv); field_launches[0] += len(iv)
            field_samples[0] += sum(it["n_new"] for it in tr.iterations())
    # time during which at least one field kernel was executing (frames in flight overlap their launches)
    intervals.sort()
    span_ms = (max(e for _, e in intervals) - intervals[0][0]) if intervals else 0.0      # window the traced launches span
    busy_ms, cur_b, cur_e = 0.0, None, None
    for b, e in intervals:
        if cur_e is None or b > cur_e:
            if cur_e is not None:
                busy_ms += cur_e - cur_b
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy_ms += cur_e - cur_b
    # the same from the kernel's own stamps (first workgroup in -> last workgroup out on the device wall clock): the
    # time a field kernel was EXECUTING; an event pair also holds the launch's wait for CUs that the marching /
    # compositing kernels of the other calls occupy
    busy_dev_ms, raw_dev_ms = None, None
    if intervals_dev:
        intervals_dev.sort()
        busy_dev_ms, cur_b, cur_e = 0.0, None, None
        for b, e in intervals_dev:
            if cur_e is None or b > cur_e:
                if cur_e is not None:
                    busy_dev_ms += cur_e - cur_b
                cur_b, cur_e = b, e
            else:
                cur_e = max(cur_e, e)
        busy_dev_ms += cur_e - cur_b
        raw_dev_ms = sum(e - b for b, e in intervals_dev) / len(intervals_dev)
    prof = {"field": {"ms": field_ms[0], "launches": field_launches[0], "units": float(field_samples[0])}}
    tt = torch.tensor([dt, float(samples_local)], device=dev, dtype=torch.float64)
    if world > 1:
        tmax = tt.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tt.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0]); samples_total = float(tsum[1])
    else:
        samples_total = float(samples_local)
    # the same step in the other MLP arithmetic modes (shorter run; same barrier / max-over-ranks discipline)
    others = {}
    for prec in [p for p in args.also.split(",") if p and p != args.mlp_precision]:
        field.set_mlp_precision(prec)
        field._descriptor()
        step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        k = max(1, min(args.steps, 6))
        t_a = time.perf_counter()
        rows_o = renderer.render_steps(ts, k)
        s_loc = sum(o["local_samples"] for row in rows_o for o in row)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        tt2 = torch.tensor([time.perf_counter() - t_a, float(s_loc)], device=dev, dtype=torch.float64)
        if world > 1:
            t2max = tt2.clone(); dist.all_reduce(t2max, op=dist.ReduceOp.MAX)
            t2sum = tt2.clone(); dist.all_reduce(t2sum, op=dist.ReduceOp.SUM)
            tt2 = torch.stack([t2max[0], t2sum[1]])
        others[prec] = {"value": float(tt2[1]) / float(tt2[0]), "unit": "samples/s", "steps": k,
                        "ms_per_frame": 1e3 * float(tt2[0]) / k / n_frames,
                        "rays_per_sec": n_frames * args.width * args.height * k / float(tt2[0])}
    field.set_mlp_precision(args.mlp_precision)
    field._descriptor()
    # the same step with the hash table stored in the OTHER type (fp16 <-> fp32), briefly: for hash_lookup_hbm_frac
    other_table = None
    if os.environ.get("CED_BENCH_OTHER_TABLE", "1") != "0":
        p2 = dict(sc["params"]); h2 = dict(p2["hash"])
        h2["table"] = np.ascontiguousarray(h2["table"].astype(np.float32 if h2["table"].dtype == np.float16 else np.float16))
        p2["hash"] = h2
        field2 = DNGPradianceField.from_params(p2, dev, mlp_precision=args.mlp_precision).eval()
        field2._descriptor()
        for r_ in lane_renderers:
            r_.field = field2
        step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        k = max(1, min(args.steps, 6))
        t_a = time.perf_counter()
        rows_o = renderer.render_steps(ts, k)
        s_loc = sum(o["local_samples"] for row in rows_o for o in row)
        del rows_o
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        d_o, s_o = reduce_window(time.perf_counter() - t_a, s_loc)
        other_table = {"table": "f32" if h2["table"].dtype == np.float32 else "f16", "value": s_o / d_o, "unit": "samples/s", "steps": k}
        for r_ in lane_renderers:
            r_.field = field
        del field2
    # several GPUs: the same steps in the OTHER scaling mode, briefly (same barrier / max-over-ranks discipline)
    other_scaling = None
    if world > 1:
        other = "weak" if args.scaling == "strong" else "strong"
        fc_o = frames_per_call_of(other)
        lanes_o = make_lanes(fc_o)
        pipe_o = cdist.PipelinedRenderer(lanes_o, async_gather=True, comm_timeout_s=args.comm_timeout,
                                         field_max_blocks=int(os.environ.get("CED_FIELD_MAX_BLOCKS", "256")))
        pipe_o.render(ts)
        pipe_o.wait_gathers()
        dist.barrier()
        torch.cuda.synchronize()
        k = max(1, min(args.steps, 6))
        t_a = time.perf_counter()
        rows_o = pipe_o.render_steps(ts, k)
        s_loc = sum(o["local_samples"] for row in rows_o for o in row)
        del rows_o
        torch.cuda.synchronize()
        dist.barrier()
        d_o, s_o = reduce_window(time.perf_counter() - t_a, s_loc)
        nf_o = lanes * fc_o
        other_scaling = {"scaling": other, "value": s_o / d_o, "unit": "samples/s", "steps": k, "frames_per_step": nf_o,
                         "rays_per_sec": nf_o * args.width * args.height * k / d_o, "ms_per_step": 1e3 * d_o / k}
        del pipe_o, lanes_o
    # latency of ONE frame rendered alone (no other frame in flight; field launches on all CUs again).  With several
    # GPUs the frame is sharded like every other (image-global schedule), so ALL ranks render it together.
    for r_ in lane_renderers:
        r_.field_max_workgroups = 0          # one frame alone: field launches on all CUs
    torch.cuda.synchronize()
    single_field = {"ms": 0.0, "launches": 0, "units": 0.0}
    single_ms, single_stats = None, None
    if not args.no_single_frame:
        alone = cdist.ShardedRenderer(field, est, world, rank, dev, max_samples=args.max_samples, render_kwargs=rk,
                                      tile_order=os.environ.get("CED_TILE_ORDER", "1") != "0")
        alone.set_rays(torch.stack([T(f["origins"]) for f in frames[:1]]), torch.stack([T(f["viewdirs"]) for f in frames[:1]]))
        alone.tracer = tracers[0][0]
        alone.render_local(ts)                          # first call on this stream allocates its workspace
        torch.cuda.synchronize()
        for _ in range(5):                              # with HIP events around the field launches (roofline_single_frame)
            alone.render_local(ts)
            ms = tracers[0][0].field_ms()
            single_field["ms"] += sum(ms); single_field["launches"] += len(ms)
            single_field["units"] += float(sum(it["n_new"] for it in tracers[0][0].iterations()))
        alone.tracer = None
        single_times = []
        for _ in range(20):                             # the latency itself: untraced, every frame timed on its own
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            alone.render_local(ts)
            torch.cuda.synchronize()
            single_times.append((time.perf_counter() - t1) * 1e3)
        single_ms = float(np.median(single_times))
        single_stats = {"median": single_ms, "p10": float(np.quantile(single_times, 0.1)),
                        "p90": float(np.quantile(single_times, 0.9)), "frames": len(single_times)}
    comm_info = None
    if world > 1:
        comm_info = {"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                     "design": "one communicator, one issuing thread per rank: a lane's native call hands every iteration's "
                               "survivor-count row ([frames_per_call] int64) to the thread that called render_steps, which "
                               "takes one message per lane in turn (round robin) and enqueues the all-reduce on that lane's "
                               "stream, and the pixel all-gather when the lane reports its call finished; the order depends "
                               "on the lanes' message sequences only, which the image-global schedule makes identical on "
                               "every rank",
                     "timeout_s": args.comm_timeout,
                     "schedule_allreduces_last_call": int(getattr(lane_renderers[0].exchange, "calls", 0))}
    if rank != 0:
        if world > 1:
            dist.barrier()                  # rank 0's checks below render whole frames alone: nothing collective is left
            dist.destroy_process_group()
        return
    if world > 1:
        dist.barrier()
    n_rays_step = n_frames * args.width * args.height          # frames of a step x rays per frame, over all ranks
    # secondary entry (a2): cednerf.utils.render_image (eval) on frame 0 -- `sampling` with its visibility filter, then
    # `rendering`; the native pass evaluates the field once per sample and stops rays at the filter's threshold
    render_image_entry = None
    if not args.no_single_frame:
        from ced_nerf_amd.utils import Rays, render_image
        rays0 = Rays(T(frames[0]["origins"]), T(frames[0]["viewdirs"]))
        ri_kw = dict(timestamps=ts, **rk)
        out = render_image(field, est, rays0, **ri_kw)
        torch.cuda.synchronize()
        ri_times = []
        for _ in range(10):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            out = render_image(field, est, rays0, **ri_kw)
            torch.cuda.synchronize()
            ri_times.append((time.perf_counter() - t1) * 1e3)
        staged = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ref = render_image(field, est, rays0, native=False, **ri_kw)
            torch.cuda.synchronize()
            staged.append((time.perf_counter() - t1) * 1e3)
        render_image_entry = {
            "ms_per_frame": float(np.median(ri_times)), "p10": float(np.quantile(ri_times, 0.1)),
            "p90": float(np.quantile(ri_times, 0.9)), "frames": len(ri_times), "kept_samples": int(out[3]),
            "staged_ms_per_frame": float(np.median(staged)),
            "equal_to_staged": bool(out[3] == ref[3] and all(torch.equal(out[i], ref[i]) for i in range(3))),
            "note": "cednerf.utils.render_image (eval, utils.py:46-150) on frame 0, one GPU; staged = sampling over every "
                    "marched sample with sigma_fn, then rendering (the reference's composition on the same kernels)"}
    # the viewer's operating point (gui.py:203-237): ONE frame, rays generated on the device from the pose, max_samples = 200,
    # the reference's fp16 networks (mode f16) -- and the timed mode beside it
    gui_entry = None
    if not args.no_single_frame:
        from ced_nerf_amd import cameras
        from ced_nerf_amd.utils import render_image_test as _rit
        focal = 0.5 * args.width / np.tan(0.5 * cfg["camera_angle_x"])
        K_ = np.array([[focal, 0, args.width / 2.0], [0, focal, args.height / 2.0], [0, 0, 1]], np.float32)
        c2w_ = np.ascontiguousarray(S.look_at_c2w(cfg["radius"], 30.0, 30.0, cfg["opengl"]), dtype=np.float32)
        K_t = K_
        gui_entry = {"max_samples": 200, "note": "gui.py:203-237: pose -> rays on the device (ced_generate_rays_pinhole) -> "
                                                  "render_image_test(max_samples=200), one frame at a time, median of 20"}
        for prec in dict.fromkeys(["f16", args.mlp_precision]):
            field.set_mlp_precision(prec)
            field._descriptor()
            tms, tot = [], 0
            for it_ in range(23):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r_ = cameras.pinhole_rays(K_t, c2w_, args.width, args.height, cfg["opengl"], device=dev)
                o_ = _rit(200, field, est, r_, timestamps=ts, **rk)
                torch.cuda.synchronize()
                if it_ >= 3:
                    tms.append((time.perf_counter() - t1) * 1e3)
                tot = int(o_[3])
            gui_entry[prec] = {"ms_per_frame": float(np.median(tms)), "p90": float(np.quantile(tms, 0.9)), "samples": tot,
                               "fps": 1e3 / float(np.median(tms))}
        field.set_mlp_precision(args.mlp_precision)
        field._descriptor()
    fp16 = sc["params"]["hash"]["table"].dtype == np.float16
    line = {
        "metric": "samples_per_sec (render_image_test, 800x800 D-NeRF lego-shaped synthetic)",
        "value": samples_total / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": {"f32": "f32", "f16x2": "f32 (MLP GEMMs: split-fp16 MFMA, 22-bit operands, fp32 accumulate)",
                  "f32+h16x2": "f32 (sigma chain: exact fp32 MFMA; colour head: split-fp16 MFMA, 22-bit operands, fp32 accumulate)",
                  "f16": "f16 MLP operands, fp32 accumulate; rest f32"}[args.mlp_precision],
        "mlp_precision": args.mlp_precision, "data": "synthetic",
        "rays_per_sec": n_rays_step * args.steps / dt,
        "ms_per_frame": 1e3 * dt / args.steps / n_frames,
        "single_frame_latency_ms": single_ms, "single_frame_latency_stats_ms": single_stats,
        "render_image": render_image_entry, "gui_operating_point": gui_entry,
        "windows": {"n": len(window_rates), "steps_each": args.steps, "seconds": timed, "unit": "samples/s",
                    "median": float(np.median(window_rates)), "p10": float(np.quantile(window_rates, 0.1)),
                    "p90": float(np.quantile(window_rates, 0.9)), "first": window_rates[0],
                    "note": "value = the first (contractual) window; the others repeat it untraced"},
        "samples_per_ray": samples_total / (n_rays_step * args.steps),
        "config": {"workload": f"{args.scene} {args.width}x{args.height} render_image_test max_samples={args.max_samples}, "
                               f"hash L=16 F=2 T=2^21 {'fp16' if fp16 else 'fp32'} table, 64-wide MLPs, "
                               f"{args.regime} params, occupancy 128^3 x{cfg['grid_levels']}",
                   "frames_per_step": n_frames, "frames_in_flight_per_gpu": lanes * fpc_main,
                   "frames_per_call": fpc_main, "rays_per_step": n_rays_step,
                   "parallelism": f"{lanes} call(s) in flight per GPU x {fpc_main} frame(s) per call, every frame on the "
                                  f"render_image_test schedule of the WHOLE image; each frame's rays tile-cyclic over "
                                  f"{world} GPU(s) (a unit of a call = one rank's share of one frame), survivor counts "
                                  f"all-reduced per iteration, one all-gather of pixels per call ({args.scaling} scaling)"},
    }
    if other_scaling is not None:
        line["other_scaling"] = other_scaling
    if world > 1:
        # the world size the collective backend itself reports (nccl = RCCL on ROCm), not the flag
        line["rccl_ranks"] = int(dist.get_world_size())
        line["rccl_backend"] = str(dist.get_backend())
    if comm_info is not None:
        line["comm"] = comm_info
    fk = prof.get("field", None)
    if fk and fk["launches"] > 0:
        # With several frames in flight their field launches share the chip, so a launch's own
        # begin->end time includes the others' work.  The effective duration per launch is the time
        # some field kernel was executing divided by the number of launches; the raw per-launch
        # average (what rocprofv3 --stats reports per dispatch) is kept beside it.
        raw_avg_ms = fk["ms"] / fk["launches"]
        avg_events_ms = busy_ms / fk["launches"]
        avg_ms = avg_events_ms
        if busy_dev_ms is not None and len(intervals_dev) == fk["launches"]:
            avg_ms = busy_dev_ms / fk["launches"]           # kernel-executing time (device stamps)
        samples_per_launch = fk["units"] / fk["launches"]
        tflops = samples_per_launch * ALG_FLOPS_PER_SAMPLE / (avg_ms * 1e-3) / 1e12
        gbs = samples_per_launch * (ALG_BYTES_PER_SAMPLE_F16 if fp16 else ALG_BYTES_PER_SAMPLE_F32) / (avg_ms * 1e-3) / 1e9
        traffic, traffic_source = None, None
        exact_kernel = args.mlp_precision in ("f32", "f32+h16x2")
        pmc_json = args.pmc_json
        if pmc_json is None:       # the newest committed PMC passes of this workload (profiles/rNN_final_<mode>_pmc.json)
            import glob
            cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_final_{args.mlp_precision}_pmc.json")))
            pmc_json = cands[-1] if cands else ""
        if args.scene == "dnerf" and not fp16 and os.path.exists(pmc_json):
            try:        # HBM bytes per launch from the committed PMC passes of this same workload
                pj = json.load(open(pmc_json))
                kpat = "void ced::field_kernel" if exact_kernel else "void ced::field_half_kernel"
                k = [v for n, v in pj.items() if n.startswith(kpat)][0]
                traffic = k["fetch_bytes_per_launch"] + k["write_bytes_per_launch"]
                traffic_source = {"file": os.path.relpath(pmc_json, ROOT), "commit": pj.get("_source_commit"),
                                  "note": pj.get("_note")}
            except Exception:
                traffic = None
        # Which roofline bounds the field kernel depends on the MLP arithmetic.  f32: the exact fp32 MFMA chain is
        # matrix-bound (fp32 MFMA peak).  f16x2 / f16: the GEMMs shrink to a few % of the f16 MFMA peak and the
        # kernel is bound by the hash lookup (SURVEY 8d algorithmic bytes per sample against HBM peak -- the
        # north-star's "HBM roofline on the hash lookup"); the other bound is reported beside it.
        exact = args.mlp_precision in ("f32", "f32+h16x2")
        mixed = args.mlp_precision == "f32+h16x2"
        mfma_peak = PEAK_F32_MFMA_TFLOPS if exact else PEAK_F16_MFMA_TFLOPS
        alg_bytes = ALG_BYTES_PER_SAMPLE_F16 if fp16 else ALG_BYTES_PER_SAMPLE_F32
        kname = "field_kernel" if exact else "field_half_kernel"
        common = {"traffic_source": traffic_source,
                  "avg_launch_ms": avg_ms, "launches": fk["launches"], "samples_per_launch": samples_per_launch,
                  "avg_launch_ms_raw": raw_dev_ms if raw_dev_ms is not None else raw_avg_ms,
                  "timing": "device stamps" if avg_ms is not avg_events_ms else "hip events",
                  "hip_events": {"avg_launch_ms": avg_events_ms, "avg_launch_ms_raw": raw_avg_ms,
                                 "frac": (samples_per_launch * ((ALG_FLOPS_SIGMA_CHAIN / PEAK_F32_MFMA_TFLOPS + ALG_FLOPS_HEAD / PEAK_F16_MFMA_TFLOPS) / 1e12
                                                                if args.mlp_precision == "f32+h16x2" else
                                                                ALG_FLOPS_PER_SAMPLE / 1e12 / PEAK_F32_MFMA_TFLOPS if args.mlp_precision == "f32"
                                                                else (ALG_BYTES_PER_SAMPLE_F16 if fp16 else ALG_BYTES_PER_SAMPLE_F32) / 1e9 / PEAK_HBM_GBS)
                                          / (avg_events_ms * 1e-3))},
                  "field_busy_over_wall": (busy_dev_ms if busy_dev_ms is not None else busy_ms) / max(span_ms, 1e-9),
                  "note": "%d call(s) in flight x %d frame(s) per call: avg_launch_ms = (time with a field kernel EXECUTING) "
                          "/ launches, from the kernel's own stamps (first workgroup in -> last workgroup out, device wall "
                          "clock); avg_launch_ms_raw = mean of those intervals per launch (overlapping launches share the "
                          "chip; this is what rocprofv3 --stats lists per dispatch); hip_events = the same two figures from "
                          "the HIP event pairs recorded around every launch on its stream, which also count the launch's "
                          "wait for CUs held by the other calls' marching / compositing kernels; roofline_single_frame = the "
                          "kernel with one frame alone (events)" % (lanes, per_call)}
        r_mfma = {"kernel": f"{kname} (fused DNGPradianceField forward, mlp_precision={args.mlp_precision})",
                  "bound": "mfma", "achieved": tflops, "peak": mfma_peak, "unit": "TFLOP/s", "frac": tflops / mfma_peak,
                  "traffic": traffic, "alg_flops_per_sample": ALG_FLOPS_PER_SAMPLE}
        r_hbm = {"kernel": r_mfma["kernel"], "bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                 "frac": gbs / PEAK_HBM_GBS, "traffic": traffic, "alg_bytes_per_sample": alg_bytes}
        # uncontended figure: the same kernel while only ONE frame is in flight (measured after the timed region)
        if single_field is not None and single_field["launches"] > 0:
            s_ms = single_field["ms"] / single_field["launches"]
            s_spl = single_field["units"] / single_field["launches"]
            s_tf = s_spl * (ALG_FLOPS_SIGMA_CHAIN if mixed else ALG_FLOPS_PER_SAMPLE) / (s_ms * 1e-3) / 1e12
            s_tf16 = s_spl * ALG_FLOPS_HEAD / (s_ms * 1e-3) / 1e12 if mixed else 0.0
            s_gbs = s_spl * alg_bytes / (s_ms * 1e-3) / 1e9
            line["roofline_single_frame"] = (
                {"bound": "mfma", "achieved": s_tf, "peak": mfma_peak, "unit": "TFLOP/s",
                 "frac": s_tf / mfma_peak + s_tf16 / PEAK_F16_MFMA_TFLOPS} if exact else
                {"bound": "hbm", "achieved": s_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": s_gbs / PEAK_HBM_GBS})
            line["roofline_single_frame"].update({"avg_launch_ms": s_ms, "launches": single_field["launches"]})
        if mixed:
            # FLOPs by MFMA class: the sigma chain on the fp32-input MFMA (157.3 TF dense), the colour head on the fp16
            # MFMA (2.5 PF dense; three fp16 product blocks per algorithmic one).  `frac` = the share of the launch
            # time the two matrix pipes need at their peaks -- f16 FLOPs are never priced at the fp32 peak.
            tf32 = samples_per_launch * ALG_FLOPS_SIGMA_CHAIN / (avg_ms * 1e-3) / 1e12
            tf16 = samples_per_launch * ALG_FLOPS_HEAD / (avg_ms * 1e-3) / 1e12         # ALGORITHMIC flops (the split issues 3x)
            r_mfma.update({"achieved": tf32, "peak": PEAK_F32_MFMA_TFLOPS, "frac": tf32 / PEAK_F32_MFMA_TFLOPS + tf16 / PEAK_F16_MFMA_TFLOPS,
                           "alg_flops_per_sample": ALG_FLOPS_SIGMA_CHAIN,
                           "mfma_classes": {
                               "f32": {"alg_flops_per_sample": ALG_FLOPS_SIGMA_CHAIN, "achieved": tf32, "peak": PEAK_F32_MFMA_TFLOPS,
                                       "unit": "TFLOP/s", "frac": tf32 / PEAK_F32_MFMA_TFLOPS},
                               "f16": {"alg_flops_per_sample": ALG_FLOPS_HEAD, "issued_flops_per_sample": 3.0 * ALG_FLOPS_HEAD,
                                       "achieved": tf16, "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                       "frac": tf16 / PEAK_F16_MFMA_TFLOPS,
                                       "note": "achieved / frac count the ALGORITHMIC flops; the split form issues three times as many"}},
                           "hash_lookup_hbm_frac": samples_per_launch * 1024.0 / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS})
        # SURVEY 8d's primary figure: the hash lookup alone (L * 8 corners * F features * sizeof(feature) = 1024 B per sample
        # from an fp32 table, 512 B from an fp16 one) against HBM peak, for BOTH table types: this run's table from the
        # field kernel's executing time, the other table type from a short run of the same pipeline (`other_table`)
        hl_b = 512.0 if fp16 else 1024.0
        hash_lookup = {("f16_table_512B" if fp16 else "f32_table_1024B"): {
            "bytes_per_sample": hl_b, "frac_kernel_time": samples_per_launch * hl_b / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
            "frac_whole_pipeline": samples_total / dt * hl_b / 1e9 / PEAK_HBM_GBS / world}}
        if other_table is not None:
            ob = 1024.0 if fp16 else 512.0
            hash_lookup["f32_table_1024B" if fp16 else "f16_table_512B"] = {
                "bytes_per_sample": ob, "samples_per_sec": other_table["value"],
                "frac_whole_pipeline": other_table["value"] * ob / 1e9 / PEAK_HBM_GBS / world,
                "note": "short run (%d steps) of the same pipeline with the table stored in the other type: the bytes halve "
                        "or double, the time does not (the gathers are served from L2 / Infinity Cache)" % other_table["steps"]}
        r_hbm["hash_lookup_hbm_frac"] = hash_lookup
        r_mfma["hash_lookup_hbm_frac"] = hash_lookup
        line["roofline"] = dict(r_mfma if exact else r_hbm, **common)
        line["roofline_hbm" if exact else "roofline_mfma"] = r_hbm if exact else r_mfma
        line["kernel_ms_per_step"] = {k: v["ms"] / min(args.steps, 24) for k, v in prof.items()}
    if others:
        notes = {"f32+h16x2": "sigma chain on exact fp32 MFMA (counts / opacity / depth bit-identical to the PLAIN oracle), colour "
                              "head on split-fp16 MFMA: rgb <= 1e-4, and bit-identical to the oracle's f32+h16x2 mode",
                 "f16x2": "split-fp16 MFMA MLPs, fp32 accumulate: bit-identical to the oracle's f16x2 mode; within 1e-4 of the "
                          "plain fp32 oracle",
                 "f16": "fp16-operand MLPs (the reference's tcnn class; BASELINE config 5 with --table-dtype f16): "
                        "bit-identical to the oracle's f16 mode (tests/test_gpu_fullframe.py C5)",
                 "f32": "exact fp32 MFMA chain: bit-identical to the plain oracle"}
        for k_, v_ in others.items():
            v_["parity"] = notes[k_]
        line["other_mlp_precisions"] = others
    if world > 1 and last_row is not None:
        # the exchange, checked: lane 0's gathered frames of the last timed step (every rank's shard on the image-global
        # schedule, all-gathered and un-permuted) against the same frames rendered whole by this rank alone: the same
        # bits and the same sample totals, not a tolerance
        from ced_nerf_amd.utils import render_image_test
        renderer.wait_gathers()
        torch.cuda.synchronize()
        o0 = last_row[0]
        worst = {"rgb": 0.0, "opacity": 0.0, "depth": 0.0}
        single_total, n_over, n_pix, equal = 0, 0, 0, True
        n_check = min(o0["rgb"].shape[0], 3)
        for k in range(n_check):
            fk_ = frames[k]
            single = render_image_test(args.max_samples, field, est, Rays(T(fk_["origins"]), T(fk_["viewdirs"])),
                                       timestamps=ts, **rk)
            single_total += int(single[3])
            over = None
            for nm, a, b in (("rgb", o0["rgb"][k], single[0]), ("opacity", o0["opacity"][k], single[1]),
                             ("depth", o0["depth"][k], single[2])):
                equal = equal and bool(torch.equal(a, b.reshape(a.shape)))
                dlt = (a - b.reshape(a.shape)).abs().amax(dim=-1)
                worst[nm] = max(worst[nm], float(dlt.max()))
                over = (dlt > 1e-4) if over is None else (over | (dlt > 1e-4))
            n_over += int(over.sum()); n_pix += int(over.numel())
        # sample total of the gathered call (all ranks, all its frames) against single-rank renders of ALL its frames
        gathered_total = int(o0["total_samples_tensor"].item()) if o0.get("total_samples_tensor") is not None else None
        for k in range(n_check, o0["rgb"].shape[0]):
            fk_ = frames[k]
            single_total += int(render_image_test(args.max_samples, field, est, Rays(T(fk_["origins"]), T(fk_["viewdirs"])),
                                                  timestamps=ts, **rk)[3])
        line["gather_check"] = {"frames": n_check, "rgb_max_abs": worst["rgb"], "opacity_max_abs": worst["opacity"],
                                "depth_max_abs": worst["depth"], "samples_single_rank": single_total,
                                "samples_gathered": gathered_total, "samples_equal": gathered_total == single_total,
                                "pixels": n_pix, "pixels_over_1e-4": n_over, "bitexact": bool(equal),
                                "ok": bool(equal and n_over == 0 and gathered_total == single_total)}
    if not args.no_cpu_baseline:
        line["cpu_baseline"], oracle_out = cpu_baseline(sc, args)
        if world == 1 and args.cpu_stride == 1:
            # parity of the benchmarked workload: frame 0 (the scene's own camera) rendered alone by the HIP path in
            # the timed arithmetic mode, against the oracle's pixels of the same rays; and the frame as the timed
            # pipeline (frames in flight, several frames per call) produced it against that single render
            from ced_nerf_amd.utils import render_image_test
            single = render_image_test(args.max_samples, field, est, Rays(T(sc["origins"]), T(sc["viewdirs"])),
                                       timestamps=ts, **rk)
            timed0 = None
            if last_row is not None:
                o0 = last_row[0]
                timed0 = (o0["rgb"][0], o0["opacity"][0], o0["depth"][0])
            torch.cuda.synchronize()
            line["parity_vs_oracle"] = parity_vs_oracle(oracle_out, single, timed0, args.mlp_precision)
            line["parity_vs_oracle"]["oracle"] = "plain fp32 (the reference arithmetic); north-star: counts equal, pixels <= 1e-4"
            # every TIMED frame, not only frame 0: its sample total in the timed mode against the total of the exact fp32
            # mode (itself bit-identical to the plain oracle: tests/test_gpu_fullframe.py), and the timed pipeline's per-call
            # totals against the sums of its frames rendered alone
            if args.mlp_precision != "f32":
                tot_mode, tot_exact, worst = [], [], {"rgb": 0.0, "opacity": 0.0, "depth": 0.0}
                n_over, n_pix = 0, 0
                for fr in frames[:n_frames]:
                    rr = Rays(T(fr["origins"]), T(fr["viewdirs"]))
                    field.set_mlp_precision(args.mlp_precision)
                    a_ = render_image_test(args.max_samples, field, est, rr, timestamps=ts, **rk)
                    field.set_mlp_precision("f32")
                    b_ = render_image_test(args.max_samples, field, est, rr, timestamps=ts, **rk)
                    tot_mode.append(int(a_[3])); tot_exact.append(int(b_[3]))
                    over = None
                    for nm, x_, y_ in (("rgb", a_[0], b_[0]), ("opacity", a_[1], b_[1]), ("depth", a_[2], b_[2])):
                        dlt = (x_ - y_).abs().amax(dim=-1)
                        worst[nm] = max(worst[nm], float(dlt.max()))
                        over = (dlt > 1e-4) if over is None else (over | (dlt > 1e-4))
                    n_over += int(over.sum()); n_pix += int(over.numel())
                field.set_mlp_precision(args.mlp_precision)
                calls_ok = None
                if last_row is not None:
                    calls_ok = all(int(o_["total_samples"]) == sum(tot_mode[l * fpc_main:(l + 1) * fpc_main])
                                   for l, o_ in enumerate(last_row))
                line["parity_vs_oracle"]["timed_frames"] = {
                    "frames": len(tot_mode), "samples_equal_frames": int(sum(a == b for a, b in zip(tot_mode, tot_exact))),
                    "samples_equal": tot_mode == tot_exact, "max_abs_difference": int(max(abs(a - b) for a, b in zip(tot_mode, tot_exact))),
                    "max_rel_difference": float(max(abs(a - b) / b for a, b in zip(tot_mode, tot_exact))),
                    "rgb_max_abs": worst["rgb"], "opacity_max_abs": worst["opacity"], "depth_max_abs": worst["depth"],
                    "pixels": n_pix, "pixels_over_1e-4": n_over, "pixels_within_1e-4": bool(n_over == 0),
                    "note_pixels": "a pixel can differ by about early_stop_eps = 1e-4 itself when its ray's early-stop test "
                                   "(T < 1e-4, cednerf/utils.py:301-306) falls on the other side in the two arithmetics: the ray then "
                                   "marches one batch more or less, which changes opacity by at most its remaining transmittance",
                    "timed_calls_equal_sum_of_single_renders": calls_ok,
                    "against": "every timed frame rendered alone in the timed mode vs in the exact fp32 mode (GPU), which the tests "
                               "tie bit for bit to the plain oracle.  Counts: a ray whose transmittance lands within rounding of the "
                               "early-stop threshold may march one batch more or less in a 22-bit arithmetic than in fp32; against "
                               "the oracle's mode of the SAME arithmetic the counts are exact (parity_vs_oracle_mode)"}
            if args.mlp_precision != "f32" and args.oracle_mode_stride > 0:
                line["parity_vs_oracle_mode"] = parity_vs_oracle_mode(sc, args, field, est, rk, ts, T, args.mlp_precision,
                                                                      args.oracle_mode_stride)
                # and further timed frames (default: the middle and the last one) on a coarser ray set
                extra = []
                for fi in sorted({int(x) for x in args.oracle_mode_frames.split(",") if x.strip()} - {0}):
                    if 0 <= fi < n_frames:
                        sc_f = dict(sc, origins=frames[fi]["origins"], viewdirs=frames[fi]["viewdirs"])
                        e_ = parity_vs_oracle_mode(sc_f, args, field, est, rk, ts, T, args.mlp_precision, 2 * args.oracle_mode_stride)
                        e_["frame"] = fi
                        extra.append(e_)
                line["parity_vs_oracle_mode"]["frame"] = 0
                line["parity_vs_oracle_mode"]["more_frames"] = extra
                line["parity_vs_oracle_mode"]["all_bitexact"] = bool(line["parity_vs_oracle_mode"]["bitexact"] and all(e_["bitexact"] for e_ in extra))
            # the other arithmetic modes of `other_mlp_precisions`, same frame, same (fp32) oracle pixels: PSNR / max error
            for prec in [p for p in args.also.split(",") if p and p != args.mlp_precision]:
                field.set_mlp_precision(prec)
                alt = render_image_test(args.max_samples, field, est, Rays(T(sc["origins"]), T(sc["viewdirs"])),
                                        timestamps=ts, **rk)
                torch.cuda.synchronize()
                line.setdefault("other_mlp_precisions", {}).setdefault(prec, {})["parity_vs_oracle"] = parity_vs_oracle(
                    oracle_out, alt, None, prec)
            field.set_mlp_precision(args.mlp_precision)
        if args.torch_stride > 0:
            line["cpu_baseline_pytorch"] = cpu_baseline_pytorch(sc, args)
    print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception as e:      # noqa: BLE001
        from ced_nerf_amd.dist import ExchangeTimeout
        if not isinstance(e, ExchangeTimeout):
            raise
        # a stuck exchange: say where, and leave with a status -- lanes may still sit in native calls, so no clean-up, no
        # interpreter shutdown that would join them, and never a re-exec of a process that has touched the GPU
        sys.stderr.write(f"bench.py: rank {os.environ.get('RANK', '0')}: {e}\n")
        sys.stderr.flush()
        os._exit(3)
