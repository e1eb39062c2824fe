"""Ray-parallel multi-GPU rendering: one process per GPU, rays dealt tile-cyclically, one
all-gather of the rendered pixels per step (RCCL over xGMI when the backend is "nccl").

The reference is single-GPU (train_real.py:81); SURVEY.md section 8e defines this layer.  Rays are
independent, the field and the occupancy grid are replicated, so the only exchange step is the
gather of [n_local, 5] float32 pixels (rgb, opacity, depth); the per-rank sample count rides in
one extra row of the same payload, so a frame costs exactly one collective.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.distributed as dist

TILE = 8    # pixels; 8x8 tiles dealt round-robin balance the spatially clustered sample density


_assignment_cache: Dict = {}


def tile_cyclic_assignment(n_frames: int, height: int, width: int, world: int, tile: int = TILE):
    """Returns (owner[n_rays] int32, per-rank lists of flat ray ids in tile-raster order).  Cached per shape: a video
    re-deals every frame's rays the same way."""
    key = (n_frames, height, width, world, tile)
    if key not in _assignment_cache:
        if len(_assignment_cache) > 16:
            _assignment_cache.clear()
        _assignment_cache[key] = _tile_cyclic_assignment(n_frames, height, width, world, tile)
    return _assignment_cache[key]


def _tile_cyclic_assignment(n_frames: int, height: int, width: int, world: int, tile: int):
    ty = (np.arange(height) // tile)[:, None]
    tx = (np.arange(width) // tile)[None, :]
    tiles_x = (width + tile - 1) // tile
    tiles_y = (height + tile - 1) // tile
    tile_id = ty * tiles_x + tx                                         # [H,W]
    ids = (np.arange(n_frames)[:, None, None] * (tiles_x * tiles_y) + tile_id[None]).reshape(-1)
    owner = (ids % world).astype(np.int32)
    order = np.argsort(ids, kind="stable")                              # tile-major ray order
    shards = [order[owner[order] == r] for r in range(world)]
    return owner, shards


class ShardedRenderer:
    """Renders a batch of frames [F,H,W] with the rays sharded over `world` ranks.

    render_fn(rays_o [n,3], rays_d [n,3], timestamps) -> (rgb [n,3], opacity [n,1], depth [n,1], n_samples)
    defaults to the HIP `render_image_test`; tests inject a CPU renderer to exercise the
    sharding / gather / un-permute logic over gloo.

    Several ranks: a UNIT of the native call is this rank's share of ONE frame, and every frame runs the loop of the
    whole image -- `N_samples = N_rays // N_alive` (cednerf/utils.py:231-235) over all ranks' rays, the per-iteration
    survivor counts all-reduced on the rendering stream (ops.ScheduleExchange on `schedule_group`) -- so the gathered
    frames are bit-identical to the frames one rank renders alone.  The local rays are laid out [F, n_unit, 3]: frame
    f's share padded to the common size `n_unit` (`local_real[f]` real rays; padding is never alive and its pixels are
    dropped by the gather)."""

    def __init__(self, field, estimator, world: int, rank: int, device, max_samples: int = 1024,
                 render_kwargs: Optional[Dict] = None, render_fn: Optional[Callable] = None,
                 force_collective: bool = False, tile_order: bool = False, units: int = 1, schedule_group=None):
        self.field, self.estimator = field, estimator
        self.world, self.rank, self.device = world, rank, device
        self.max_samples = max_samples
        self.render_kwargs = dict(render_kwargs or {})
        self.render_fn = render_fn or self._hip_render
        self.shape = None
        self.tracer = None             # optional ops.FrameTracer handed to the native frame call
        self.field_max_workgroups = 0  # workgroups of this renderer's field launches (0 = one per CU); PipelinedRenderer sets it
        self.field_stream = None       # optional stream shared with other in-flight frames (PipelinedRenderer)
        self.force_collective = force_collective   # run the shard/gather/un-permute path even when world == 1
        # world == 1 without a collective: still walk the rays in 8x8-tile order (a wave's 64 rays are then one
        # tile, not a 64-pixel strip: more coherent marching depths and hash cells) and un-permute the pixels
        self.tile_order = tile_order
        self.unpermute = None
        # ONE rank, units > 1: the F frames handed to set_rays are `units` groups of F // units consecutive frames, and
        # one native call (ced_render_frames_test) renders all the groups through shared launches while every group
        # keeps its own render_image_test schedule.  With F == units a group is a frame, so every frame is rendered
        # exactly as if alone -- only with `units` times larger launches.  (Several ranks: a unit is always one frame.)
        self.units = int(units)
        assert 1 <= self.units <= 64, "units must be 1..64"
        # the process group whose ranks share the frames (None = the default group); a renderer that runs concurrently
        # with others (a lane of PipelinedRenderer) needs a group of its own: collectives of one communicator must be
        # issued in the same order on every rank, and the lanes' threads issue theirs independently
        self.schedule_group = schedule_group
        self.exchange = None           # ops.ScheduleExchange of the current image shape
        self.exchange_issuer = None    # PipelinedRenderer: who issues this renderer's collectives (ops.ScheduleExchange.issuer)
        self.sharded = False

    def _hip_render(self, rays_o, rays_d, timestamps):
        from .utils import Rays, render_frames_test, render_image_test
        if self.sharded:
            F = self.shape[0]
            ts = timestamps.reshape(-1).float()
            ts = ts.expand(F).contiguous() if ts.numel() == 1 else ts      # one time for all frames, or one per frame
            if self.exchange is None:
                from . import ops
                H, W = self.shape[1], self.shape[2]
                self.exchange = ops.ScheduleExchange(F, H * W, self.local_real, rays_o.device, self.max_samples,
                                                     float(self.render_kwargs.get("cone_angle", 0.0)),
                                                     group=self.schedule_group)
            self.exchange.issuer = self.exchange_issuer
            rgb, op, dp, totals = render_frames_test(
                self.max_samples, self.field, self.estimator, Rays(rays_o.view(F, -1, 3), rays_d.view(F, -1, 3)), timestamps=ts,
                tracer=self.tracer, field_max_workgroups=self.field_max_workgroups, exchange=self.exchange,
                **self.render_kwargs)
            return rgb.view(-1, 3), op.view(-1, 1), dp.view(-1, 1), sum(totals)
        if self.units == 1:
            return render_image_test(self.max_samples, self.field, self.estimator, Rays(rays_o, rays_d),
                                     timestamps=timestamps, tracer=self.tracer, field_stream=self.field_stream,
                                     field_max_workgroups=self.field_max_workgroups, **self.render_kwargs)
        u = self.units
        ts = timestamps.reshape(-1).float()
        ts = ts.expand(u).contiguous() if ts.numel() == 1 else ts      # one time for all groups, or one per group
        rgb, op, dp, totals = render_frames_test(
            self.max_samples, self.field, self.estimator, Rays(rays_o.view(u, -1, 3), rays_d.view(u, -1, 3)), timestamps=ts,
            tracer=self.tracer, field_stream=self.field_stream, field_max_workgroups=self.field_max_workgroups,
            **self.render_kwargs)
        return rgb.view(-1, 3), op.view(-1, 1), dp.view(-1, 1), sum(totals)

    def set_rays(self, origins: torch.Tensor, viewdirs: torch.Tensor) -> None:
        assert origins.ndim == 4 and origins.shape == viewdirs.shape, "rays must be [F,H,W,3]"
        F, H, W, _ = origins.shape
        o = origins.reshape(-1, 3); d = viewdirs.reshape(-1, 3)
        if self.shape == (F, H, W) and getattr(self, "_ray_index", None) is not None and self._ray_index.device == o.device:
            # same image shape as last time (the next frame of a video): the index tensors are already on the device
            self.local_o, self.local_d = o[self._ray_index].contiguous(), d[self._ray_index].contiguous()
            return
        self.shape = (F, H, W)
        self._ray_index = None
        self.exchange = None
        self.sharded = self.world > 1 or self.force_collective
        if not self.sharded:
            if self.units > 1 and F % self.units:
                raise ValueError(f"{F} frames do not split into {self.units} equal groups")
            self.n_local = self.n_pad = o.shape[0]
            self.gather_index = None
            if self.tile_order:
                _, shards = tile_cyclic_assignment(F, H, W, 1)
                order = torch.from_numpy(shards[0].astype(np.int64)).to(o.device)        # tile-major -> flat ray id
                self.local_o, self.local_d = o[order].contiguous(), d[order].contiguous()
                inv = torch.empty_like(order)
                inv[order] = torch.arange(order.numel(), device=o.device)
                self.unpermute = inv
                self._ray_index = order
            else:
                self.local_o, self.local_d = o.contiguous(), d.contiguous()
                self.unpermute = None
            return
        if F > 64:
            raise ValueError("at most 64 frames per sharded call")
        _, shards = tile_cyclic_assignment(F, H, W, self.world)
        n_img = H * W
        # rank r's share of frame f (tile order), and the common padded size of a share
        per = [[s[(s >= f * n_img) & (s < (f + 1) * n_img)] for f in range(F)] for s in shards]
        self.n_unit = max(1, max(len(x) for row in per for x in row))
        mine = per[self.rank]
        self.local_real = [len(x) for x in mine]
        self.n_local = int(sum(self.local_real))
        self.n_pad = F * self.n_unit
        self.global_rays = n_img

        def padded(x):      # padding repeats a real ray (any ray of the image if the share is empty): never alive, never kept
            fill = x[-1:] if len(x) else np.zeros((1,), np.int64)
            return np.concatenate([x, np.repeat(fill, self.n_unit - len(x))])
        idx = np.concatenate([padded(x) for x in mine])
        idx_t = torch.from_numpy(idx.astype(np.int64)).to(o.device)
        self._ray_index = idx_t
        self.local_o = o[idx_t].contiguous()
        self.local_d = d[idx_t].contiguous()
        # destination (flat ray id) of every gathered row; padded rows and every rank's trailing count row go to a
        # position outside the image (dropped by the scatter)
        n_rays = F * n_img
        dest = np.full((self.world, F, self.n_unit), n_rays, np.int64)
        for r in range(self.world):
            for f in range(F):
                dest[r, f, :len(per[r][f])] = per[r][f]
        dest = np.concatenate([dest.reshape(self.world, -1), np.full((self.world, 1), n_rays, np.int64)], axis=1)
        self.gather_index = torch.from_numpy(dest.reshape(-1)).to(o.device)

    @torch.no_grad()
    def render_local(self, timestamps: torch.Tensor):
        """Render this rank's shard (no communication)."""
        rgb, op, dp, n_samples = self.render_fn(self.local_o, self.local_d, timestamps)
        return rgb, op, dp, int(n_samples)

    @torch.no_grad()
    def gather(self, local, sync_total: bool = True) -> Dict:
        """All-gather the shards' pixels and put them back in raster order (one collective).  sync_total=False
        leaves the all-rank sample count on the device (`total_samples_tensor`, float64 scalar) instead of
        reading it back, so the call only enqueues work (used when gathers overlap the next frames)."""
        F, H, W = self.shape
        rgb, op, dp, n_samples = local
        if self.gather_index is None:
            if self.unpermute is not None:
                if rgb.is_cuda:        # tile order -> raster in one HIP pass (ced_scatter_pixels)
                    from . import ops
                    rgb, op, dp, _ = ops.scatter_pixels(self._ray_index, F * H * W, rgb.reshape(-1, 3), op.reshape(-1, 1),
                                                        dp.reshape(-1, 1))
                else:
                    rgb, op, dp = rgb[self.unpermute], op[self.unpermute], dp[self.unpermute]
            return dict(rgb=rgb.view(F, H, W, 3), opacity=op.view(F, H, W, 1), depth=dp.view(F, H, W, 1),
                        local_samples=n_samples, total_samples=n_samples)
        payload = torch.empty((self.n_pad + 1, 5), device=rgb.device, dtype=torch.float32)
        payload[:-1, 0:3] = rgb.reshape(-1, 3)
        payload[:-1, 3:4] = op.reshape(-1, 1)
        payload[:-1, 4:5] = dp.reshape(-1, 1)
        payload[-1].zero_()
        payload[-1, 0] = float(n_samples >> 16)          # exact in float32: two 16-bit halves
        payload[-1, 1] = float(n_samples & 0xFFFF)
        gathered = torch.empty((self.world, self.n_pad + 1, 5), device=rgb.device, dtype=torch.float32)
        if payload.is_cuda and dist.get_backend(self.schedule_group) == "gloo":
            # rehearsal of the multi-rank path without RCCL (ranks sharing one card): stage through the host
            parts = [torch.empty((self.n_pad + 1, 5), dtype=torch.float32) for _ in range(self.world)]
            dist.all_gather(parts, payload.cpu(), group=self.schedule_group)
            gathered.copy_(torch.stack(parts))
        else:
            dist.all_gather_into_tensor(gathered.view(-1, 5), payload, group=self.schedule_group)
        tail = gathered[:, -1, :2].to(torch.float64)
        total_t = (tail[:, 0] * 65536.0 + tail[:, 1]).sum()
        total = int(total_t.item()) if sync_total else None
        n_rays = F * H * W
        rows = gathered.view(-1, 5)
        if rows.is_cuda:               # the un-permute of the gathered shards: one HIP pass (ced_scatter_pixels)
            from . import ops
            o_rgb, o_op, o_dp, _ = ops.scatter_pixels(self.gather_index, n_rays, rows[:, 0:3], rows[:, 3:4], rows[:, 4:5])
        else:                          # CPU tensors: the gloo tests of the sharding logic
            image = torch.empty((n_rays + 1, 5), dtype=torch.float32)
            image[self.gather_index] = rows
            o_rgb, o_op, o_dp = image[:n_rays, 0:3], image[:n_rays, 3:4], image[:n_rays, 4:5]
        return dict(rgb=o_rgb.reshape(F, H, W, 3), opacity=o_op.reshape(F, H, W, 1), depth=o_dp.reshape(F, H, W, 1),
                    local_samples=n_samples, total_samples=total, total_samples_tensor=total_t)

    @torch.no_grad()
    def render(self, timestamps: torch.Tensor) -> Dict:
        return self.gather(self.render_local(timestamps))


def _all_reduce_on(row: torch.Tensor, stream: int, group) -> None:
    """all-reduce(sum) of `row`, enqueued on the HIP stream with handle `stream` (0: the current one; host tensors: now)"""
    if not row.is_cuda:
        dist.all_reduce(row, op=dist.ReduceOp.SUM, group=group)
        return
    ext = torch.cuda.ExternalStream(stream, device=row.device) if stream else torch.cuda.current_stream(row.device)
    with torch.cuda.stream(ext):
        dist.all_reduce(row, op=dist.ReduceOp.SUM, group=group)


class ExchangeTimeout(RuntimeError):
    """A lane or the collecting thread of a PipelinedRenderer waited longer than `comm_timeout_s` for its counterpart."""


class PipelinedRenderer:
    """Several independent ray batches ("lanes", e.g. consecutive frames of a video) in flight at once.

    Each lane is a ShardedRenderer with its own HIP stream and host thread, so the latency-bound
    marching / compositing launches and the per-iteration host round trip of one frame overlap the
    MFMA-bound field kernel of another (frames are independent: train_real.py:531-558 renders them
    one after the other).  Every lane is a complete render_image_test call, so per-frame results
    are exactly those of rendering the frames one at a time.

    Collectives (several ranks): ONE issuing thread, ONE communicator, an order no rank can disagree on.  A lane never
    calls torch.distributed itself.  Its native call asks for the all-reduce of an iteration's survivor counts through
    `ops.ScheduleExchange.issuer`: the request goes into the lane's message queue and the lane's thread waits until the
    collecting thread (the caller of `render` / `render_steps`) has enqueued the collective on the lane's stream; a
    finished call is a message too, answered with the pixel all-gather.  The collecting thread takes ONE message from
    each lane in turn, round robin, skipping only lanes that have delivered all their steps.  A lane's message sequence
    (the iterations of its calls and their ends) is fixed by the image-global schedule, identical on every rank, so the
    order in which collectives reach the communicator is identical on every rank whatever the lanes' relative speed --
    which is all RCCL asks for; there is nothing for two ranks to wait for each other on.  (Rounds 2-3 gave every lane a
    communicator of its own and let the lanes' threads issue; that is deadlock-free only if every communicator's stream
    progresses independently of the others, a property of hardware-queue placement that could not be verified.)
    All lanes must use the same process group (ShardedRenderer.schedule_group); a mixture raises.  Every wait of this
    protocol has a deadline (`comm_timeout_s`, default $CED_COMM_TIMEOUT_S or 120): on expiry ExchangeTimeout names
    the lane, step and iteration that is stuck; a collective stuck ON THE DEVICE is the process group's own timeout
    (torch.distributed.init_process_group(timeout=...), which bench.py sets to the same figure).

    Streams: every lane has its own stream (plus one for the gathers).  HIP maps a process's streams onto 4 hardware
    queues by default and streams sharing a queue serialise: with more than 3 lanes, or 3 lanes and async_gather, export
    GPU_MAX_HW_QUEUES=8 before the process initialises HIP (bench.py does).  That is a matter of speed only.

    async_gather=True (multi-rank video rendering): the gathers and the un-permute run on a communication
    stream of their own and `render` returns without waiting for them, so the exchange of one step overlaps
    the marching / field kernels of the next; the returned images are valid after `wait_gathers()` (or a
    device synchronise), and the all-rank sample count stays on the device (`total_samples_tensor`)."""

    def __init__(self, lanes, share_field_stream: bool = False, async_gather: bool = False,
                 field_max_blocks: Optional[int] = 128, comm_timeout_s: Optional[float] = None):
        import os
        from concurrent.futures import ThreadPoolExecutor
        self.lanes = list(lanes)
        # With several frames in flight a field launch is capped at half the CUs (a per-call property of the lanes'
        # native calls, ced_field_desc.max_workgroups): two frames' field kernels then run side by side and the third
        # frame's marching / compositing launches find free CUs instead of queueing behind a chip-wide kernel (+5 %
        # frames/s with 3 lanes).  A renderer outside this pipeline keeps one workgroup per CU: nothing is process-wide.
        self.field_max_blocks = field_max_blocks if len(self.lanes) > 1 else None
        if self.field_max_blocks is not None:
            for lane in self.lanes:
                lane.field_max_workgroups = int(self.field_max_blocks)
        self.async_gather = bool(async_gather)
        self.comm_timeout_s = float(comm_timeout_s if comm_timeout_s is not None else os.environ.get("CED_COMM_TIMEOUT_S", "120"))
        self.comm_stream = None
        self.streams = [torch.cuda.Stream(device=l.device) if torch.cuda.is_available() and str(l.device) != "cpu"
                        else None for l in self.lanes]
        self.pool = ThreadPoolExecutor(max_workers=len(self.lanes)) if len(self.lanes) > 1 else None
        collective = [l for l in self.lanes if getattr(l, "world", 1) > 1 or getattr(l, "force_collective", False)]
        if self.pool is not None and collective:
            groups = {id(getattr(l, "schedule_group", None)) for l in self.lanes}
            if len(groups) != 1:
                raise ValueError("PipelinedRenderer: the lanes use different process groups; every lane's collectives are "
                                 "issued by ONE thread on ONE communicator (pass the same schedule_group, or None, to all)")
            for i, lane in enumerate(self.lanes):
                lane.exchange_issuer = self._issuer_of_lane(i)
                if getattr(lane, "exchange", None) is not None:
                    lane.exchange.issuer = lane.exchange_issuer
        self._msgq = None              # per-lane message queues of the running render_steps
        self._abort = None             # the exception that ended it, seen by lanes waiting for an acknowledgement
        self._where = None             # (lane, step) the collecting thread is waiting for: diagnostics
        if share_field_stream and self.pool is not None and self.streams[0] is not None:
            # Optional: one stream for every lane's field kernel, so those launches queue instead of
            # sharing the chip (clean per-launch timings).  Off by default: letting the field kernels of
            # different frames overlap fills each other's launch tails and is measurably faster.
            self.field_stream = torch.cuda.Stream(device=self.lanes[0].device)
            for lane in self.lanes:
                lane.field_stream = self.field_stream

    def _issuer_of_lane(self, i):
        """ops.ScheduleExchange.issuer of lane i: runs on the lane's thread inside its native call."""
        import threading

        def issue(exchange, row, stream, iteration):
            if self._abort is not None:    # the run this lane belongs to has failed or timed out: never issue on its own now
                raise self._abort
            q = self._msgq
            if q is None:                  # the lane renders outside render / render_steps (lane.render_local): on its own
                _all_reduce_on(row, stream, exchange.group)
                return
            ack, box = threading.Event(), {}
            q[i].put(("reduce", iteration, row, stream, exchange.group, ack, box))
            waited = 0.0
            while not ack.wait(0.05):
                waited += 0.05
                if self._abort is not None:
                    raise self._abort
                if waited >= self.comm_timeout_s:
                    raise ExchangeTimeout(f"lane {i}: the survivor-count all-reduce of iteration {iteration} was not issued within "
                                          f"{self.comm_timeout_s:.0f} s; the collecting thread is waiting for (lane, step) {self._where}")
            if "error" in box:
                raise box["error"]
        return issue

    def _lane(self, i, timestamps):
        lane, stream = self.lanes[i], self.streams[i]
        if stream is None:
            return lane.render_local(timestamps)
        with torch.cuda.stream(stream):
            return lane.render_local(timestamps)

    @torch.no_grad()
    def render(self, timestamps: torch.Tensor):
        """One step: one call per lane, gathered.  (`render_steps` with a single step.)"""
        return self.render_steps(timestamps, 1)[0]

    @torch.no_grad()
    def render_steps(self, timestamps: torch.Tensor, n_steps: int, before_frame: Optional[Callable] = None):
        """`n_steps` consecutive steps (each step = one frame per lane) WITHOUT joining the lanes between
        steps: every lane renders its frame n_steps times back to back, so a lane that finishes early starts its next
        frame instead of waiting for the slowest one (a video is a stream of frames).  The calling thread collects the
        lanes' messages -- requests for an iteration's all-reduce, finished calls -- one per lane in turn and issues every
        collective itself (class comment).
        `before_frame(lane_index, step)` runs on the lane's thread (with the lane's stream current) before each frame,
        e.g. to swap tracers or to hand the lane the rays of its next frame.  `timestamps` may be a callable
        `(lane_index, step) -> tensor` for per-frame times.  An exception on a lane's thread is re-raised here.
        Returns a list (per step) of lists (per lane) of `gather` results."""
        import queue
        L = len(self.lanes)
        main = torch.cuda.current_stream() if self.streams[0] is not None else None
        if main is not None:
            for s in self.streams:
                s.wait_stream(main)             # the inputs were produced on the caller's stream
        msgq = [queue.Queue() for _ in range(L)]
        self._msgq, self._abort = msgq, None

        def work(i):
            step = 0
            try:
                for step in range(n_steps):
                    if before_frame is not None:
                        if self.streams[i] is not None:
                            with torch.cuda.stream(self.streams[i]):
                                before_frame(i, step)
                        else:
                            before_frame(i, step)
                    if callable(timestamps) and self.streams[i] is not None:
                        # on the lane's stream, like everything else of the frame: a tensor the callable BUILDS (e.g. the
                        # concatenated times of a call's frames, video.render_video) would otherwise be produced on this
                        # thread's default stream, which nothing orders before the lane's kernels that read it
                        with torch.cuda.stream(self.streams[i]):
                            ts = timestamps(i, step)
                    else:
                        ts = timestamps(i, step) if callable(timestamps) else timestamps
                    msgq[i].put(("done", step, self._lane(i, ts)))
            except BaseException as e:           # hand the failure to the collecting thread instead of leaving it waiting
                msgq[i].put(("error", step, e))

        if self.async_gather and self.streams[0] is not None and self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=self.lanes[0].device)

        def gather(i, loc):
            lane = self.lanes[i]
            if self.async_gather and self.streams[0] is not None:
                with torch.cuda.stream(self.comm_stream):
                    self.comm_stream.wait_stream(self.streams[i])
                    for t in loc[:3]:
                        t.record_stream(self.comm_stream)       # produced on the lane's stream, consumed here
                    return lane.gather(loc, sync_total=False)
            if main is not None:
                main.wait_stream(self.streams[i])
                for t in loc[:3]:
                    t.record_stream(main)
            return lane.gather(loc)

        outs = [[None] * L for _ in range(n_steps)]
        if self.pool is None:
            # one lane: it runs on this thread; without an issuer its exchange issues its own collectives, in program order
            work(0)
            for step in range(n_steps):
                kind, st, payload = msgq[0].get()
                if kind == "error":
                    self._msgq = None
                    raise payload
                outs[st][0] = gather(0, payload)
            self._msgq = None
            return outs
        threads = [self.pool.submit(work, i) for i in range(L)]
        step_of = [0] * L
        failure = None
        lane_i = 0
        try:
            while any(s < n_steps for s in step_of):
                if step_of[lane_i] >= n_steps:
                    lane_i = (lane_i + 1) % L
                    continue
                self._where = (lane_i, step_of[lane_i])
                try:
                    msg = msgq[lane_i].get(timeout=self.comm_timeout_s)
                except queue.Empty:
                    raise ExchangeTimeout(f"lane {lane_i}, step {step_of[lane_i]}: no message (all-reduce request or finished "
                                          f"call) for {self.comm_timeout_s:.0f} s; steps delivered per lane: {step_of}") from None
                if msg[0] == "reduce":
                    _, iteration, row, stream, group, ack, box = msg
                    try:
                        _all_reduce_on(row, stream, group)
                    except BaseException as e:
                        box["error"] = e
                        ack.set()
                        raise
                    ack.set()
                elif msg[0] == "done":
                    outs[msg[1]][lane_i] = gather(lane_i, msg[2])
                    step_of[lane_i] += 1
                else:
                    raise msg[2]
                lane_i = (lane_i + 1) % L
        except BaseException as e:
            failure = e
            self._abort = e                       # lanes waiting for an acknowledgement give up with the same error
        finally:
            if failure is not None and not isinstance(failure, ExchangeTimeout):
                for t in threads:                 # the lanes end by themselves (their waits see _abort); a timeout does not wait
                    try:
                        t.result(timeout=self.comm_timeout_s)
                    except BaseException:
                        pass
            elif failure is None:
                for t in threads:
                    t.result()
            if failure is None:
                self._msgq = None
            # after a failure `_abort` stays set and the queues stay in place: a lane that is still inside a native call
            # (a timeout does not wait for it) gets the error from its next request instead of issuing a collective itself.
            # The renderer is not usable after an ExchangeTimeout (bench.py leaves the process).
        if failure is not None:
            raise failure
        return outs

    def wait_gathers(self) -> None:
        """Make the caller's stream wait for every gather issued so far (async_gather mode)."""
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
