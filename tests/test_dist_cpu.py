"""CPU suite, part 3: the N>1 path (ray sharding + pixel all-gather) with world_size 2 over gloo.
The per-rank renderer is the CPU oracle's per-ray-deterministic `render_image`, so the gathered
image must equal the single-process image bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _scene():
    from ced_nerf_amd import synthetic as S
    a = S.make_scene("dnerf", 40, 24, "trained", log2_hashmap_size=12)
    b = S.make_scene("dnerf", 40, 24, "trained", log2_hashmap_size=12, azim_deg=70.0)
    return a, b


def _oracle_render_fn(sc):
    from oracle import oracle as O
    cfg = sc["cfg"]
    f = O.OracleField(sc["params"]); est = O.OracleEstimator(cfg["aabb"], 128, 1, sc["binaries"])

    def fn(rays_o, rays_d, timestamps):
        out = O.render_image(f, est, rays_o.numpy(), rays_d.numpy(), timestamps=timestamps.numpy(), **sc["render"])
        return torch.from_numpy(out[0]), torch.from_numpy(out[1]), torch.from_numpy(out[2]), out[3]
    return fn


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ced_nerf_amd.dist import ShardedRenderer
    a, b = _scene()
    r = ShardedRenderer(None, None, world, rank, "cpu", render_fn=_oracle_render_fn(a))
    o = torch.from_numpy(np.stack([a["origins"], b["origins"]])); d = torch.from_numpy(np.stack([a["viewdirs"], b["viewdirs"]]))
    r.set_rays(o, d)
    out = r.render(torch.from_numpy(a["timestamps"]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), rgb=out["rgb"].numpy(), opacity=out["opacity"].numpy(),
             depth=out["depth"].numpy(), total=out["total_samples"], local=out["local_samples"], n_local=r.n_local)
    dist.destroy_process_group()


def _global_schedule_render_fn(sc, renderer, group=None, local_schedule=False):
    """render_image_test of the oracle on this rank's shares, one frame after the other, every frame on the schedule
    of the WHOLE image: the alive count all-reduced over the ranks each iteration (cednerf/utils.py:231-235).
    local_schedule=True is the non-conforming variant (every shard on a schedule of its own), for contrast."""
    from oracle import oracle as O
    cfg = sc["cfg"]
    f = O.OracleField(sc["params"]); est = O.OracleEstimator(cfg["aabb"], 128, 1, sc["binaries"])

    def reduce(n):
        t = torch.tensor([n], dtype=torch.int64)
        dist.all_reduce(t, group=group)
        return int(t[0])

    def fn(rays_o, rays_d, timestamps):
        F, n_unit = renderer.shape[0], renderer.n_unit
        o = rays_o.numpy().reshape(F, n_unit, 3); d = rays_d.numpy().reshape(F, n_unit, 3)
        outs, total = [], 0
        for k in range(F):
            kw = {} if local_schedule else dict(alive_reduce=reduce, n_total=renderer.global_rays)
            out = O.render_image_test(64, f, est, o[k], d[k], timestamps=timestamps.numpy(), n_real=renderer.local_real[k],
                                      **kw, **sc["render"])
            outs.append(out); total += out[3]
        cat = lambda i: torch.from_numpy(np.concatenate([x[i] for x in outs]))
        return cat(0), cat(1), cat(2), total
    return fn


def _worker_global_schedule(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ced_nerf_amd.dist import ShardedRenderer
    a, b = _scene()
    o = torch.from_numpy(np.stack([a["origins"], b["origins"]])); d = torch.from_numpy(np.stack([a["viewdirs"], b["viewdirs"]]))
    res = {}
    for name, local in (("global", False), ("local", True)):
        r = ShardedRenderer(None, None, world, rank, "cpu")
        r.render_fn = _global_schedule_render_fn(a, r, local_schedule=local)
        r.set_rays(o, d)
        out = r.render(torch.from_numpy(a["timestamps"]))
        res.update({f"{name}_{k}": out[k].numpy() for k in ("rgb", "opacity", "depth")})
        res[f"{name}_total"] = out["total_samples"]
        res["n_unit"], res["local_real"] = r.n_unit, np.asarray(r.local_real)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **res)
    dist.destroy_process_group()


def test_sharded_render_image_test_runs_the_image_global_schedule(tmp_path):
    """render_image_test's loop is one loop per IMAGE (N_samples = N_rays // N_alive over all its rays, a ray alive on
    packed_info[:, 1] == N_samples: cednerf/utils.py:231-235,301-306).  Two ranks, every frame's rays dealt in 8x8 tiles
    (15 tiles per frame: uneven, padded shares), the alive count all-reduced per iteration: the gathered frames and the
    sample total are those of one process rendering the whole frames -- bit for bit; shards on schedules of their own
    (what round 2 did) are not."""
    from oracle import oracle as O
    port = _free_port()
    mp.spawn(_worker_global_schedule, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = _scene()
    cfg = a["cfg"]
    f = O.OracleField(a["params"]); est = O.OracleEstimator(cfg["aabb"], 128, 1, a["binaries"])
    whole = [O.render_image_test(64, f, est, s["origins"], s["viewdirs"], timestamps=a["timestamps"], **a["render"]) for s in (a, b)]
    r0 = np.load(os.path.join(tmp_path, "rank0.npz")); r1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    assert int(r0["n_unit"]) == 512 and sorted(r0["local_real"].tolist() + r1["local_real"].tolist()) == [448, 448, 512, 512]
    for i, k in enumerate(("rgb", "opacity", "depth")):
        want = np.stack([w[i] for w in whole])
        assert np.array_equal(r0[f"global_{k}"], r1[f"global_{k}"])
        assert np.array_equal(r0[f"global_{k}"].reshape(want.shape), want), k
    total = sum(w[3] for w in whole)
    assert int(r0["global_total"]) == int(r1["global_total"]) == total and total > 1000
    # the contrast: per-shard schedules march other sample sets
    assert int(r0["local_total"]) != total


def test_tile_cyclic_assignment_is_a_balanced_partition():
    from ced_nerf_amd.dist import tile_cyclic_assignment
    for (F, H, W, world) in ((1, 800, 800, 8), (2, 40, 24, 2), (3, 50, 70, 4), (1, 13, 9, 3)):
        owner, shards = tile_cyclic_assignment(F, H, W, world)
        allr = np.concatenate(shards)
        assert np.array_equal(np.sort(allr), np.arange(F * H * W))
        assert all(np.all(owner[s] == r) for r, s in enumerate(shards))
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 64 * ((H + 7) // 8 + 1)
    _, shards = tile_cyclic_assignment(1, 800, 800, 8)
    assert [len(s) for s in shards] == [80000] * 8


def test_frame_groups_and_padded_frame_shares():
    """One rank, units=U: the frames split into U equal groups (one native call renders them as U units).  Several
    ranks: a unit is one rank's share of ONE frame, padded to the largest share; the gather index sends every real
    row to its pixel and every padding / count row outside the image."""
    from ced_nerf_amd import dist as cdist
    from ced_nerf_amd.dist import ShardedRenderer
    fn = lambda o, d, ts: (torch.zeros(o.shape[0], 3), torch.zeros(o.shape[0], 1), torch.zeros(o.shape[0], 1), 0)
    rays = lambda F, H, W: (torch.zeros(F, H, W, 3), torch.ones(F, H, W, 3))
    r = ShardedRenderer(None, None, 1, 0, "cpu", render_fn=fn, units=2)
    with pytest.raises(ValueError):
        r.set_rays(*rays(3, 16, 16))                       # 3 frames do not split into 2 groups
    r.set_rays(*rays(4, 16, 16))
    assert r.n_local == 4 * 256 and not r.sharded
    two = ShardedRenderer(None, None, 2, 1, "cpu", render_fn=fn)
    two.set_rays(*rays(4, 16, 16))                         # 4 tiles per frame, 2 ranks: equal shares
    assert two.sharded and two.n_unit == 128 and two.local_real == [128] * 4 and two.n_local == 512 and two.n_pad == 512
    odd = ShardedRenderer(None, None, 2, 0, "cpu", render_fn=fn)
    odd.set_rays(*rays(3, 8, 24))                          # 3 tiles per frame over 2 ranks: shares of 2 and 1 tiles
    assert odd.n_unit == 128 and sorted(odd.local_real) == [64, 128, 128] and odd.n_pad == 3 * 128
    g = odd.gather_index.numpy().reshape(2, 3 * 128 + 1)
    real = g[g < 3 * 8 * 24]
    assert np.array_equal(np.sort(real), np.arange(3 * 8 * 24))       # every pixel exactly once
    assert (g[:, -1] == 3 * 8 * 24).all()                            # the count rows are dropped
    with pytest.raises(AssertionError):
        ShardedRenderer(None, None, 1, 0, "cpu", render_fn=fn, units=65)
    a = cdist.tile_cyclic_assignment(4, 16, 16, 2)
    assert cdist.tile_cyclic_assignment(4, 16, 16, 2) is a          # cached per shape


def test_sharded_render_world2_gloo_matches_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    from ced_nerf_amd.dist import ShardedRenderer
    a, b = _scene()
    single = ShardedRenderer(None, None, 1, 0, "cpu", render_fn=_oracle_render_fn(a))
    single.set_rays(torch.from_numpy(np.stack([a["origins"], b["origins"]])),
                    torch.from_numpy(np.stack([a["viewdirs"], b["viewdirs"]])))
    want = single.render(torch.from_numpy(a["timestamps"]))
    r0 = np.load(os.path.join(tmp_path, "rank0.npz")); r1 = np.load(os.path.join(tmp_path, "rank1.npz"))
    for k in ("rgb", "opacity", "depth"):
        assert np.array_equal(r0[k], r1[k])                       # every rank holds the whole image
        assert np.array_equal(r0[k], want[k].numpy()), k          # and it is the single-process image
    assert int(r0["total"]) == int(r1["total"]) == want["total_samples"] == int(r0["local"]) + int(r1["local"])
    assert int(r0["n_local"]) + int(r1["n_local"]) == 2 * 40 * 24
    assert want["rgb"].shape == (2, 24, 40, 3) and want["rgb"].std() > 0.01


# ---- several lanes, several ranks: ONE collecting thread issues every collective in an order timing cannot change ----
def _lane_script(lane, step):
    """how many all-reduce requests the call (lane, step) makes: differs per lane and step, identical on every rank"""
    return 2 + (3 * lane + 5 * step) % 4


def _worker_funnel(rank, world, port, out_dir, stall_lane):
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ced_nerf_amd.dist import ExchangeTimeout, PipelinedRenderer, ShardedRenderer

    class Ex:                                  # what the issuer needs of an ops.ScheduleExchange
        group = None

    H, W, L, n_steps = 16, 16, 3, 4
    rng = np.random.default_rng(100 + rank)    # rank-DEPENDENT delays: the lanes interleave differently on the two ranks
    log = []
    lanes = []
    for l in range(L):
        r = ShardedRenderer(None, None, world, rank, "cpu")
        state = {"step": 0}

        def fn(o, d, ts, r=r, l=l, state=state):
            step = state["step"]; state["step"] += 1
            for it in range(_lane_script(l, step)):
                time.sleep(float(rng.uniform(0, 0.02)))
                if stall_lane == l and rank == 1 and step == 1 and it == 1:
                    time.sleep(3.0)            # one rank's lane goes silent: the other rank must time out, not hang
                row = torch.tensor([rank + 1, 10 * l + step, it], dtype=torch.int64)
                r.exchange_issuer(Ex, row, 0, it)
                assert row.tolist() == [3, 2 * (10 * l + step), 2 * it], (l, step, it, row.tolist())
                log.append((l, step, it))
            n = o.shape[0]
            val = float(100 * l + step)
            return torch.full((n, 3), val), torch.full((n, 1), val), torch.full((n, 1), float(rank)), 7
        r.render_fn = fn
        r.set_rays(torch.zeros(1, H, W, 3), torch.ones(1, H, W, 3))
        lanes.append(r)
    pipe = PipelinedRenderer(lanes, comm_timeout_s=1.0 if stall_lane >= 0 else 30.0)
    try:
        outs = pipe.render_steps(torch.zeros(1), n_steps)
        ok = all(float(outs[s][l]["rgb"].min()) == float(outs[s][l]["rgb"].max()) == 100 * l + s and outs[s][l]["total_samples"] == 14
                 for s in range(n_steps) for l in range(L))
        res = "ok" if ok and len(log) == sum(_lane_script(l, s) for l in range(L) for s in range(n_steps)) else "wrong"
    except ExchangeTimeout as e:
        res = "timeout: " + str(e)
    except BaseException as e:                 # e.g. gloo: "connection closed by peer" once the other rank has left
        res = "error: " + str(e)
    open(os.path.join(out_dir, f"rank{rank}.txt"), "w").write(res)
    if res != "ok":
        os._exit(0)          # as bench.py does: lanes may still sit in a wait; never hang the process on them
    dist.destroy_process_group()


def test_lanes_of_two_ranks_funnel_their_collectives_through_one_thread(tmp_path):
    """3 lanes x 4 steps on 2 gloo ranks; every call asks for 2-5 all-reduces after random, rank-dependent delays.  Every
    reduced row must be the sum of the SAME (lane, step, iteration) row of both ranks -- any disagreement about the order
    of the collectives on the shared communicator would add up different rows (or hang) -- and every gathered image is
    its (lane, step)'s own."""
    mp.spawn(_worker_funnel, args=(2, _free_port(), str(tmp_path), -1), nprocs=2, join=True)
    assert [open(os.path.join(tmp_path, f"rank{r}.txt")).read() for r in range(2)] == ["ok", "ok"]


def test_a_stuck_lane_ends_in_a_named_timeout_not_in_a_hang(tmp_path):
    """Rank 1's lane 2 goes silent for 3 s with a 1 s deadline: rank 1 names the (lane, step) it was waiting for and
    leaves; rank 0 (whose collecting thread sits in gloo's blocking all-reduce of that very iteration) ends with a timeout
    of its other lanes or with gloo's "peer gone".  Nobody waits for ever, nobody reports success."""
    mp.spawn(_worker_funnel, args=(2, _free_port(), str(tmp_path), 2), nprocs=2, join=True)
    msgs = [open(os.path.join(tmp_path, f"rank{r}.txt")).read() for r in range(2)]
    assert all(m.startswith("timeout: ") or m.startswith("error: ") for m in msgs), msgs
    assert msgs[1].startswith("timeout: ") and "lane 2, step 1" in msgs[1], msgs


def test_lanes_on_different_process_groups_are_refused():
    from ced_nerf_amd.dist import PipelinedRenderer, ShardedRenderer
    a = ShardedRenderer(None, None, 2, 0, "cpu", schedule_group=object())
    b = ShardedRenderer(None, None, 2, 0, "cpu", schedule_group=object())
    with pytest.raises(ValueError, match="ONE communicator"):
        PipelinedRenderer([a, b])
    PipelinedRenderer([ShardedRenderer(None, None, 1, 0, "cpu"), ShardedRenderer(None, None, 1, 0, "cpu")])   # no collectives: fine
