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


def test_frame_groups_need_equal_ray_shares():
    """ShardedRenderer(units=U): the frames split into U groups and every rank's share of every group must be the same
    number of rays (one native call renders them as U equal units); other shapes are refused up front, and the
    assignment of a shape is computed once."""
    from ced_nerf_amd import dist as cdist
    from ced_nerf_amd.dist import ShardedRenderer
    fn = lambda o, d, ts: (torch.zeros(o.shape[0], 3), torch.zeros(o.shape[0], 1), torch.zeros(o.shape[0], 1), 0)
    rays = lambda F, H, W: (torch.zeros(F, H, W, 3), torch.ones(F, H, W, 3))
    r = ShardedRenderer(None, None, 1, 0, "cpu", render_fn=fn, units=2)
    with pytest.raises(ValueError):
        r.set_rays(*rays(3, 16, 16))                       # 3 frames do not split into 2 groups
    r.set_rays(*rays(4, 16, 16))
    assert r.n_local == 4 * 256
    two = ShardedRenderer(None, None, 2, 1, "cpu", render_fn=fn, units=2)
    two.set_rays(*rays(4, 16, 16))                         # 4 tiles per frame, 2 ranks: equal shares
    assert two.n_local == two.n_pad == 2 * 256
    odd = ShardedRenderer(None, None, 2, 0, "cpu", render_fn=fn, units=3)
    with pytest.raises(ValueError):
        odd.set_rays(*rays(3, 8, 24))                      # 3 tiles per frame over 2 ranks: uneven shares per group
    with pytest.raises(AssertionError):
        ShardedRenderer(None, None, 1, 0, "cpu", render_fn=fn, units=9)
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
