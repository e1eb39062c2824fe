"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): sample indices / counts bit-exact; composited RGB / depth within
1e-4 abs.  Because the kernels and the oracle share one arithmetic contract, most float outputs
are in fact compared bit for bit.
"""
import os

import numpy as np
import pytest
import torch

from conftest import assert_bitexact, free_port

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def T(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def random_rays(n, seed, radius=4.0, spread=1.2):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3)); o = o / np.linalg.norm(o, axis=1, keepdims=True) * radius
    target = rng.uniform(-spread, spread, size=(n, 3))
    d = target - o; d = d / np.linalg.norm(d, axis=1, keepdims=True)
    o = o.astype(np.float32); d = d.astype(np.float32)
    # edge cases: axis-parallel rays, zero components, origins inside the box, rays that miss
    o[0] = [-2, 0.01, 0.01]; d[0] = [1, 0, 0]
    o[1] = [0.1, 0.2, -3]; d[1] = [0, 0, 1]
    o[2] = [0, 0, 0]; d[2] = [0.6, 0.8, 0]
    o[3] = [5, 5, 5]; d[3] = [1, 0, 0]
    o[4] = [0.3, -0.2, 0.1]; d[4] = [-0.57735026, 0.57735026, 0.57735026]
    return o, d


@pytest.mark.parametrize("levels", [1, 2, 4])
def test_ray_aabb_intersect(oracle, levels):
    from ced_nerf_amd import nerfacc_api as A
    o, d = random_rays(5000, 1)
    aabbs = oracle.make_aabbs([-1, -1, -1, 1, 1, 1], levels)
    want = oracle.ray_aabb_intersect(o, d, aabbs)
    got = A.ray_aabb_intersect(T(o), T(d), T(aabbs))
    for g, w, nm in zip(got, want, ("t_mins", "t_maxs", "hits")):
        assert_bitexact(N(g), w, nm)


def _scene(name, w, h, regime="trained", **kw):
    from ced_nerf_amd import synthetic as S
    return S.make_scene(name, w, h, regime, **kw)


@pytest.mark.parametrize("name,limit,masked", [("dnerf", 0, False), ("dnerf", 7, True), ("hypernerf", 0, False),
                                               ("hypernerf", 4, True), ("dynerf", 0, False), ("dynerf", 16, True)])
def test_traverse_grids(oracle, name, limit, masked):
    from ced_nerf_amd import nerfacc_api as A
    sc = _scene(name, 96, 64, log2_hashmap_size=15)
    o = sc["origins"].reshape(-1, 3); d = sc["viewdirs"].reshape(-1, 3)
    n = o.shape[0]
    cfg = sc["cfg"]
    aabbs = oracle.make_aabbs(cfg["aabb"], cfg["grid_levels"])
    near = np.full(n, cfg["near_plane"], np.float32); far = np.full(n, cfg["far_plane"], np.float32)
    mask = None
    if masked:
        mask = np.random.default_rng(0).uniform(size=n) < 0.7
    want = oracle.traverse_grids(o, d, sc["binaries"], aabbs, near, far, cfg["render_step_size"], cfg["cone_angle"],
                                 limit, False, mask)
    assert want["t_starts"].shape[0] > 1000
    i_, s_, term = A.traverse_grids(T(o), T(d), T(sc["binaries"]), T(aabbs), T(near), T(far),
                                    cfg["render_step_size"], cfg["cone_angle"], limit if limit else None, False,
                                    None if mask is None else T(mask))
    assert_bitexact(N(s_.packed_info), want["packed_info"], "packed_info")
    assert_bitexact(N(s_.ray_indices), want["ray_indices"], "ray_indices")
    assert_bitexact(N(i_.vals[i_.is_left]), want["t_starts"], "t_starts")
    assert_bitexact(N(i_.vals[i_.is_right]), want["t_ends"], "t_ends")
    assert_bitexact(N(term), want["termination_planes"], "termination_planes")
    if limit:
        # nerfacc's over-allocated layout: same samples after the caller-side compaction
        i2, s2, term2 = A.traverse_grids(T(o), T(d), T(sc["binaries"]), T(aabbs), T(near), T(far),
                                         cfg["render_step_size"], cfg["cone_angle"], limit, True,
                                         None if mask is None else T(mask))
        assert_bitexact(N(i2.vals[i2.is_left]), want["t_starts"], "t_starts (over-allocated)")
        assert_bitexact(N(i2.vals[i2.is_right]), want["t_ends"], "t_ends (over-allocated)")
        assert_bitexact(N(s2.ray_indices[s2.is_valid]), want["ray_indices"], "ray_indices (over-allocated)")
        assert_bitexact(N(s2.packed_info[:, 1]), want["packed_info"][:, 1], "counts (over-allocated)")
        assert_bitexact(N(s2.packed_info[:, 0]), np.arange(n) * limit, "starts (over-allocated)")
        assert_bitexact(N(term2), want["termination_planes"], "termination_planes (over-allocated)")


def test_traverse_known_answer(oracle):
    """SURVEY Appendix B.2 on the GPU."""
    from ced_nerf_amd import nerfacc_api as A
    b = np.zeros((1, 2, 2, 2), bool); b[0, 1, 0, 0] = True
    o = np.array([[-2, -.5, -.5]], np.float32); d = np.array([[1, 0, 0]], np.float32)
    aabbs = np.array([[-1, -1, -1, 1, 1, 1]], np.float32)
    i_, s_, term = A.traverse_grids(T(o), T(d), T(b), T(aabbs), T(np.zeros(1, np.float32)),
                                    T(np.full(1, 1e10, np.float32)), 0.25, 0.0)
    assert N(i_.vals[i_.is_left]).tolist() == [2.0, 2.25, 2.5, 2.75]
    assert N(i_.vals[i_.is_right]).tolist() == [2.25, 2.5, 2.75, 3.0]
    assert N(s_.packed_info).tolist() == [[0, 4]] and N(term).tolist() == [3.0]


def _points(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0, 1, size=(n, 3)).astype(np.float32)
    x[0] = [0, 0, 0]; x[1] = [1, 1, 1]; x[2] = [0.5, 0.5, 0.5]; x[3] = [1, 0, 1]; x[4] = [-0.2, 1.3, 0.5]
    return x


@pytest.mark.parametrize("max_res,log2T,dtype,temporal", [
    (1024, 21, np.float32, False), (1024, 21, np.float16, False), (4096, 19, np.float32, False),
    (8192, 15, np.float16, False), (1024, 17, np.float16, True), (4096, 17, np.float32, True)])
def test_hash_encode(oracle, max_res, log2T, dtype, temporal):
    from ced_nerf_amd import ops, synthetic as S
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-4, max_res, log2T, regime="trained", table_dtype=dtype,
                            temporal_hash=temporal)
    of = oracle.OracleField({"hash": p["hash"]})
    x = _points(20000, 3)
    t = np.random.default_rng(4).uniform(0, 1, size=x.shape[0]).astype(np.float32)
    t[:3] = [0.0, 1.0, 1.0 / 3.0]
    want = of.hash_encode(x, t if temporal else None)
    table = T(p["hash"]["table"])      # the descriptor holds a raw pointer: keep the tensor alive
    desc, _ = ops.make_hash_desc(table, 16, max_res, 16, log2T, temporal)
    got = ops.hash_encode(desc, T(x), T(t) if temporal else None)
    assert_bitexact(N(got), want, "hash features")
    assert np.abs(want).max() > 0.1


@pytest.mark.parametrize("max_res,log2T,dtype", [(1024, 17, np.float32), (4096, 15, np.float16), (256, 19, np.float32)])
def test_hash_encode_backward(oracle, max_res, log2T, dtype):
    """SURVEY 8f row 2, first piece: hash_encoder_backward_kernel (hash_encoder_half.py:164-226).  The table gradient
    is a sum of atomic adds in no particular order, so it is compared with the oracle's double-precision sum to fp32
    rounding noise; zero output-gradient levels are skipped as in the reference; the position gradient follows the
    reference's w / (d w) form, compared away from the cell faces where that form divides by ~0."""
    from ced_nerf_amd import ops, synthetic as S
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-4, max_res, log2T, regime="trained", table_dtype=dtype)
    of = oracle.OracleField({"hash": p["hash"]})
    n = 30000
    x = _points(n, 5)[:n]
    rng = np.random.default_rng(6)
    dy = rng.normal(size=(x.shape[0], 32)).astype(np.float32)
    dy[::7, 4:6] = 0.0                                   # some all-zero levels (skipped, :209)
    dy[3] = 0.0
    want_grad, want_dx = of.hash_encode_backward(x, dy)
    table = T(p["hash"]["table"])
    desc, _ = ops.make_hash_desc(table, 16, max_res, 16, log2T, False)
    grad, dx = ops.hash_encode_backward(desc, T(x), T(dy))
    g = N(grad).astype(np.float64)
    scale = np.abs(want_grad).max()
    assert scale > 1.0
    # fp32 atomic sums of up to ~1e3 terms: error ~ sqrt(terms) * eps * magnitude
    assert np.abs(g - want_grad).max() <= 2e-5 * scale, np.abs(g - want_grad).max() / scale
    assert np.array_equal(g == 0, want_grad == 0)         # exactly the entries the oracle touches
    # accumulation into an existing gradient
    grad2, _ = ops.hash_encode_backward(desc, T(x), T(dy), grad_table=grad.clone(), want_dx=False)
    assert np.abs(N(grad2).astype(np.float64) - 2 * want_grad).max() <= 4e-5 * scale
    # position gradient: away from cell faces (every level's fractional position in [0.02, 0.98])
    lv = of.levels
    frac = (np.clip(x, 0, 1)[:, None, :] * lv["scale"][:16, None].astype(np.float32) + np.float32(0.5)) % 1.0
    interior = ((frac > 0.02) & (frac < 0.98)).all(axis=(1, 2))
    assert interior.sum() > 100
    d_err = np.abs(N(dx)[interior] - want_dx[interior]).max() / np.abs(want_dx[interior]).max()
    assert d_err <= 1e-4, d_err
    _, want_dxs = of.hash_encode_backward(x, dy, dx_scaled=True)
    _, dxs = ops.hash_encode_backward(desc, T(x), T(dy), dx_scaled=True)
    assert np.abs(N(dxs)[interior] - want_dxs[interior]).max() / np.abs(want_dxs[interior]).max() <= 1e-4
    # duality (size-independent property): the forward is linear in the table, so for any table perturbation dT
    # <dy, encode(T + dT) - encode(T)> == <grad_table, dT>
    if dtype == np.float32:
        dT = torch.randn_like(table) * 0.1
        d2, _ = ops.make_hash_desc((table + dT).contiguous(), 16, max_res, 16, log2T, False)
        t2 = (table + dT).contiguous()
        d2, _ = ops.make_hash_desc(t2, 16, max_res, 16, log2T, False)
        lhs = ((ops.hash_encode(d2, T(x)) - ops.hash_encode(desc, T(x))).double() * T(dy).double()).sum().item()
        rhs = (grad.double() * dT.double()).sum().item()
        assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)


@pytest.mark.parametrize("max_res,log2T,dtype", [(1024, 15, np.float32), (4096, 13, np.float16)])
def test_hash_encode_backward_temporal(oracle, max_res, log2T, dtype):
    """The temporal table's backward (hash_encoder_inter.py:202-275): per corner the key-frames k and k + 1 of the
    sample's time receive w dy (1 - t_frac) and w dy t_frac -- against the oracle's double-precision sums, with t = 0, 1,
    the key-frame times themselves and all-zero levels in the batch; duality with the forward (linear in the table)."""
    from ced_nerf_amd import ops, synthetic as S
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-4, max_res, log2T, regime="trained", table_dtype=dtype,
                            temporal_hash=True, use_time_embedding=True)
    of = oracle.OracleField({"hash": p["hash"]})
    n = 20000
    x = _points(n, 8)[:n]
    rng = np.random.default_rng(9)
    t = rng.uniform(0, 1, size=n).astype(np.float32)
    t[:8] = [0.0, 1.0, 1.0 / 3.0, 2.0 / 3.0, 0.999999, 0.3333333, 0.5, 0.6666667]
    dy = rng.normal(size=(n, 32)).astype(np.float32)
    dy[::7, 4:6] = 0.0
    dy[3] = 0.0
    want = of.hash_encode_backward_temporal(x, t, dy)
    table = T(p["hash"]["table"])
    assert table.shape[1] == 8
    desc, _ = ops.make_hash_desc(table, 16, max_res, 16, log2T, True)
    grad = ops.hash_encode_backward_temporal(desc, T(x), T(t), T(dy))
    g = N(grad).astype(np.float64)
    scale = np.abs(want).max()
    assert scale > 1.0 and grad.shape == (table.shape[0], 8)
    assert np.abs(g - want).max() <= 2e-5 * scale, np.abs(g - want).max() / scale
    assert np.array_equal(g == 0, want == 0)                      # exactly the slots the oracle touches
    grad2 = ops.hash_encode_backward_temporal(desc, T(x), T(t), T(dy), grad_table=grad.clone())
    assert np.abs(N(grad2).astype(np.float64) - 2 * want).max() <= 4e-5 * scale
    if dtype == np.float32:
        dT = torch.randn_like(table) * 0.1
        t2 = (table + dT).contiguous()
        d2, _ = ops.make_hash_desc(t2, 16, max_res, 16, log2T, True)
        lhs = ((ops.hash_encode(d2, T(x), T(t)) - ops.hash_encode(desc, T(x), T(t))).double() * T(dy).double()).sum().item()
        rhs = (grad.double() * dT.double()).sum().item()
        assert abs(lhs - rhs) <= 2e-4 * max(abs(lhs), abs(rhs), 1.0), (lhs, rhs)


def test_temporal_hash_encoder_module():
    """The reference-shaped temporal `HashEncoder` (hash_encoder_inter.py:281-430): xyzt in, [N, 32] out, a table gradient and
    no position gradient; a few Adam steps fit a target that depends on time."""
    from ced_nerf_amd.hashgrid import TemporalHashEncoder
    torch.manual_seed(0)
    enc = TemporalHashEncoder(max_params=2 ** 14, levels=16, base_res=16, max_res=256, device=DEV)
    assert enc.n_output_dims == 32 and enc.hash_table.shape[1] == 8
    xyzt = torch.rand(4096, 4, device=DEV, requires_grad=True)
    target = torch.sin(xyzt.detach()[:, :3].sum(dim=1, keepdim=True) * 4.0 + xyzt.detach()[:, 3:] * 5.0).expand(-1, 32) * 0.1
    opt = torch.optim.Adam(enc.parameters(), lr=1e-2)
    losses = []
    for _ in range(60):
        opt.zero_grad()
        loss = ((enc(xyzt) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.3 * losses[0], losses[::10]
    assert xyzt.grad is None                                       # the reference's temporal encoder gives inputs no gradient


def test_trainable_hash_encoder_module(oracle):
    """The reference-shaped `HashEncoder` module (hash_encoder_half.py:231-385) on the HIP kernels: a few SGD steps on
    a regression target reduce the loss, and autograd's gradients are the kernels' gradients."""
    from ced_nerf_amd.hashgrid import HashEncoder
    torch.manual_seed(0)
    enc = HashEncoder(max_params=2 ** 15, levels=16, base_res=16, max_res=512, device=DEV)
    assert enc.n_output_dims == 32 and enc.hash_table.shape[1] == 2
    x = torch.rand(4096, 3, device=DEV, requires_grad=True)
    target = torch.sin(x.detach().sum(dim=1, keepdim=True) * 6.0).expand(-1, 32) * 0.1
    opt = torch.optim.Adam(enc.parameters(), lr=1e-2)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        y = enc(x)
        loss = ((y - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert losses[-1] < 0.5 * losses[0], losses[::6]
    assert x.grad is not None and torch.isfinite(x.grad).all() and x.grad.abs().max() > 0
    assert enc.hash_table.grad.shape == enc.hash_table.shape


FIELD_CASES = [
    dict(), dict(use_div_offsets=True), dict(use_time_embedding=True),
    dict(use_time_embedding=True, use_time_attenuation=True, use_div_offsets=True),
    dict(table_dtype=np.float16), dict(temporal_hash=True, table_dtype=np.float16, use_time_embedding=True),
]


@pytest.mark.parametrize("case", range(len(FIELD_CASES)))
@pytest.mark.parametrize("regime", ["init", "trained"])
def test_field_forward(oracle, case, regime):
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    kw = dict(FIELD_CASES[case])
    aabb = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
    p = S.init_field_params(aabb, 1.0 / 64 if regime == "trained" else 1e-4, 1024, 17, regime=regime, seed=7 + case,
                            **kw)
    of = oracle.OracleField(p)
    rng = np.random.default_rng(11)
    n = 5000 + 37          # ragged tail: not a multiple of the 64-sample wave tile
    pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)      # some points outside the aabb
    pos[0] = [1.5, 0, 0]; pos[1] = [-1.5, -1.5, -1.5]
    t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32); t[2] = 0; t[3] = 1
    d = rng.normal(size=(n, 3)).astype(np.float32)
    want = of.forward(pos, t, d, want_geo=True)
    f = DNGPradianceField.from_params(p, DEV).eval()
    rgb, res = f(T(pos), T(t), T(d))
    assert rgb.shape == (n, 3) and res["density"].shape == (n, 1) and res["base_mlp_out"].shape == (n, 15)
    assert_bitexact(N(res["base_mlp_out"]), want["base_mlp_out"], "base_mlp_out")
    assert_bitexact(N(res["density"])[:, 0], want["density"], "density")
    assert_bitexact(N(rgb), want["rgb"], "rgb")
    assert (want["density"] == 0).any() and (want["density"] > 0).any()
    dens = f.query_density(T(pos), T(t))
    assert_bitexact(N(dens["density"])[:, 0], want["density"], "query_density")


# Half-precision MLP modes (include/cednerf_hip.h, CED_MLP_*).  Since round 4 the oracle restates what the fp16-operand
# matrix instruction computes (oracle/mfma_f16_model.h: blocks of eight products, cut below 2^(Emax-24), eight guard bits,
# one rounding -- fitted to and confirmed on 6.8 M recorded dot products of the MI355X, tests/golden/mfma_f16_records.npz
# holds a sample) and the kernels use the exact kernel's deterministic elementwise math, so every output of these modes
# is compared BIT FOR BIT with the oracle's mode of the same name:
#   f16:   operands rounded to fp16 (the reference's tcnn class, cednerf/model.py:200-222,280-309; BASELINE config 5)
#   f16x2: every operand split into two fp16 numbers (22 significant bits)
# f16x2 is additionally held to the north-star's tolerance against the PLAIN fp32 oracle (quantile bounds = measured x 3-4;
# "trained" = the amplified regime of synthetic.init_field_params, where one rounding step is magnified ~100x).
HALF_Q = {      # (prec, regime) -> {quantity: (p99, p999, mean, max)}; rgb abs, density relative, base_mlp_out / scale
    ("f16x2", "init"): dict(rgb=(3e-7, 5e-7, 1e-7, 1e-6), sig=(1e-6, 1.5e-6, 3e-7, 2e-6), geo=(2.5e-3, 3e-3, 1.5e-3, 4e-3)),
    ("f16x2", "trained"): dict(rgb=(4e-5, 1.2e-4, 2e-6, 4e-4), sig=(8e-4, 2e-3, 3e-5, 6e-3), geo=(5e-5, 1.2e-4, 2e-6, 4e-4)),
}


def _check_quantiles(err, bounds, what):
    e = np.asarray(err, np.float64).reshape(-1)
    got = (np.quantile(e, 0.99), np.quantile(e, 0.999), e.mean(), e.max())
    print(f"  QSTAT {what}: p50 {np.quantile(e, .5):.2e} p99 {got[0]:.2e} p999 {got[1]:.2e} mean {got[2]:.2e} max {got[3]:.2e}")
    for nm, g, b in zip(("p99", "p99.9", "mean", "max"), got, bounds):
        assert g <= b, f"{what}: {nm} {g:.3e} > {b:.1e}"


@pytest.mark.parametrize("case", range(len(FIELD_CASES)))
@pytest.mark.parametrize("regime", ["init", "trained"])
@pytest.mark.parametrize("prec", ["f16x2", "f16"])
def test_field_forward_half_precision(oracle, prec, case, regime):
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    kw = dict(FIELD_CASES[case])
    aabb = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
    p = S.init_field_params(aabb, 1.0 / 64 if regime == "trained" else 1e-4, 1024, 17, regime=regime, seed=7 + case,
                            **kw)
    of = oracle.OracleField(p, mlp_half=prec)
    rng = np.random.default_rng(11)
    n = 20000 + 37
    pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)
    pos[0] = [1.5, 0, 0]; pos[1] = [-1.5, -1.5, -1.5]
    t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32); t[2] = 0; t[3] = 1
    d = rng.normal(size=(n, 3)).astype(np.float32)
    want = of.forward(pos, t, d, want_geo=True)
    f = DNGPradianceField.from_params(p, DEV, mlp_precision=prec).eval()
    rgb, res = f(T(pos), T(t), T(d))
    tag = f"field {prec} {regime} case{case}"
    assert_bitexact(N(res["base_mlp_out"]), want["base_mlp_out"], tag + " base_mlp_out")
    assert_bitexact(N(res["density"])[:, 0], want["density"], tag + " density")
    assert_bitexact(N(rgb), want["rgb"], tag + " rgb")
    dens = f.query_density(T(pos), T(t))                       # the density-only launch of the same kernel
    assert_bitexact(N(dens["density"])[:, 0], want["density"], tag + " query_density")
    if prec == "f16x2":
        plain = oracle.OracleField(p).forward(pos, t, d, want_geo=True)
        Q = HALF_Q[(prec, regime)]
        got_sig = N(res["density"])[:, 0]
        assert np.array_equal(got_sig == 0, plain["density"] == 0), "selector pattern differs from the fp32 oracle's"
        both = plain["density"] != 0
        rel = np.abs(got_sig[both] - plain["density"][both]) / plain["density"][both]
        geo_scale = np.abs(plain["base_mlp_out"]).max()
        _check_quantiles(np.abs(N(rgb) - plain["rgb"]).max(axis=1), Q["rgb"], tag + " rgb vs fp32 oracle")
        _check_quantiles(rel, Q["sig"], tag + " density(rel) vs fp32 oracle")
        _check_quantiles(np.abs(N(res["base_mlp_out"]) - plain["base_mlp_out"]).max(axis=1) / geo_scale, Q["geo"],
                         tag + " geo vs fp32 oracle")


# CED_MLP_F32_HEAD16X2 ("f32+h16x2"): everything a sample count, an opacity or a depth depends on is the exact chain --
# density and the 15 geometry features BIT-IDENTICAL to the oracle -- and only mlp_head runs on split-fp16 operands:
# rgb within the north-star's 1e-4 (bounds = measured x ~3; "trained" amplifies the head's output x16).
MIXED_RGB_Q = {"init": (3e-7, 5e-7, 1e-7, 1e-6), "trained": (6e-6, 2e-5, 5e-7, 1e-4)}


@pytest.mark.parametrize("case", range(len(FIELD_CASES)))
@pytest.mark.parametrize("regime", ["init", "trained"])
def test_field_forward_exact_sigma_chain_with_split_fp16_head(oracle, case, regime):
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    kw = dict(FIELD_CASES[case])
    aabb = [-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]
    p = S.init_field_params(aabb, 1.0 / 64 if regime == "trained" else 1e-4, 1024, 17, regime=regime, seed=7 + case,
                            **kw)
    of = oracle.OracleField(p)
    rng = np.random.default_rng(11)
    n = 5000 + 37
    pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)
    pos[0] = [1.5, 0, 0]; pos[1] = [-1.5, -1.5, -1.5]
    t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32); t[2] = 0; t[3] = 1
    d = rng.normal(size=(n, 3)).astype(np.float32)
    want = of.forward(pos, t, d, want_geo=True)
    f = DNGPradianceField.from_params(p, DEV, mlp_precision="f32+h16x2").eval()
    rgb, res = f(T(pos), T(t), T(d))
    assert_bitexact(N(res["base_mlp_out"]), want["base_mlp_out"], "base_mlp_out")
    assert_bitexact(N(res["density"])[:, 0], want["density"], "density")
    _check_quantiles(np.abs(N(rgb) - want["rgb"]).max(axis=1), MIXED_RGB_Q[regime], f"field f32+h16x2 {regime} case{case} rgb")
    # and rgb bit for bit against the oracle's mode of the same name (colour head on the matrix-instruction model)
    assert_bitexact(N(rgb), oracle.OracleField(p, mlp_half="f32+h16x2").forward(pos, t, d)["rgb"], "rgb vs oracle f32+h16x2")
    dens = f.query_density(T(pos), T(t))                       # the density-only launch of the same kernel
    assert_bitexact(N(dens["density"])[:, 0], want["density"], "query_density")


def test_field_forward_large_persistent_launch(oracle):
    """One 15 M-sample launch (every wave loops over ~150 tiles): a random subset against the oracle's mode of the same
    name, bit for bit in every arithmetic mode."""
    from ced_nerf_amd import ops, synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator, march_packed
    sc = S.make_scene("dnerf", 800, 800, "trained")
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], DEV).eval()
    o = T(sc["origins"]).reshape(-1, 3); d = T(sc["viewdirs"]).reshape(-1, 3)
    est = OccGridEstimator(cfg["aabb"], 128, cfg["grid_levels"]).to(DEV); est.set_binaries(T(sc["binaries"]))
    n = o.shape[0]
    near = torch.full((n,), cfg["near_plane"], device=DEV); far = torch.full((n,), cfg["far_plane"], device=DEV)
    t0, t1, ri, _, _ = march_packed(o, d, est.binaries, est.aabbs, near, far, cfg["render_step_size"], cfg["cone_angle"])
    ts = T(sc["timestamps"]).reshape(-1)
    assert t0.shape[0] > 10_000_000
    pick = np.sort(np.random.default_rng(5).choice(t0.shape[0], size=60000, replace=False))
    pick[-1] = t0.shape[0] - 1; pick[0] = 0
    pk = torch.from_numpy(pick).to(DEV)
    o_np, d_np, ts_np = N(o), N(d), N(ts)
    sub = (N(ri[pk]), N(t0[pk]), N(t1[pk]))
    for prec in ("f32", "f16x2", "f16"):
        f.set_mlp_precision(prec)
        rgb, sigma = ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, True)
        # run-to-run reproducibility over all 15 M samples: fails within one launch if a packed-fp32 op_sel:[0,1]
        # instruction runs beside the half kernels' 16x16x32 MFMAs (field_half_device.hpp mfma_k32)
        rgb2, sigma2 = ops.field_forward_rays(f._descriptor(), o, d, ri, t0, t1, ts, False, True)
        assert torch.equal(rgb, rgb2) and torch.equal(sigma, sigma2), f"{prec}: two launches differ"
        of = oracle.OracleField(sc["params"], mlp_half=prec)
        w_rgb, w_sig = of.forward_rays(o_np, d_np, sub[0], sub[1], sub[2], ts_np, t_per_ray=False)
        assert_bitexact(N(sigma[pk]), w_sig, f"density (large launch, {prec})")
        assert_bitexact(N(rgb[pk]), w_rgb, f"rgb (large launch, {prec})")


def test_field_tile_mappings_agree():
    """The tile -> wave mapping of the persistent field kernels (field_device.hpp: field_tile_range; option
    field_spread_tiles 0 / 1 / 2 = XCD-contiguous) and the workgroup count of a launch are launch properties: every
    combination must cover every sample exactly once, i.e. give the bits of the default launch -- for sample counts around
    the tile, group and XCD-region boundaries, in the exact and the default arithmetic."""
    from ced_nerf_amd import _lib, ops, synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-3, 1024, 15, regime="trained")
    f = DNGPradianceField.from_params(p, DEV).eval()
    rng = np.random.default_rng(11)
    try:
        for n in (1, 31, 33, 127, 129, 1023, 4097, 40001, 300007):
            pos = T(rng.uniform(-1, 1, size=(n, 3)).astype(np.float32))
            t = T(rng.uniform(0, 1, size=(n, 1)).astype(np.float32))
            d = T(rng.normal(size=(n, 3)).astype(np.float32))
            for prec in ("f32", "f16x2"):
                f.set_mlp_precision(prec)
                _lib.check(_lib.lib().ced_set_option(b"field_spread_tiles", 2))
                rgb0, sig0 = ops.field_forward(f._descriptor(), pos, t, d)[:2]
                for spread in (0, 1, 2):
                    _lib.check(_lib.lib().ced_set_option(b"field_spread_tiles", spread))
                    for wgs in (0, 8, 13, 40, 256):
                        out = ops.field_forward(ops._with_workgroups(f._descriptor(), wgs), pos, t, d)
                        assert torch.equal(out[0], rgb0) and torch.equal(out[1], sig0), (n, prec, spread, wgs)
    finally:
        _lib.check(_lib.lib().ced_set_option(b"field_spread_tiles", 2))


def test_field_small_and_empty(oracle):
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-3, 1024, 15, regime="trained")
    of = oracle.OracleField(p)
    f = DNGPradianceField.from_params(p, DEV).eval()
    for n in (0, 1, 15, 16, 17, 63, 64, 65):
        rng = np.random.default_rng(n)
        pos = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
        t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32)
        d = rng.normal(size=(n, 3)).astype(np.float32)
        rgb, res = f(T(pos), T(t), T(d))
        assert rgb.shape == (n, 3)
        if n:
            want = of.forward(pos, t, d)
            assert_bitexact(N(rgb), want["rgb"], f"rgb n={n}")
            assert_bitexact(N(res["density"])[:, 0], want["density"], f"density n={n}")


def _packed_problem(n_rays, seed, empty_frac=0.3):
    rng = np.random.default_rng(seed)
    counts = rng.integers(0, 40, size=n_rays)
    counts[rng.uniform(size=n_rays) < empty_frac] = 0
    base = np.cumsum(counts) - counts
    S_ = int(counts.sum())
    packed = np.stack([base, counts], -1).astype(np.int64)
    t0 = np.sort(rng.uniform(0, 5, size=S_)).astype(np.float32)
    t1 = (t0 + rng.uniform(1e-3, 2e-2, size=S_)).astype(np.float32)
    sig = (rng.uniform(0, 1, size=S_) ** 4 * 300).astype(np.float32)
    rgbs = rng.uniform(0, 1, size=(S_, 3)).astype(np.float32)
    return packed, t0, t1, sig, rgbs


def test_skip_march_closed_form_on_device_paths(oracle):
    """Long empty runs before the first sample (camera far from a tiny occupied blob, several step
    sizes): exercises the closed-form empty-space skip inside the marching kernels."""
    from ced_nerf_amd import nerfacc_api as A
    b = np.zeros((1, 128, 128, 128), bool); b[0, 60:68, 60:68, 60:68] = True
    aabbs = np.array([[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5]], np.float32)
    o, d = random_rays(4000, 21, radius=6.0, spread=0.3)
    for step in (5e-3, 1e-3, 3.3e-3, 2.0 ** -7):
        n = o.shape[0]
        near = np.zeros(n, np.float32); far = np.full(n, 1e10, np.float32)
        want = oracle.traverse_grids(o, d, b, aabbs, near, far, step, 0.0)
        i_, s_, term = A.traverse_grids(T(o), T(d), T(b), T(aabbs), T(near), T(far), step, 0.0)
        assert want["t_starts"].shape[0] > 1000
        assert_bitexact(N(s_.packed_info), want["packed_info"], f"packed_info step={step}")
        assert_bitexact(N(i_.vals[i_.is_left]), want["t_starts"], f"t_starts step={step}")
        assert_bitexact(N(term), want["termination_planes"], f"termination step={step}")


def test_compositing_ops(oracle):
    from ced_nerf_amd import nerfacc_api as A, ops, render as R
    packed, t0, t1, sig, rgbs = _packed_problem(3000, 5)
    n_rays = packed.shape[0]
    prefix = np.random.default_rng(6).uniform(0, 1, size=t0.shape[0]).astype(np.float32)
    for pf in (None, prefix):
        w, tr, al = oracle.render_weight_from_density(t0, t1, sig, packed, pf)
        gw, gtr, gal = A.render_weight_from_density(T(t0), T(t1), T(sig), packed_info=T(packed),
                                                    prefix_trans=None if pf is None else T(pf))
        assert_bitexact(N(gw), w, "weights"); assert_bitexact(N(gtr), tr, "trans"); assert_bitexact(N(gal), al, "alphas")
    gw2, gtr2, gal2 = R.render_weight_from_density_prefix(T(t0), T(t1), T(sig), T(prefix), packed_info=T(packed))
    assert_bitexact(N(gtr2), tr, "prefix trans"); assert_bitexact(N(gal2), al, "prefix alphas")
    # ray_indices form
    ri = np.repeat(np.arange(n_rays), packed[:, 1]).astype(np.int64)
    gw3, _, _ = A.render_weight_from_density(T(t0), T(t1), T(sig), ray_indices=T(ri), n_rays=n_rays)
    w0, _, _ = oracle.render_weight_from_density(t0, t1, sig, packed, None)
    assert_bitexact(N(gw3), w0, "weights via ray_indices")
    for vals, C_ in ((rgbs, 3), (None, 1)):
        out = np.random.default_rng(8).uniform(size=(n_rays, C_)).astype(np.float32)
        want = oracle.accumulate_along_rays_(w0, vals, packed, out.copy())
        got = T(out.copy())
        A.accumulate_along_rays_(T(w0), values=None if vals is None else T(vals), ray_indices=T(ri), outputs=got)
        assert_bitexact(N(got), want, f"accumulate C={C_}")
    got = A.accumulate_along_rays(T(w0), values=T(rgbs), ray_indices=T(ri), n_rays=n_rays)
    want = oracle.accumulate_along_rays_(w0, rgbs, packed, np.zeros((n_rays, 3), np.float32))
    assert_bitexact(N(got), want, "accumulate_along_rays")
    for eps, thre in ((1e-4, 0.0), (1e-4, 1e-2), (0.0, 0.5)):
        want = oracle.visibility_mask(t0, t1, sig, packed, eps, thre)
        got = A.render_visibility_from_density(T(t0), T(t1), T(sig), packed_info=T(packed), early_stop_eps=eps,
                                               alpha_thre=thre)
        assert_bitexact(N(got), want, f"visibility eps={eps} thre={thre}")
        assert 0 < want.mean() < 1


def test_composite_backward(oracle):
    """backward of render_weight_from_density + accumulate_along_rays (cednerf/render.py:158-169)
    against the oracle's float64 derivative, and end to end through autograd (`rendering_train`) against a
    finite-difference directional derivative."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.render import rendering_train
    packed, t0, t1, sig, rgbs = _packed_problem(3000, 21)
    S, n_rays = sig.shape[0], packed.shape[0]
    rng = np.random.default_rng(22)
    sig = (sig * rng.uniform(0.05, 2.0, size=S)).astype(np.float32)            # keep transmittances away from 0
    d_color = rng.normal(size=(n_rays, 3)).astype(np.float32)
    d_op = rng.normal(size=(n_rays,)).astype(np.float32)
    d_dp = rng.normal(size=(n_rays,)).astype(np.float32)
    w_ds, w_dc = oracle.composite_backward(packed, t0, t1, sig, rgbs, d_color, d_op, d_dp)
    ds, dc = ops.composite_backward(T(packed), T(t0), T(t1), T(sig), T(rgbs), T(d_color), T(d_op), T(d_dp))
    sc_s, sc_c = np.abs(w_ds).max(), np.abs(w_dc).max()
    assert np.abs(N(ds) - w_ds).max() <= 2e-5 * sc_s, np.abs(N(ds) - w_ds).max() / sc_s
    assert np.abs(N(dc) - w_dc).max() <= 1e-5 * sc_c
    # autograd path: d/d eps of L(sigma + eps u, rgb + eps v) == <grad, (u, v)>
    ri = np.repeat(np.arange(n_rays), packed[:, 1]).astype(np.int64)
    bk = torch.tensor([0.2, 0.5, 0.9], device=DEV)
    u = torch.randn(S, device=DEV, dtype=torch.float64) * 0.1; v = torch.randn(S, 3, device=DEV, dtype=torch.float64) * 0.1
    # (the normalised depth = depth / opacity is ill-conditioned on the near-empty rays of this random problem: its
    #  raw-depth gradient is covered by the kernel comparison above, the end-to-end check uses colour and opacity)
    wc = T(d_color); wo = T(d_op); wd = T(d_dp) * 0.0

    def loss_of(s_, c_):
        col, op, dp, _ = rendering_train(T(t0), T(t1), T(ri), n_rays, lambda a, b, c: (c_, s_), render_bkgd=bk)
        return (col * wc).sum() + (op[:, 0] * wo).sum() + (dp[:, 0] * wd).sum()

    s_t = T(sig).requires_grad_(True); c_t = T(rgbs).requires_grad_(True)
    loss_of(s_t, c_t).backward()
    analytic = (s_t.grad.double() * u).sum().item() + (c_t.grad.double() * v).sum().item()
    eps = 1e-3
    with torch.no_grad():
        lp = loss_of((T(sig).double() + eps * u).float(), (T(rgbs).double() + eps * v).float()).double().item()
        lm = loss_of((T(sig).double() - eps * u).float(), (T(rgbs).double() - eps * v).float()).double().item()
    numeric = (lp - lm) / (2 * eps)
    assert abs(numeric - analytic) <= 2e-2 * max(abs(numeric), abs(analytic), 1.0), (numeric, analytic)


def test_composite_test_kernel(oracle):
    """cednerf/taichi_kernel/volume_render_test.py:composite_test on the GPU vs its C restatement."""
    import ctypes as C
    from ced_nerf_amd import ops
    packed, t0, t1, sig, rgbs = _packed_problem(2000, 9)
    n_alive = packed.shape[0]
    rng = np.random.default_rng(10)
    n_rays = 2500
    alive = rng.permutation(n_rays)[:n_alive].astype(np.int64)
    opacity = rng.uniform(0, 0.5, size=(n_rays, 1)).astype(np.float32)
    depth = rng.uniform(0, 1, size=(n_rays, 1)).astype(np.float32)
    rgb = rng.uniform(0, 1, size=(n_rays, 3)).astype(np.float32)
    w_alive, w_op, w_dp, w_rgb = alive.copy(), opacity.copy(), depth.copy(), rgb.copy()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.lib().ced_o_composite_test(C.c_int64(n_alive), p(sig), p(rgbs), p(t0), p(t1), p(packed), p(w_alive),
                                      C.c_float(1e-2), C.c_float(1e-3), p(w_op), p(w_dp), p(w_rgb))
    g_alive, g_op, g_dp, g_rgb = T(alive), T(opacity), T(depth), T(rgb)
    ops.composite_test_(T(sig)[:, None].contiguous(), T(rgbs), T(t0)[:, None].contiguous(),
                        T(t1)[:, None].contiguous(), T(packed), g_alive, 1e-2, 1e-3, g_op, g_dp, g_rgb)
    assert_bitexact(N(g_alive), w_alive, "alive_indices")
    assert_bitexact(N(g_op), w_op, "opacity"); assert_bitexact(N(g_dp), w_dp, "depth"); assert_bitexact(N(g_rgb), w_rgb, "rgb")
    assert (w_alive == -1).any() and (w_alive != -1).any()


def _setup(oracle, sc):
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays
    cfg = sc["cfg"]
    of = oracle.OracleField(sc["params"])
    oest = oracle.OracleEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"], sc["binaries"])
    f = DNGPradianceField.from_params(sc["params"], DEV).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    rays = Rays(origins=T(sc["origins"]), viewdirs=T(sc["viewdirs"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    return of, oest, f, est, rays, rk


@pytest.mark.parametrize("name,regime,wh", [("dnerf", "trained", (80, 60)), ("dnerf", "init", (64, 48)),
                                            ("hypernerf", "trained", (48, 64)), ("dynerf", "trained", (64, 48))])
def test_render_image_test_parity(oracle, name, regime, wh):
    """a1: render_image_test end to end; per-iteration sample counts bit-exact, pixels <= 1e-4."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import render_image_test, render_image_test_staged
    sc = _scene(name, wh[0], wh[1], regime, log2_hashmap_size=17)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    max_samples = 256 if regime == "init" else 1024
    trace = []
    w_rgb, w_op, w_dp, w_total = oracle.render_image_test(max_samples, of, oest, sc["origins"], sc["viewdirs"],
                                                          timestamps=sc["timestamps"], trace=trace, **sc["render"])
    tracer = ops.FrameTracer(capacity=1100, with_events=False)
    rgb, op, dp, total = render_image_test(max_samples, f, est, rays, timestamps=T(sc["timestamps"]), tracer=tracer, **rk)
    assert rgb.shape == (wh[1], wh[0], 3) and op.shape == (wh[1], wh[0], 1) and dp.shape == (wh[1], wh[0], 1)
    assert total == w_total and total > 2000
    # the image-global schedule (N_alive, N_samples, samples marched) of every iteration, bit-exact
    assert tracer.iterations() == [dict(n_alive=t["n_alive"], n_samples=t["n_samples"], n_new=t["n_new"]) for t in trace]
    # the Python-staged loop over the nerfacc-shaped ops renders the same frame
    s_rgb, s_op, s_dp, s_total = render_image_test_staged(max_samples, f, est, rays, timestamps=T(sc["timestamps"]), **rk)
    assert s_total == total
    assert_bitexact(N(s_rgb), N(rgb), "staged vs native rgb"); assert_bitexact(N(s_dp), N(dp), "staged vs native depth")
    assert_bitexact(N(s_op), N(op), "staged vs native opacity")
    assert np.abs(N(rgb) - w_rgb).max() <= 1e-4
    assert np.abs(N(op) - w_op).max() <= 1e-4
    assert np.abs(N(dp) - w_dp).max() <= 1e-4
    assert_bitexact(N(rgb), w_rgb, "rgb (bit-exact)")
    assert_bitexact(N(dp), w_dp, "depth (bit-exact)")


def _fuzz_scene(seed):
    """A random configuration: grid resolution and level count, step size, cone angle, near / far planes, alpha
    threshold, camera convention and distance (including a camera inside the box), irregular occupancy, model flags."""
    from ced_nerf_amd import synthetic as S
    rng = np.random.default_rng(1000 + seed)
    res = int(rng.choice([32, 64, 128]))
    levels = int(rng.choice([1, 2, 3]))
    half = float(rng.choice([1.0, 1.5]))
    flags = [dict(), dict(use_div_offsets=True), dict(use_time_embedding=True),
             dict(use_time_embedding=True, use_time_attenuation=True, use_div_offsets=True)][int(rng.integers(4))]
    cfg = dict(aabb=[-half] * 3 + [half] * 3, near_plane=float(rng.choice([0.0, 0.13])),
               far_plane=float(rng.choice([1e10, 4.2])), moving_step=float(rng.choice([1e-4, 1.0 / 512])),
               hash_max_res=int(rng.choice([512, 2048])), grid_resolution=res, grid_levels=levels,
               render_step_size=float(rng.uniform(2e-3, 1.7e-2)), alpha_thre=float(rng.choice([0.0, 1e-2])),
               cone_angle=float(rng.choice([0.0, 0.0037, 0.012])), bkgd=[float(x) for x in rng.uniform(0, 1, 3)],
               opengl=bool(rng.integers(2)), camera_angle_x=float(rng.uniform(0.5, 1.1)),
               radius=float(rng.choice([0.6 * half, 1.9 * half, 2.7 * half])), flags=flags)
    S.CONFIGS["fuzz"] = cfg
    try:
        sc = S.make_scene("fuzz", 40, 30, "trained", azim_deg=float(rng.uniform(0, 360)), elev_deg=float(rng.uniform(-40, 60)),
                          seed=seed, log2_hashmap_size=15, timestamp=float(rng.uniform(0, 1)))
    finally:
        del S.CONFIGS["fuzz"]
    # irregular occupancy: sparse random cells plus the analytic shape, per level
    b = sc["binaries"].copy()
    b |= rng.uniform(size=b.shape) < 0.02
    if seed % 3 == 0:
        b[-1] = False                                        # an entirely empty (coarsest) level
    sc["binaries"] = b
    return sc, cfg, rng


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CED_FUZZ_SEEDS", "8"))))
def test_render_image_test_random_configurations(oracle, seed):
    """Fuzz over the knobs the three dataset configs do not vary together (see _fuzz_scene).  The native frame loop
    against the oracle: schedule, counts, pixels."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import render_image_test
    sc, cfg, rng = _fuzz_scene(seed)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    # every third seed in one of the fp16-MFMA modes (CED_FUZZ_MODE forces one for a one-off run): against the oracle's mode
    mode = os.environ.get("CED_FUZZ_MODE") or ("f32", "f32", "f16x2", "f32", "f32", "f16")[seed % 6]
    if mode != "f32":
        of = oracle.OracleField(sc["params"], mlp_half=mode)
        f.set_mlp_precision(mode)
    max_samples = int(rng.choice([64, 300, 1024]))
    trace = []
    w_rgb, w_op, w_dp, w_total = oracle.render_image_test(max_samples, of, oest, sc["origins"], sc["viewdirs"],
                                                          timestamps=sc["timestamps"], trace=trace, **sc["render"])
    tracer = ops.FrameTracer(capacity=1100, with_events=False)
    rgb, op, dp, total = render_image_test(max_samples, f, est, rays, timestamps=T(sc["timestamps"]), tracer=tracer, **rk)
    assert total == w_total, (cfg, total, w_total)
    assert tracer.iterations() == [dict(n_alive=t["n_alive"], n_samples=t["n_samples"], n_new=t["n_new"]) for t in trace]
    assert_bitexact(N(rgb), w_rgb, f"rgb {cfg}")
    assert_bitexact(N(dp), w_dp, f"depth {cfg}")
    assert np.abs(N(op) - w_op).max() <= 1e-6


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("CED_FUZZ_SEEDS", "6"))))
def test_render_image_random_configurations(oracle, seed):
    """The same fuzz through render_image (sampling -> visibility filter -> rendering, cednerf/utils.py:46-150)."""
    from ced_nerf_amd.utils import render_image
    sc, cfg, rng = _fuzz_scene(100 + seed)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    mode = os.environ.get("CED_FUZZ_MODE") or ("f32", "f16x2", "f32", "f32", "f16", "f32")[seed % 6]
    if mode != "f32":
        of = oracle.OracleField(sc["params"], mlp_half=mode)
        f.set_mlp_precision(mode)
    w = oracle.render_image(of, oest, sc["origins"], sc["viewdirs"], timestamps=sc["timestamps"], **sc["render"])
    g = render_image(f, est, rays, timestamps=T(sc["timestamps"]), **rk)
    assert g[3] == w[3], (cfg, g[3], w[3])
    for i, nm in enumerate(("colors", "opacities", "depths")):
        assert_bitexact(N(g[i]), w[i].reshape(N(g[i]).shape), f"{nm} {cfg}")


@pytest.mark.parametrize("prec", ["f16x2", "f16", "f32+h16x2"])
@pytest.mark.parametrize("name,wh,kw", [("dnerf", (80, 60), {}), ("hypernerf", (48, 64), {}), ("dynerf", (64, 48), {}),
                                        ("dnerf", (80, 60), {"table_dtype": np.float16})])   # last: BASELINE config 5
def test_render_image_test_half_precision(oracle, prec, name, wh, kw):
    """render_image_test with the half-precision MLP modes against the oracle's mode of the same name: the image-global
    schedule, the sample total and every pixel of rgb / opacity / depth BIT FOR BIT (f16 + fp16 table = BASELINE config 5:
    "fp16 hash features + fp16 MFMA MLP").  f16x2 also against the PLAIN fp32 oracle at the north-star bar: 1e-4 abs on
    rgb / opacity / depth and the same sample count on these frames."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import render_image_test
    sc = _scene(name, wh[0], wh[1], "trained", log2_hashmap_size=17, **kw)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    ofm = oracle.OracleField(sc["params"], mlp_half=prec)
    f.set_mlp_precision(prec)
    trace = []
    w_rgb, w_op, w_dp, w_total = oracle.render_image_test(1024, ofm, oest, sc["origins"], sc["viewdirs"],
                                                          timestamps=sc["timestamps"], trace=trace, **sc["render"])
    tracer = ops.FrameTracer(capacity=1100, with_events=False)
    rgb, op, dp, total = render_image_test(1024, f, est, rays, timestamps=T(sc["timestamps"]), tracer=tracer, **rk)
    print(f"[{prec} {name} {kw}] samples {total} vs {w_total}")
    assert total == w_total
    assert tracer.iterations() == [dict(n_alive=t["n_alive"], n_samples=t["n_samples"], n_new=t["n_new"]) for t in trace]
    for nm, g_, w_ in (("rgb", N(rgb), w_rgb), ("opacity", N(op), w_op), ("depth", N(dp), w_dp)):
        assert_bitexact(g_, w_, f"frame {prec} {name} {sorted(kw)} {nm}")
    if prec != "f16":
        p_rgb, p_op, p_dp, p_total = oracle.render_image_test(1024, of, oest, sc["origins"], sc["viewdirs"],
                                                              timestamps=sc["timestamps"], **sc["render"])
        assert total == p_total
        for nm, g_, w_ in (("rgb", N(rgb), p_rgb), ("opacity", N(op), p_op), ("depth", N(dp), p_dp)):
            assert np.abs(g_ - w_).max() <= 1e-4, f"{prec} {nm} vs fp32 oracle: {np.abs(g_ - w_).max():.2e}"
        if prec == "f32+h16x2":
            assert_bitexact(N(op), p_op, "opacity vs fp32 oracle"); assert_bitexact(N(dp), p_dp, "depth vs fp32 oracle")


@pytest.mark.parametrize("name,regime,wh,prec", [("dnerf", "trained", (80, 60), "f32"), ("hypernerf", "trained", (48, 64), "f32"),
                                                 ("dynerf", "init", (40, 30), "f32"), ("dnerf", "trained", (80, 60), "f16x2"),
                                                 ("hypernerf", "trained", (48, 64), "f16"), ("hypernerf", "trained", (48, 64), "f32+h16x2")])
def test_render_image_parity(oracle, name, regime, wh, prec):
    """a2/a3: render_image (sampling -> visibility filter -> rendering): surviving sample indices,
    t_starts/t_ends and counts bit-exact per chunk; pixels bit-exact -- in every arithmetic mode against the oracle's mode
    of the same name (the visibility filter's decisions hang on sigma: they are the oracle's only because sigma is)."""
    from ced_nerf_amd.utils import render_image
    sc = _scene(name, wh[0], wh[1], regime, log2_hashmap_size=17)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    if prec != "f32":
        of = oracle.OracleField(sc["params"], mlp_half=prec)
        f.set_mlp_precision(prec)
    chunk = 1000
    w = oracle.render_image(of, oest, sc["origins"], sc["viewdirs"], timestamps=sc["timestamps"],
                            test_chunk_size=chunk, **sc["render"])
    g = render_image(f, est, rays, timestamps=T(sc["timestamps"]), test_chunk_size=chunk, **rk)
    assert g[3] == w[3] and len(g[4]) == len(w[4])
    for ge, we in zip(g[4], w[4]):
        for k in ("ray_indices", "t_starts", "t_ends", "sigmas", "rgbs", "weights", "trans", "alphas"):
            assert_bitexact(N(ge[k]), we[k], f"extras[{k}]")
    for i, nm in enumerate(("colors", "opacities", "depths")):
        assert g[i].shape == w[i].shape
        assert np.abs(N(g[i]) - w[i]).max() <= 1e-4, nm
        assert_bitexact(N(g[i]), w[i], nm)
    if regime == "trained":
        assert w[3] < w[5]          # the visibility filter removed something


@pytest.mark.parametrize("name,wh,chunk,alpha_thre", [("dnerf", (80, 60), 1000, 0.0), ("dnerf", (80, 60), 8192, 0.0),
                                                       ("hypernerf", (48, 64), 700, 0.0), ("dnerf", (64, 48), 1000, 0.02)])
def test_render_image_native_pass_matches_staged_composition(oracle, name, wh, chunk, alpha_thre):
    """The eval pass of render_image on ced_render_image (one field evaluation per sample, rays stopped at the
    visibility threshold) against the staged composition it replaces (sampling with sigma_fn over every marched sample,
    then rendering): every returned array bit for bit, chunked and unchunked, with and without an alpha threshold."""
    from ced_nerf_amd.utils import render_image
    sc = _scene(name, wh[0], wh[1], "trained", log2_hashmap_size=17)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    rk = dict(rk); rk["alpha_thre"] = alpha_thre
    if alpha_thre > 0.0:
        est.occs = torch.full_like(est.occs, 1.0)               # nerfacc clamps alpha_thre to occs.mean()
    ts = T(sc["timestamps"])
    a = render_image(f, est, rays, timestamps=ts, test_chunk_size=chunk, native=True, **rk)
    b = render_image(f, est, rays, timestamps=ts, test_chunk_size=chunk, native=False, **rk)
    assert a[3] == b[3] and len(a[4]) == len(b[4]) and a[3] > 0
    for ae, be in zip(a[4], b[4]):
        assert set(ae) == set(be)
        for k in be:
            assert ae[k].dtype == be[k].dtype and ae[k].shape == be[k].shape, k
            assert torch.equal(ae[k], be[k]), f"extras[{k}]"
    for i, nm in enumerate(("colors", "opacities", "depths")):
        assert a[i].shape == b[i].shape and torch.equal(a[i], b[i]), nm
    if alpha_thre > 0.0:
        al = torch.cat([e["alphas"] for e in a[4]])
        assert bool((al >= alpha_thre).all())
        # render_image takes the two-pass form with a threshold; the one-pass kernel handles it too (C-ABI contract)
        from ced_nerf_amd import ops
        o = rays.origins.reshape(-1, 3).contiguous(); d = rays.viewdirs.reshape(-1, 3).contiguous()
        t0, t1, _, packed = est.march(o, d, near_plane=rk["near_plane"], far_plane=rk["far_plane"],
                                      render_step_size=rk["render_step_size"], cone_angle=rk["cone_angle"])
        one = ops.render_image_eval_native(f._descriptor(), o, d, packed, t0, t1, 1e-4, alpha_thre, ts.reshape(-1), False,
                                           rk["render_bkgd"].reshape(-1).float().contiguous())
        for i in range(3):
            assert torch.equal(one[i].reshape(b[i].shape), b[i])
        for k in ("weights", "trans", "alphas", "sigmas", "rgbs", "t_starts", "t_ends"):
            assert torch.equal(one[3][k], torch.cat([e[k] for e in b[4]])), k


@pytest.mark.parametrize("name,wh", [("hypernerf", (268, 480)), ("dynerf", (338, 254)), ("dnerf", (200, 200))])
def test_one_shot_march_matches_traverse_grids(oracle, name, wh):
    """ced_march_all on one, two and four grid levels (cone angle 0 and 0.004), common and stratified near planes,
    against ced_traverse_grids: packed info and every sample bit for bit."""
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    sc = _scene(name, wh[0], wh[1], "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    o = T(sc["origins"]).reshape(-1, 3).contiguous(); d = T(sc["viewdirs"]).reshape(-1, 3).contiguous()
    for stratified in (False, True):
        kw = dict(near_plane=cfg["near_plane"], far_plane=cfg["far_plane"], render_step_size=cfg["render_step_size"],
                  cone_angle=cfg["cone_angle"], stratified=stratified)
        torch.manual_seed(11)
        a = est.march(o, d, fast=True, **kw)
        torch.manual_seed(11)
        b = est.march(o, d, fast=False, **kw)
        assert b[0].shape[0] > 10000
        for x, y, nm in zip(a, b, ("t_starts", "t_ends", "ray_indices", "packed_info")):
            assert x.dtype == y.dtype and torch.equal(x, y), (name, nm, stratified)


@pytest.mark.parametrize("name,wh", [("dnerf", (200, 200)), ("dynerf", (169, 127))])
def test_render_image_one_pass_march_and_its_fallback(oracle, name, wh):
    """The second eval render_image of a given size marches in ONE pass into arrays sized by the previous total (rays in
    workgroup arrival order); a frame that does not fit is redone with the exact march.  Same arrays every time."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import render_image
    sc = _scene(name, wh[0], wh[1], "trained", log2_hashmap_size=15)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    rk = dict(rk); rk["alpha_thre"] = 0.0
    ts = T(sc["timestamps"])
    n = wh[0] * wh[1]
    first = render_image(f, est, rays, timestamps=ts, native=True, **rk)
    total = est._march_totals[n]
    assert total > 100000
    second = render_image(f, est, rays, timestamps=ts, native=True, **rk)               # one pass, capacity 1.25 x total
    est._march_totals[n] = 10                                                            # capacity far too small
    third = render_image(f, est, rays, timestamps=ts, native=True, **rk)                # overflow -> exact march
    assert est._march_totals[n] == total
    staged = render_image(f, est, rays, timestamps=ts, native=False, **rk)
    for other in (second, third, staged):
        assert other[3] == first[3] and len(other[4]) == len(first[4])
        for i in range(3):
            assert torch.equal(other[i], first[i])
        for a, b in zip(other[4], first[4]):
            for k in b:
                assert torch.equal(a[k], b[k]), k
    # the one-pass march itself: per-ray sample sets equal to the two-pass march's, ranges disjoint and within the total
    o = rays.origins.reshape(-1, 3).contiguous(); d = rays.viewdirs.reshape(-1, 3).contiguous()
    t0, t1, packed, tot = est.march_onepass(o, d, rk["near_plane"], rk["far_plane"], rk["render_step_size"],
                                            rk["cone_angle"], total + 1000)
    e0, e1, _, epacked = est.march(o, d, near_plane=rk["near_plane"], far_plane=rk["far_plane"],
                                   render_step_size=rk["render_step_size"], cone_angle=rk["cone_angle"])
    assert int(tot.item()) == total == e0.shape[0] and torch.equal(packed[:, 1], epacked[:, 1])
    order = torch.argsort(packed[:, 0] + (packed[:, 1] == 0) * (1 << 40))
    st, ct = packed[order, 0], packed[order, 1]
    nz = ct > 0
    assert bool((st[nz][1:] == (st[nz] + ct[nz])[:-1]).all()) and int(st[nz][0]) == 0
    idx = torch.repeat_interleave(packed[:, 0], epacked[:, 1]) + (torch.arange(total, device=DEV) - torch.repeat_interleave(epacked[:, 0], epacked[:, 1]))
    assert torch.equal(t0[idx], e0) and torch.equal(t1[idx], e1)


def test_render_image_several_internal_passes(oracle, monkeypatch):
    """render_image cuts a large eval frame into internal passes of whole chunks: with the pass size forced down to two
    chunks the output (pixels, the per-chunk extras list) is that of one pass, on both paths, and equals the oracle's."""
    from ced_nerf_amd import utils as U
    sc = _scene("dnerf", 80, 60, "trained", log2_hashmap_size=17)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    ts = T(sc["timestamps"])
    one = U.render_image(f, est, rays, timestamps=ts, test_chunk_size=1000, native=True, **rk)
    monkeypatch.setattr(U, "_EVAL_PASS_RAYS", 2000)
    for native in (True, False):
        many = U.render_image(f, est, rays, timestamps=ts, test_chunk_size=1000, native=native, **rk)
        assert many[3] == one[3] and len(many[4]) == len(one[4]) == 5
        for i in range(3):
            assert torch.equal(many[i], one[i])
        for a, b in zip(many[4], one[4]):
            for k in b:
                assert torch.equal(a[k], b[k]), (native, k)
    w = oracle.render_image(of, oest, sc["origins"], sc["viewdirs"], timestamps=sc["timestamps"], test_chunk_size=1000,
                            **sc["render"])
    assert one[3] == w[3]
    for ge, we in zip(one[4], w[4]):
        assert_bitexact(N(ge["ray_indices"]), we["ray_indices"], "ray_indices")
        assert_bitexact(N(ge["weights"]), we["weights"], "weights")


@pytest.mark.parametrize("prec,table", [("f16x2", np.float32), ("f16", np.float16)])
def test_render_image_native_pass_in_the_half_modes(oracle, prec, table):
    """The native render_image pass on the half-precision field kernels (their per-sample arrays are addressed through
    the same device-side window): identical to the staged composition in the same arithmetic mode, bit for bit."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays, render_image
    sc = S.make_scene("dnerf", 160, 120, "trained", table_dtype=table, log2_hashmap_size=17)
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], DEV, mlp_precision=prec).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    rays = Rays(T(sc["origins"]), T(sc["viewdirs"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    ts = T(sc["timestamps"])
    a = render_image(f, est, rays, timestamps=ts, native=True, **rk)
    a2 = render_image(f, est, rays, timestamps=ts, native=True, **rk)            # one-pass march this time
    b = render_image(f, est, rays, timestamps=ts, native=False, **rk)
    for other in (a2, b):
        assert a[3] == other[3] > 10000 and len(a[4]) == len(other[4])
        for i in range(3):
            assert torch.equal(a[i], other[i])
        for x, y in zip(a[4], other[4]):
            for k in y:
                assert torch.equal(x[k], y[k]), (prec, k)


def test_render_image_native_pass_empty_and_ragged(oracle):
    """Rays that miss the grid (no samples), a pass smaller than a chunk, and zero rays."""
    from ced_nerf_amd.utils import Rays, render_image
    sc = _scene("dnerf", 40, 30, "trained", log2_hashmap_size=17)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    ts = T(sc["timestamps"])
    o = rays.origins.reshape(-1, 3).clone(); d = rays.viewdirs.reshape(-1, 3).clone()
    d[::3] = -d[::3]                                               # a third of the rays look away from the box
    r2 = Rays(o[:777].contiguous(), d[:777].contiguous())
    a = render_image(f, est, r2, timestamps=ts, native=True, **rk)
    b = render_image(f, est, r2, timestamps=ts, native=False, **rk)
    assert a[3] == b[3] and len(a[4]) == len(b[4]) == 1
    for k in b[4][0]:
        assert torch.equal(a[4][0][k], b[4][0][k]), k
    for i in range(3):
        assert torch.equal(a[i], b[i])
    away = Rays(o[::3][:50].contiguous(), d[::3][:50].contiguous())
    e = render_image(f, est, away, timestamps=ts, native=True, **rk)
    g = render_image(f, est, away, timestamps=ts, native=False, **rk)
    assert e[3] == g[3] and all(torch.equal(e[i], g[i]) for i in range(3))


# ---- full-size checks (BASELINE.json config 2: 800x800, T = 2^21) ---------------------------------
@pytest.fixture(scope="module")
def full_frame(oracle):
    sc = _scene("dnerf", 800, 800, "trained")
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    return sc, of, oest, f, est, rays, rk


def test_full_size_one_shot_march_on_the_accelerated_walk(oracle, full_frame):
    """ced_march_all (sphere-traced one-shot march, one grid level) against ced_traverse_grids on all 640 000 rays of
    config 2: counts, starts and every sample bit for bit -- with the frame's near plane and with per-ray (stratified)
    near planes, as the training step marches."""
    sc, of, oest, f, est, rays, rk = full_frame
    o = rays.origins.reshape(-1, 3).contiguous(); d = rays.viewdirs.reshape(-1, 3).contiguous()
    for stratified in (False, True):
        torch.manual_seed(3)
        a = est.march(o, d, near_plane=rk["near_plane"], far_plane=rk["far_plane"], render_step_size=rk["render_step_size"],
                      cone_angle=rk["cone_angle"], stratified=stratified, fast=True)
        torch.manual_seed(3)
        b = est.march(o, d, near_plane=rk["near_plane"], far_plane=rk["far_plane"], render_step_size=rk["render_step_size"],
                      cone_angle=rk["cone_angle"], stratified=stratified, fast=False)
        assert a[0].shape[0] > 10_000_000
        for x, y, nm in zip(a, b, ("t_starts", "t_ends", "ray_indices", "packed_info")):
            assert x.dtype == y.dtype and torch.equal(x, y), (nm, stratified)


def test_field_kernel_device_stamps(oracle, full_frame):
    """ced_frame_trace.field_stamps: the field kernel's own {first workgroup in, last workgroup out} stamps lie inside
    the HIP event pair recorded around the same launch and add up to most of it when the frame is alone on the chip."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import render_image_test
    sc, of, oest, f, est, rays, rk = full_frame
    ts = T(sc["timestamps"])
    assert ops._lib.lib().ced_wall_clock_khz() > 1000
    tracer = ops.FrameTracer(capacity=64, with_events=True)
    tracer.enable_device_stamps(torch.device(DEV))
    for _ in range(2):
        render_image_test(1024, f, est, rays, timestamps=ts, tracer=tracer, **rk)
    torch.cuda.synchronize()
    ev = tracer.field_ms()
    dv = tracer.field_intervals_device()
    assert len(ev) == len(dv) >= 8 and all(x is not None for x in dv)
    d_ms = [e - b for b, e in dv]
    assert all(0.0 < d <= e * 1.02 + 0.01 for d, e in zip(d_ms, ev)), list(zip(d_ms, ev))
    assert all(dv[i + 1][0] >= dv[i][1] for i in range(len(dv) - 1))            # launches of one stream, in order
    assert sum(d_ms) > 0.8 * sum(ev)


def test_full_size_render_image_native_pass(oracle, full_frame):
    """800x800 (BASELINE config 2): render_image through ced_render_image equals the staged composition in every pixel
    and every per-sample array, and evaluates the field on far fewer samples than the march holds."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import render_image
    sc, of, oest, f, est, rays, rk = full_frame
    ts = T(sc["timestamps"])
    a = render_image(f, est, rays, timestamps=ts, native=True, **rk)
    b = render_image(f, est, rays, timestamps=ts, native=False, **rk)
    assert a[3] == b[3] and len(a[4]) == len(b[4]) == (640000 + 8191) // 8192
    for i in range(3):
        assert torch.equal(a[i], b[i])
    for c in (0, 17, len(a[4]) // 2, len(a[4]) - 1):
        for k in b[4][c]:
            assert torch.equal(a[4][c][k], b[4][c][k]), (c, k)
    cat = lambda lst, k: torch.cat([e[k] for e in lst])
    for k in ("weights", "rgbs", "ray_indices", "t_starts"):
        assert torch.equal(cat(a[4], k), cat(b[4], k)), k
    # work: one field evaluation per processed sample; the march holds several times the kept samples
    o = rays.origins.reshape(-1, 3).contiguous(); d = rays.viewdirs.reshape(-1, 3).contiguous()
    t0, t1, _, packed = est.march(o, d, near_plane=rk.get("near_plane", 0.0), far_plane=rk.get("far_plane", 1e10),
                                  render_step_size=rk["render_step_size"], cone_angle=rk.get("cone_angle", 0.0))
    out = ops.render_image_eval_native(f._descriptor(), o, d, packed, t0, t1, 1e-4, 0.0, ts.reshape(-1), False,
                                       rk["render_bkgd"].reshape(-1).float().contiguous())
    processed, iters = out[5]
    assert a[3] <= processed <= int(1.35 * a[3]) and processed < t0.shape[0] // 2, (a[3], processed, t0.shape[0])
    assert int(out[4][-1]) == a[3] and iters < 64


def test_full_size_frame_properties(oracle, full_frame):
    """800x800 render_image_test: size-independent properties + an oracle spot check."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import Rays, render_image, render_image_test, render_image_test_staged
    sc, of, oest, f, est, rays, rk = full_frame
    ts = T(sc["timestamps"])
    tracer = ops.FrameTracer(capacity=1100, with_events=False)
    rgb, op, dp, total = render_image_test(1024, f, est, rays, timestamps=ts, tracer=tracer, **rk)
    its = tracer.iterations()
    assert sum(i["n_new"] for i in its) == total and its[0]["n_alive"] == 640000 and its[0]["n_samples"] == 1
    assert all(i["n_samples"] == max(min(640000 // i["n_alive"], 64), 1) for i in its)        # utils.py:235
    assert all(i["n_new"] <= i["n_alive"] * i["n_samples"] for i in its)
    # idempotence / determinism: a second render and the Python-staged loop give the same bits
    rgb2, op2, dp2, total2 = render_image_test(1024, f, est, rays, timestamps=ts, **rk)
    assert total2 == total and torch.equal(rgb, rgb2) and torch.equal(dp, dp2)
    s_rgb, s_op, s_dp, s_total = render_image_test_staged(1024, f, est, rays, timestamps=ts, **rk)
    assert s_total == total and torch.equal(s_rgb, rgb) and torch.equal(s_op, op) and torch.equal(s_dp, dp)
    # linearity in the background colour: rgb(white) - rgb(black) == 1 - opacity
    rk_black = dict(rk); rk_black["render_bkgd"] = torch.zeros(3, device=DEV)
    rgb_b, op_b, _, _ = render_image_test(1024, f, est, rays, timestamps=ts, **rk_black)
    assert torch.equal(op_b, op)
    assert (rgb - rgb_b - (1.0 - op)).abs().max().item() <= 1e-6
    o, opn = N(op), N(op)
    assert opn.min() >= 0.0 and opn.max() <= 1.0 + 1e-5
    miss = opn[..., 0] == 0
    assert miss.mean() > 0.5 and np.all(N(rgb)[miss] == 1.0) and np.all(N(dp)[miss] == 0.0)
    hit_depth = N(dp)[~miss]
    assert hit_depth.min() > 1.3 and hit_depth.max() < 6.7          # camera at radius 4, aabb half-diagonal 2.6
    # oracle spot check on a random subset of rays of the same frame, via the per-ray-deterministic
    # render_image (the schedule of render_image_test is image-global, a subset would change it)
    rng = np.random.default_rng(0)
    hit_idx = np.flatnonzero(~miss.reshape(-1))
    pick = np.concatenate([rng.choice(hit_idx, 1500, replace=False), rng.choice(640000, 500, replace=False)])
    so = sc["origins"].reshape(-1, 3)[pick]; sd = sc["viewdirs"].reshape(-1, 3)[pick]
    w = oracle.render_image(of, oest, so, sd, timestamps=sc["timestamps"], **sc["render"])
    g = render_image(f, est, Rays(T(so), T(sd)), timestamps=ts, **rk)
    assert g[3] == w[3] and w[3] > 10000
    for i, nm in enumerate(("colors", "opacities", "depths")):
        assert_bitexact(N(g[i]), w[i], f"full-size subset {nm}")
    # and the full-frame render_image agrees with itself when the image is cut in two (per-ray determinism)
    full = render_image(f, est, rays, timestamps=ts, **rk)
    top = render_image(f, est, Rays(rays.origins[:400].contiguous(), rays.viewdirs[:400].contiguous()), timestamps=ts, **rk)
    assert torch.equal(full[0][:400], top[0]) and torch.equal(full[2][:400], top[2])
    assert len(full[4]) == (640000 + 8191) // 8192 and sum(len(e["t_starts"]) for e in full[4]) == full[3]
    assert torch.equal(full[0].reshape(-1, 3)[torch.from_numpy(pick).to(DEV)], g[0])


@pytest.mark.parametrize("name,wh,min_samples", [("hypernerf", (536, 960), 4), ("dynerf", (1352, 1014), 4)])
def test_full_size_other_configs(oracle, name, wh, min_samples):
    """BASELINE configs 3 and 4 at their full sizes (HyperNeRF 536x960 with -te -ta -df, 2 grid levels, cone 0.004;
    DyNeRF 1352x1014, 4 levels): size-independent properties of render_image_test and an oracle check of a ray
    subset through the per-ray-deterministic render_image."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.utils import Rays, render_image, render_image_test, render_image_test_staged
    sc = _scene(name, wh[0], wh[1], "trained")
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    ts = T(sc["timestamps"])
    n_rays = wh[0] * wh[1]
    tracer = ops.FrameTracer(capacity=1100, with_events=False)
    rgb, op, dp, total = render_image_test(1024, f, est, rays, timestamps=ts, tracer=tracer, **rk)
    its = tracer.iterations()
    assert sum(i["n_new"] for i in its) == total and its[0]["n_alive"] == n_rays
    assert all(i["n_samples"] == max(min(n_rays // i["n_alive"], 64), min_samples) for i in its)      # utils.py:235
    assert all(i["n_new"] <= i["n_alive"] * i["n_samples"] for i in its)
    rgb2, op2, dp2, total2 = render_image_test(1024, f, est, rays, timestamps=ts, **rk)
    assert total2 == total and torch.equal(rgb, rgb2) and torch.equal(dp, dp2) and torch.equal(op, op2)
    s_rgb, s_op, s_dp, s_total = render_image_test_staged(1024, f, est, rays, timestamps=ts, **rk)
    assert s_total == total and torch.equal(s_rgb, rgb) and torch.equal(s_op, op) and torch.equal(s_dp, dp)
    opn = N(op)
    assert opn.min() >= 0.0 and opn.max() <= 1.0 + 1e-5 and (opn > 0.5).mean() > 0.01
    # oracle on a subset of rays
    rng = np.random.default_rng(1)
    hit_idx = np.flatnonzero(opn.reshape(-1) > 0)
    pick = np.concatenate([rng.choice(hit_idx, 600, replace=False), rng.choice(n_rays, 300, replace=False)])
    so = sc["origins"].reshape(-1, 3)[pick]; sd = sc["viewdirs"].reshape(-1, 3)[pick]
    w = oracle.render_image(of, oest, so, sd, timestamps=sc["timestamps"], **sc["render"])
    g = render_image(f, est, Rays(T(so), T(sd)), timestamps=ts, **rk)
    assert g[3] == w[3] and w[3] > 100
    for i, nm in enumerate(("colors", "opacities", "depths")):
        assert_bitexact(N(g[i]), w[i], f"{name} full-size subset {nm}")


def test_full_size_frames_f16x2_match_exact_mode(oracle, full_frame):
    """The frames bench.py times (800x800 turntable, azimuth 30/42/54 deg), rendered with the split-fp16 MLPs,
    against the exact fp32 mode (itself bit-identical to the oracle: tests above): pixels within the north-star
    1e-4 (2e-4 on opacity, whose early-stop cut is itself 1e-4 wide).  The sample count may differ where a ray's
    transmittance lands within rounding of the early-stop threshold -- a handful of samples in 4.7 M (6 on the
    first frame), bounded here at 1e-5 relative; the exact mode is the one that reproduces the counts bit for bit."""
    import copy
    from ced_nerf_amd import ops, synthetic as S
    from ced_nerf_amd.utils import Rays, render_image_test
    sc, of, oest, f, est, rays, rk = full_frame
    ts = T(sc["timestamps"])
    cfg = sc["cfg"]
    fh = copy.deepcopy(f).set_mlp_precision("f16x2")
    for azim in (30.0, 42.0, 54.0):
        c2w = S.look_at_c2w(cfg["radius"], 30.0, azim, cfg["opengl"])
        o, d = S.make_camera_rays(800, 800, cfg["camera_angle_x"], c2w, cfg["opengl"])
        r = Rays(T(o), T(d))
        tr_a, tr_b = ops.FrameTracer(capacity=1100, with_events=False), ops.FrameTracer(capacity=1100, with_events=False)
        a = render_image_test(1024, f, est, r, timestamps=ts, tracer=tr_a, **rk)
        b = render_image_test(1024, fh, est, r, timestamps=ts, tracer=tr_b, **rk)
        errs = [(b[i] - a[i]).abs().max().item() for i in range(3)]
        print(f"[f16x2 800x800 azim {azim}] rgb {errs[0]:.2e} opacity {errs[1]:.2e} depth {errs[2]:.2e} samples {b[3]} vs {a[3]}")
        assert abs(b[3] - a[3]) <= a[3] * 1e-5
        assert errs[0] <= 1e-4 and errs[2] <= 1e-4 and errs[1] <= 2e-4


def test_sharded_renderer_collective_path_on_gpu(oracle, full_frame):
    """The shard -> render -> all-gather -> un-permute path on device tensors over RCCL (single-rank
    group: the collective and the permutation logic are the same code that runs with N ranks)."""
    import torch.distributed as dist
    from ced_nerf_amd.dist import ShardedRenderer
    from ced_nerf_amd.utils import render_image_test
    sc, of, oest, f, est, rays, rk = full_frame
    created = False
    if not dist.is_initialized():
        import os
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        created = True
    try:
        ts = T(sc["timestamps"])
        o = rays.origins[None, :256, :384].contiguous(); d = rays.viewdirs[None, :256, :384].contiguous()
        r = ShardedRenderer(f, est, 1, 0, torch.device(DEV), max_samples=1024, render_kwargs=rk, force_collective=True)
        r.set_rays(o, d)
        out = r.render(ts)
        # the sharded call (ced_render_frames_test_sharded): the per-iteration survivor counts go through an RCCL
        # all-reduce on the rendering stream (ops.ScheduleExchange) before the scheduling launch reads them
        from ced_nerf_amd.utils import Rays
        want = render_image_test(1024, f, est, Rays(o[0], d[0]), timestamps=ts, **rk)
        assert out["total_samples"] == out["local_samples"] == want[3] and want[3] > 1000
        assert r.sharded and r.exchange is not None and r.exchange.calls >= 3
        assert torch.equal(out["rgb"][0], want[0]) and torch.equal(out["depth"][0], want[2])
        assert torch.equal(out["opacity"][0], want[1])
        # frames in flight with the gathers on their own stream and no read-back (what bench.py does with N > 1
        # ranks): two steps back to back, then wait -- every step's image equals the synchronous one
        from ced_nerf_amd.dist import PipelinedRenderer
        lanes = []
        for k in range(2):
            lr = ShardedRenderer(f, est, 1, 0, torch.device(DEV), max_samples=1024, render_kwargs=rk, force_collective=True)
            lr.set_rays(o, d)
            lanes.append(lr)
        pipe = PipelinedRenderer(lanes, async_gather=True)
        steps = [pipe.render(ts) for _ in range(3)]
        pipe.wait_gathers()
        torch.cuda.synchronize()
        for outs in steps:
            for a in outs:
                assert a["total_samples"] is None and int(a["total_samples_tensor"].item()) == want[3]
                assert a["local_samples"] == want[3]
                assert torch.equal(a["rgb"], out["rgb"]) and torch.equal(a["depth"], out["depth"])
                assert torch.equal(a["opacity"], out["opacity"])
    finally:
        if created:
            dist.destroy_process_group()


def test_pipelined_frames_equal_sequential_frames(oracle, full_frame):
    """Three frames in flight (own stream + host thread each) give, frame by frame, the bits of
    rendering them one after the other: every lane is a complete render_image_test call."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.dist import PipelinedRenderer, ShardedRenderer
    from ced_nerf_amd.utils import Rays, render_image_test
    sc, of, oest, f, est, rays, rk = full_frame
    cfg = sc["cfg"]
    ts = T(sc["timestamps"])
    W, H = 320, 240
    lanes, singles = [], []
    for k in range(3):
        c2w = S.look_at_c2w(cfg["radius"], 30.0, 20.0 + 25.0 * k, cfg["opengl"])
        o, d = S.make_camera_rays(W, H, cfg["camera_angle_x"], c2w, cfg["opengl"])
        r = ShardedRenderer(f, est, 1, 0, torch.device(DEV), max_samples=1024, render_kwargs=rk)
        r.set_rays(T(o)[None], T(d)[None])
        lanes.append(r)
        singles.append(render_image_test(1024, f, est, Rays(T(o), T(d)), timestamps=ts, **rk))
    # the same frames with the rays walked in 8x8-tile order and the pixels un-permuted (bench.py's layout at one
    # GPU): the image-global schedule does not depend on the ray order, so the bits are the same
    for k, r in enumerate(lanes):
        rt = ShardedRenderer(f, est, 1, 0, torch.device(DEV), max_samples=1024, render_kwargs=rk, tile_order=True)
        rt.set_rays(r.local_o.view(1, H, W, 3), r.local_d.view(1, H, W, 3))
        out = rt.render(ts)
        assert out["total_samples"] == singles[k][3]
        assert torch.equal(out["rgb"][0], singles[k][0]) and torch.equal(out["depth"][0], singles[k][2])
        assert torch.equal(out["opacity"][0], singles[k][1])
    pipe = PipelinedRenderer(lanes)
    for _ in range(3):                                   # repeat: concurrency bugs are intermittent
        outs = pipe.render(ts)
        torch.cuda.synchronize()
        for out, want in zip(outs, singles):
            assert out["total_samples"] == want[3] and want[3] > 10000
            assert torch.equal(out["rgb"][0], want[0]) and torch.equal(out["opacity"][0], want[1])
            assert torch.equal(out["depth"][0], want[2])
    # the same as a stream of steps (lanes do not wait for each other between steps)
    for row in pipe.render_steps(ts, 4):
        torch.cuda.synchronize()
        for out, want in zip(row, singles):
            assert out["total_samples"] == want[3]
            assert torch.equal(out["rgb"][0], want[0]) and torch.equal(out["depth"][0], want[2])
    shared = PipelinedRenderer(lanes, share_field_stream=True)
    outs = shared.render(ts)
    torch.cuda.synchronize()
    for out, want in zip(outs, singles):
        assert out["total_samples"] == want[3] and torch.equal(out["rgb"][0], want[0])


@pytest.mark.parametrize("name,max_samples", [("dnerf", 1024), ("dnerf", 20), ("hypernerf", 1024), ("dynerf", 64)])
def test_render_frames_test_equals_frames_alone(oracle, name, max_samples):
    """ced_render_frames_test: several frames (different cameras and times, one of them looking away from the scene)
    share the launches of an iteration, each on its own schedule -- every frame's pixels and sample count are exactly
    those of render_image_test on that frame alone.  max_samples = 20 / 64 end the frames' loops by the sample budget
    (at different iterations) instead of by running out of rays."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays, render_frames_test, render_image_test
    W, H = 96, 72
    sc = _scene(name, W, H, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], DEV).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    frames = []
    for k, (elev, az, radius) in enumerate([(30.0, 10.0, 1.0), (55.0, 140.0, 0.8), (5.0, 250.0, 1.6), (30.0, 40.0, 1.0),
                                            (80.0, 300.0, 1.2)]):
        c2w = S.look_at_c2w(cfg["radius"] * radius, elev, az, cfg["opengl"])
        if k == 3:
            c2w = c2w.copy(); c2w[:3, :3] = -c2w[:3, :3]                   # looks away: no sample at all
        o, d = S.make_camera_rays(W, H, cfg["camera_angle_x"], c2w, cfg["opengl"])
        frames.append((o, d, np.float32(k / 4.0)))
    alone = [render_image_test(max_samples, f, est, Rays(T(o), T(d)), timestamps=torch.tensor([[t]], device=DEV), **rk)
             for o, d, t in frames]
    assert alone[3][3] == 0 and alone[0][3] != alone[1][3]
    for n in (1, 2, 5):
        rays = Rays(torch.stack([T(o) for o, _, _ in frames[:n]]), torch.stack([T(d) for _, d, _ in frames[:n]]))
        ts = torch.tensor([t for _, _, t in frames[:n]], device=DEV)
        for _ in range(2):                                             # the workspace is reused: no state may leak
            rgb, op, dp, totals = render_frames_test(max_samples, f, est, rays, timestamps=ts, **rk)
            torch.cuda.synchronize()
            assert rgb.shape == (n, H, W, 3) and op.shape == (n, H, W, 1) and len(totals) == n
            for k in range(n):
                assert totals[k] == alone[k][3], (n, k, totals, [a[3] for a in alone])
                assert torch.equal(rgb[k], alone[k][0]) and torch.equal(op[k], alone[k][1]) and torch.equal(dp[k], alone[k][2])
    assert sum(a[3] for a in alone) > 5000
    with pytest.raises(ValueError):
        many = Rays(torch.zeros(65, 4, 4, 3, device=DEV), torch.ones(65, 4, 4, 3, device=DEV))
        render_frames_test(64, f, est, many, timestamps=torch.zeros(65, device=DEV), **rk)


@pytest.mark.parametrize("m", [1, 2, 3, 4, 8])
def test_sort_intersections_is_torch_stable_sort(oracle, m):
    """ced_sort_intersections (the event list of cednerf/utils.py:219-225 in one launch) against
    torch.sort(cat([t_mins, t_maxs], -1), stable=True): values and indices, with ties (rays that miss a level carry +inf
    twice; equal entry times of nested boxes), NaN keys and real ray/box intervals."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.nerfacc_api import ray_aabb_intersect, sort_intersections
    rng = np.random.default_rng(40 + m)
    n = 50001
    tmin = rng.uniform(0, 5, size=(n, m)).astype(np.float32)
    tmax = (tmin + rng.uniform(0, 3, size=(n, m))).astype(np.float32)
    tmin[::7] = np.inf; tmax[::7] = np.inf                       # misses
    tmin[1::11, :] = tmin[1::11, :1]                              # ties between levels
    tmax[2::13, 0] = tmin[2::13, m - 1]                           # an exit equal to an entry
    tmin[5::1001, 0] = np.nan
    a, b = T(tmin), T(tmax)
    ts, ti = ops.sort_intersections(a, b)
    want_s, want_i = torch.sort(torch.cat([a, b], -1), dim=-1, stable=True)
    assert ti.dtype == torch.int64 and torch.equal(ti, want_i)
    assert torch.equal(torch.nan_to_num(ts, nan=-1.0), torch.nan_to_num(want_s, nan=-1.0))
    # through the nerfacc-shaped helper on real intersections
    o = T(rng.uniform(-3, 3, size=(n, 3)).astype(np.float32)); d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    aabbs = T(np.stack([[-(2.0 ** k)] * 3 + [2.0 ** k] * 3 for k in range(m)]).astype(np.float32))
    t0, t1, _ = ray_aabb_intersect(o, T(d.astype(np.float32)), aabbs)
    ts2, ti2 = sort_intersections(t0, t1)
    w_s, w_i = torch.sort(torch.cat([t0, t1], -1), dim=-1, stable=True)
    assert torch.equal(ti2, w_i if m > 1 else ti2) and torch.equal(ts2, w_s)


@pytest.mark.parametrize("name", ["dnerf", "hypernerf", "dynerf"])
def test_first_iteration_forms_give_the_same_frames(oracle, name):
    """The first marching iteration has two forms -- one pass; culling pass + candidate list -- selected by
    ced_set_option.  Both render the same frames bit for bit (several frames per call, a frame that sees nothing,
    a sample budget that ends the loop early)."""
    from ced_nerf_amd import _lib, synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays, render_frames_test
    W, H = 160, 120
    sc = _scene(name, W, H, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], DEV).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    os_, ds_ = [], []
    for k, (elev, az, radius) in enumerate([(30.0, 10.0, 1.0), (55.0, 140.0, 0.8), (30.0, 40.0, 1.0), (5.0, 250.0, 1.6)]):
        c2w = S.look_at_c2w(cfg["radius"] * radius, elev, az, cfg["opengl"])
        if k == 2:
            c2w = c2w.copy(); c2w[:3, :3] = -c2w[:3, :3]
        o, d = S.make_camera_rays(W, H, cfg["camera_angle_x"], c2w, cfg["opengl"])
        os_.append(T(o)); ds_.append(T(d))
    rays = Rays(torch.stack(os_), torch.stack(ds_))
    ts = torch.tensor([0.0, 0.3, 0.6, 1.0], device=DEV)
    L = _lib.lib()
    outs = {}
    try:
        for form, two in {"one pass": 0, "cull + list": 1}.items():
            assert L.ced_set_option(b"march_two_pass", two) == 0
            for ms in (1024, 24):
                outs[(form, ms)] = render_frames_test(ms, f, est, rays, timestamps=ts, **rk)
                torch.cuda.synchronize()
    finally:
        L.ced_set_option(b"march_two_pass", -1)
    for ms in (1024, 24):
        ref = outs[("one pass", ms)]
        assert sum(ref[3]) > 20000 and ref[3][2] == 0
        for form in ("cull + list",):
            got = outs[(form, ms)]
            assert list(got[3]) == list(ref[3]), (form, ms, got[3], ref[3])
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2]), (form, ms)


def test_sharded_renderer_units_equal_frames_alone(oracle):
    """ShardedRenderer(units=3) (bench.py's default at one GPU: three frames per native call, rays in 8x8-tile order,
    pixels un-permuted): every frame equals render_image_test on it alone; also through frames in flight."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.dist import PipelinedRenderer, ShardedRenderer
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays, render_image_test
    W, H = 96, 72
    sc = _scene("dnerf", W, H, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], DEV).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    ts = T(sc["timestamps"])
    lanes, alone = [], []
    for l in range(2):
        os_, ds_ = [], []
        for k in range(3):
            c2w = S.look_at_c2w(cfg["radius"], 30.0, 20.0 + 25.0 * (3 * l + k), cfg["opengl"])
            o, d = S.make_camera_rays(W, H, cfg["camera_angle_x"], c2w, cfg["opengl"])
            os_.append(T(o)); ds_.append(T(d))
            alone.append(render_image_test(1024, f, est, Rays(T(o), T(d)), timestamps=ts, **rk))
        r = ShardedRenderer(f, est, 1, 0, torch.device(DEV), max_samples=1024, render_kwargs=rk, tile_order=True, units=3)
        r.set_rays(torch.stack(os_), torch.stack(ds_))
        lanes.append(r)
    out = lanes[0].render(ts)
    assert out["total_samples"] == sum(a[3] for a in alone[:3])
    for k in range(3):
        assert torch.equal(out["rgb"][k], alone[k][0]) and torch.equal(out["depth"][k], alone[k][2])
    pipe = PipelinedRenderer(lanes)
    for row in pipe.render_steps(ts, 3):
        torch.cuda.synchronize()
        for l, o in enumerate(row):
            for k in range(3):
                assert torch.equal(o["rgb"][k], alone[3 * l + k][0]) and torch.equal(o["opacity"][k], alone[3 * l + k][1])
    with pytest.raises(ValueError):
        bad = ShardedRenderer(f, est, 1, 0, torch.device(DEV), render_kwargs=rk, units=2)
        bad.set_rays(torch.stack(os_), torch.stack(ds_))                       # 3 frames do not split into 2 groups


@pytest.mark.parametrize("name,world,n_frames,wh", [("dnerf", 3, 4, (72, 56)), ("hypernerf", 2, 3, (48, 80)),
                                                    ("dynerf", 4, 2, (88, 64))])
def test_sharded_frames_on_the_image_global_schedule(oracle, name, world, n_frames, wh):
    """ced_render_frames_test_sharded, `world` ranks emulated by `world` threads of this process (own stream each, the
    exchange step a host-side sum behind a thread barrier): frames dealt in 8x8 tiles (uneven, padded shares; frames
    that finish at different iterations; cone-angle and multi-level scenes), every frame on the loop of the WHOLE image.
    The re-assembled frames and the per-frame sample totals are those of render_image_test on each whole frame, bit for
    bit, and every rank made the same number of exchange calls (the loop ends on a rank-independent plan)."""
    import threading
    from ced_nerf_amd import ops, synthetic as S
    from ced_nerf_amd.dist import ShardedRenderer
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays, render_image_test
    W, H = wh
    sc = _scene(name, W, H, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], DEV).eval()
    f._descriptor()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    est.occupancy_accel()
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    os_, ds_, alone = [], [], []
    times = torch.linspace(0.1, 0.9, n_frames, device=DEV)
    for k in range(n_frames):
        c2w = S.look_at_c2w(cfg["radius"] * (1.0 + 0.15 * k), 30.0, 20.0 + 40.0 * k, cfg["opengl"])
        o, d = S.make_camera_rays(W, H, cfg["camera_angle_x"], c2w, cfg["opengl"])
        os_.append(T(o)); ds_.append(T(d))
        alone.append(render_image_test(96, f, est, Rays(T(o), T(d)), timestamps=times[k:k + 1], **rk))
    torch.cuda.synchronize()
    acc, lock, barrier = {}, threading.Lock(), threading.Barrier(world)
    ranks = []
    for r in range(world):
        sr = ShardedRenderer(f, est, world, r, torch.device(DEV), max_samples=96, render_kwargs=rk)
        sr.set_rays(torch.stack(os_), torch.stack(ds_))
        n_calls = [0]

        def reduce_fn(row, n_calls=n_calls):
            torch.cuda.current_stream().synchronize()
            with lock:
                acc[n_calls[0]] = acc.get(n_calls[0], 0) + row.clone()
            barrier.wait(timeout=120)
            row.copy_(acc[n_calls[0]])
            n_calls[0] += 1

        sr.exchange = ops.ScheduleExchange(n_frames, H * W, sr.local_real, torch.device(DEV), 96,
                                           float(rk.get("cone_angle", 0.0)), reduce_fn=reduce_fn)
        ranks.append((sr, n_calls))
    assert len({tuple(sr.local_real) for sr, _ in ranks}) > 1 or (W * H) % (64 * world) == 0     # uneven shares are covered
    outs, errs = [None] * world, []

    def work(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                outs[r] = ranks[r][0].render_local(times)
                torch.cuda.current_stream().synchronize()
        except BaseException as e:
            errs.append(e)
            barrier.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    assert len({n[0] for _, n in ranks}) == 1 and ranks[0][1][0] >= 3            # same number of exchange steps everywhere
    n_img = H * W
    img = torch.zeros((n_frames * n_img + 1, 5), device=DEV)
    for r, (sr, _) in enumerate(ranks):
        rgb, op, dp, _ = outs[r]
        dest = sr.gather_index.view(world, -1)[r, :-1]
        img[dest] = torch.cat([rgb, op, dp], dim=1)
    img = img[:-1].view(n_frames, H, W, 5)
    assert sum(o[3] for o in outs) == sum(a[3] for a in alone) > 3000
    for k in range(n_frames):
        assert torch.equal(img[k, ..., 0:3], alone[k][0]), f"frame {k} rgb"
        assert torch.equal(img[k, ..., 3:4], alone[k][1]) and torch.equal(img[k, ..., 4:5], alone[k][2]), f"frame {k}"


def test_frame_to_uint8_bitexact(oracle):
    """the 8-bit frames of the video step (train_real.py:556-557) -- colours x 255 truncated and
    flipped along the width, depth min-max normalised -- bit-exact against the numpy statement, for ragged sizes,
    negative depths, and without the flip."""
    from ced_nerf_amd import ops
    rng = np.random.default_rng(31)
    for (H, W) in [(1, 1), (7, 13), (72, 96), (600, 801)]:
        rgb = rng.uniform(0, 1, size=(H, W, 3)).astype(np.float32)
        rgb.reshape(-1)[:: 17] = 1.0; rgb.reshape(-1)[1:: 29] = 0.0
        depth = (rng.uniform(-2, 6, size=(H, W)) ** 2).astype(np.float32) - np.float32(1.5)
        if H * W == 1:
            depth[:] = 2.0                                   # constant image: 0/0 -> 0
        for flip in (True, False):
            got = N(ops.frame_to_rgb8(T(rgb), flip)); want = oracle.frame_to_rgb8(rgb, flip)
            assert got.dtype == np.uint8 and got.shape == (H, W, 3)
            assert_bitexact(got, want, f"rgb8 {H}x{W} flip={flip}")
            got_d = N(ops.depth_to_u8(T(depth), flip))
            if H * W == 1:
                assert got_d.item() == 0
                continue
            assert_bitexact(got_d, oracle.depth_to_u8(depth, flip), f"depth8 {H}x{W} flip={flip}")
            assert got_d.min() == 0 and got_d.max() == 255
            assert_bitexact(N(ops.depth_to_u8(T(depth)[..., None].contiguous(), flip)), got_d, "depth [H,W,1]")
    with pytest.raises(NotImplementedError):
        ops.frame_to_rgb8(torch.zeros(4, 4, 3))


def test_reduce_along_rays_matches_scatter_reduce(oracle):
    """a5: cednerf/render.py:8-39 on the HIP kernel, against torch's own scatter_reduce_ on the CPU (the statement the
    reference makes) and a float64 numpy sum: ray-packed and shuffled indices, sum and mean, broadcast and per-channel
    weights, empty rays, empty input, n_rays inferred."""
    from ced_nerf_amd.render import reduce_along_rays
    rng = np.random.default_rng(9)
    n_rays, C = 700, 5
    counts = rng.integers(0, 40, size=n_rays); counts[::7] = 0
    ri = np.repeat(np.arange(n_rays), counts).astype(np.int64)
    S = ri.shape[0]
    vals = rng.normal(size=(S, C)).astype(np.float32)
    w1 = rng.uniform(0, 1, size=(S, 1)).astype(np.float32)
    for shuffle in (False, True):
        order = rng.permutation(S) if shuffle else np.arange(S)
        r_, v_, w_ = ri[order], vals[order], w1[order]
        for reduce in ("sum", "mean"):
            for w in (None, w_, np.repeat(w_, C, axis=1)):
                got = N(reduce_along_rays(T(r_), T(v_), n_rays, None if w is None else T(w), reduce=reduce))
                src = v_ if w is None else w * v_
                ref = torch.zeros((n_rays, C)).scatter_reduce_(0, torch.from_numpy(r_)[:, None].expand(-1, C), torch.from_numpy(src),
                                                               reduce=reduce).numpy()
                acc = np.zeros((n_rays, C)); np.add.at(acc, r_, src.astype(np.float64))
                if reduce == "mean":
                    acc = acc / (counts[:, None] + 1)
                assert got.shape == (n_rays, C)
                assert np.abs(got - acc).max() <= 2e-5 and np.abs(got - ref).max() <= 2e-5, (shuffle, reduce)
    assert reduce_along_rays(T(ri[:0]), T(vals[:0]), 9).shape == (9, C)
    assert N(reduce_along_rays(T(ri[:0]), T(vals[:0]), 9)).sum() == 0
    assert reduce_along_rays(T(ri), T(vals), None, reduce="sum").shape == (int(ri.max()) + 1, C)
    with pytest.raises(AssertionError):
        reduce_along_rays(T(ri), T(vals), n_rays, T(w1[:-1]))


def test_checkpoint_round_trip_renders_the_same_frame(oracle):
    """f3 (train_real.py:433-441,524-529): torch.save({"radiance_field": sd, "occupancy_grid": sd}) -> fresh modules ->
    load_state_dict -> the frame renders bit-identically; the estimator state keys are nerfacc's."""
    import io
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import render_image_test
    sc = _scene("hypernerf", 48, 64, "trained", log2_hashmap_size=15)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    cfg = sc["cfg"]
    ts = T(sc["timestamps"])
    want = render_image_test(1024, f, est, rays, timestamps=ts, **rk)
    buf = io.BytesIO()
    torch.save({"radiance_field": f.state_dict(), "occupancy_grid": est.state_dict()}, buf)
    buf.seek(0)
    ck = torch.load(buf, map_location=DEV)
    assert set(ck["occupancy_grid"]) == {"resolution", "aabbs", "occs", "binaries"}
    f2 = DNGPradianceField(aabb=cfg["aabb"], dst_resolution=cfg["hash_max_res"], log2_hashmap_size=15,
                           moving_step=cfg["moving_step"], seed=11, **cfg["flags"]).to(DEV).eval()
    est2 = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    before = render_image_test(1024, f2, est2, rays, timestamps=ts, **rk)          # empty grid, random weights
    assert before[3] == 0
    f2.load_state_dict(ck["radiance_field"]); est2.load_state_dict(ck["occupancy_grid"])
    got = render_image_test(1024, f2, est2, rays, timestamps=ts, **rk)
    assert got[3] == want[3] and want[3] > 1000
    for a, b in zip(got[:3], want[:3]):
        assert torch.equal(a, b)


def test_reference_checkpoint_renders_the_same_frame(oracle, tmp_path):
    """f3, tiny-cuda-nn half (ced_nerf_amd/checkpoint.py): the synthetic field written as a REFERENCE `model.pth` under
    the documented tcnn layout hypothesis (flat params, 16-padding, a bias in mlp_head's ones-padded input column), loaded
    with load_reference_checkpoint into fresh modules built with the reference's constructor flags: sample count,
    opacity and depth bit-identical (the sigma chain's tensors come back bit for bit), rgb to 1e-5 (the bias folded
    into the constant Y00 input is the same function in real arithmetic)."""
    import subprocess, sys
    from ced_nerf_amd import checkpoint as CK
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import render_image_test
    sc = _scene("dnerf", 64, 48, "trained", log2_hashmap_size=15)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    cfg = sc["cfg"]
    ts = T(sc["timestamps"])
    want = render_image_test(1024, f, est, rays, timestamps=ts, **rk)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = str(tmp_path / "model.pth")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "write_reference_checkpoint.py"), path, "--head-bias"],
                         capture_output=True, text=True, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    f2 = DNGPradianceField(aabb=cfg["aabb"], dst_resolution=cfg["hash_max_res"], log2_hashmap_size=15,
                           moving_step=cfg["moving_step"], seed=11, **cfg["flags"]).to(DEV).eval()
    est2 = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    f2._descriptor()                                             # packed once with the random weights: must be re-packed
    CK.load_reference_checkpoint(path, f2, est2, assume_tcnn_layout=CK.TCNN_LAYOUT)
    got = render_image_test(1024, f2, est2, rays, timestamps=ts, **rk)
    assert got[3] == want[3] and want[3] > 1000
    assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
    assert (got[0] - want[0]).abs().max().item() <= 1e-5


@pytest.mark.parametrize("regime", ["trained", "init"])
@pytest.mark.parametrize("prec", ["f16", "f16x2"])
def test_gui_operating_point(oracle, prec, regime):
    """The viewer's frame (gui.py:203-237): `render_image_test` with max_samples = 200 under fp16 autocast (the reference's
    tcnn networks evaluate in fp16: mode f16; f16x2 is this package's fp32-grade form of it), rays generated on the device
    from the camera pose.  With a random-init field (no ray ends early) a sample budget binds -- the loop ends on
    `max_samples`, not on dead rays (checked at a budget of 64): schedule, totals and every pixel against the oracle's mode of
    the same name, bit for bit."""
    from ced_nerf_amd import cameras, ops
    from ced_nerf_amd.utils import render_image_test
    W, H = 128, 96
    sc = _scene("dnerf", W, H, regime, log2_hashmap_size=17)
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    cfg = sc["cfg"]
    f.set_mlp_precision(prec)
    focal = 0.5 * W / np.tan(0.5 * cfg["camera_angle_x"])
    K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], np.float32)
    c2w = S_look_at(cfg)
    r = cameras.pinhole_rays(K, c2w, W, H, cfg["opengl"], device=DEV)
    o_np, d_np = N(r.origins), N(r.viewdirs)
    ts = torch.tensor([[0.0]], device=DEV)                      # gui.py:200: the viewer starts at t = 0
    ofm = oracle.OracleField(sc["params"], mlp_half=prec)
    totals = {}
    for max_samples in (200, 1024) if regime == "trained" else (200, 64):        # (init: 64 = a budget that certainly binds)
        trace = []
        want = oracle.render_image_test(max_samples, ofm, oest, o_np, d_np, timestamps=N(ts), trace=trace, **sc["render"])
        tracer = ops.FrameTracer(capacity=1100, with_events=False)
        got = render_image_test(max_samples, f, est, r, timestamps=ts, tracer=tracer, **rk)
        assert got[3] == want[3] > 1000
        assert tracer.iterations() == [dict(n_alive=t["n_alive"], n_samples=t["n_samples"], n_new=t["n_new"]) for t in trace]
        for nm, g_, w_ in (("rgb", got[0], want[0]), ("opacity", got[1], want[1]), ("depth", got[2], want[2])):
            assert_bitexact(N(g_), w_, f"gui frame {prec} max_samples={max_samples} {nm}")
        totals[max_samples] = (got[3], sum(t["n_samples"] for t in trace))
    # random-init field: at 200 the loop ends on the budget (cednerf/utils.py:230) with rays still alive
    assert totals[200][1] >= 200 and (regime != "init" or (totals[64][0] < totals[200][0] and totals[64][1] < 200)), totals


def S_look_at(cfg):
    from ced_nerf_amd import synthetic as S
    return S.look_at_c2w(cfg["radius"], 25.0, 75.0, cfg["opengl"])


def test_checkpoint_consistency_gate_flags_a_wrong_layout(tmp_path):
    """tools/verify_checkpoint.py's core (checkpoint.checkpoint_consistency): a `model.pth` whose occupancy grid was
    thresholded from the field's own density (as the reference's trainer does, train_real.py:324-336) loads as "consistent"
    under the documented layout; the same file with the hash levels' blocks rotated, with mlp_base's matrices transposed,
    or with its two halves taken from different models loads without any error -- and is flagged: the numbers fall to
    chance.  This is the check a user with a real reference checkpoint can run; no training data needed."""
    from ced_nerf_amd import checkpoint as CK, synthetic as S
    from ced_nerf_amd.hashgrid import level_tables
    from ced_nerf_amd.model import DNGPradianceField, make_occ_eval_fn
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    cfg = S.CONFIGS["dnerf"]
    p = S.init_field_params(cfg["aabb"], cfg["moving_step"], cfg["hash_max_res"], 15, regime="trained", seed=5)
    lv = level_tables(16, cfg["hash_max_res"], 16, 15)
    tab = p["hash"]["table"].copy()
    tab[int(lv["offset"][5]):] *= 0.02                       # a scene-like density: smooth at the scale of a grid cell
    p["hash"]["table"] = tab
    f = DNGPradianceField.from_params(p, DEV).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    occ_eval_fn = make_occ_eval_fn(f, torch.linspace(0, 1, 5, device=DEV)[:, None], cfg["render_step_size"])
    for step in range(4):
        est._update(step, occ_eval_fn, occ_thre=1e-2)
    frac = float(est.binaries.float().mean())
    assert 0.05 < frac < 0.95, frac

    def load(state, grid):
        f2 = DNGPradianceField(aabb=cfg["aabb"], dst_resolution=cfg["hash_max_res"], log2_hashmap_size=15,
                               moving_step=cfg["moving_step"], seed=11).to(DEV).eval()
        e2 = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
        path = str(tmp_path / "model.pth")
        torch.save({"radiance_field": state, "occupancy_grid": grid}, path)
        CK.load_reference_checkpoint(path, f2, e2, assume_tcnn_layout=CK.TCNN_LAYOUT)
        return CK.checkpoint_consistency(f2, e2, render_step_size=cfg["render_step_size"], n_cells=6000)

    good = CK.reference_state_from_field(f)
    rep = load(good, est.state_dict())
    print("right layout:", {k: rep[k] for k in ("separation_min", "auc_min", "verdict")}, rep["sigma"])
    assert rep["verdict"] == "consistent" and rep["separation_min"] >= 0.5 and rep["auc_min"] >= 0.8, rep
    assert rep["sigma"]["finite"] and rep["rgb"]["finite"] and 0 < rep["rgb"]["std"]
    # (1) the hash table's level blocks rotated by one level (a level-major / level-size misunderstanding)
    bad = dict(good)
    hp = good["hash_encoder.params"].clone()
    bad["hash_encoder.params"] = torch.roll(hp, int(lv["size"][0]) * 2)
    # (2) mlp_base's matrices stored [in][out] instead of [out][in]
    bad2 = dict(good)
    mats, _ = CK.split_tcnn_mlp(good["mlp_base.params"], 32, 16, 1)
    bad2["mlp_base.params"] = CK.join_tcnn_mlp([np.ascontiguousarray(mats[0].T).reshape(mats[0].shape),
                                                 np.ascontiguousarray(mats[1].T).reshape(mats[1].shape)])
    # (3) a grid that belongs to another model
    other = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    other.set_binaries(T(S.make_occupancy(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"])))
    for name, state, grid in (("rotated hash levels", bad, est.state_dict()), ("transposed mlp_base", bad2, est.state_dict()),
                              ("foreign grid", good, other.state_dict())):
        r = load(state, grid)
        print(name + ":", {k: r.get(k) for k in ("separation_min", "auc_min", "verdict")})
        assert r["verdict"] != "consistent" and r["separation_min"] < 0.35 and r["separation_min"] < rep["separation_min"] - 0.3, (name, r)


def test_scatter_pixels_unpermutes_and_converts(oracle):
    """ced_scatter_pixels: rows in marching / gather order -> raster images, padding rows dropped, and the 8-bit colour
    frame of train_real.py:556 in the same pass (bit-exact against numpy)."""
    from ced_nerf_amd import ops
    rng = np.random.default_rng(4)
    H, W, F = 12, 20, 2
    n = F * H * W
    perm = rng.permutation(n)
    rows = rng.uniform(-0.1, 1.1, size=(n + 7, 5)).astype(np.float32)
    dest = np.concatenate([perm, np.full(7, n)]).astype(np.int64)          # 7 padding rows
    order = rng.permutation(n + 7)                                         # padding anywhere in the payload
    rows, dest = rows[order], dest[order]
    G = T(rows)
    rgb, op, dp, rgb8 = ops.scatter_pixels(T(dest), n, G[:, 0:3], G[:, 3:4], G[:, 4:5], want_rgb8_width=W, flip_w=True)
    want = np.zeros((n, 5), np.float32)
    keep = dest < n
    want[dest[keep]] = rows[keep]
    assert_bitexact(N(rgb), want[:, 0:3], "rgb"); assert_bitexact(N(op), want[:, 3:4], "opacity"); assert_bitexact(N(dp), want[:, 4:5], "depth")
    img = want[:, 0:3].reshape(F, H, W, 3)
    want8 = np.flip(np.clip(img * 255, 0, 255), axis=2).astype(np.uint8)
    assert np.array_equal(N(rgb8).reshape(F, H, W, 3), want8)
    # separate source arrays (strides 3, 1, 1): the single-GPU tile-order path
    a, b, c = T(rows[:, 0:3].copy()), T(rows[:, 3:4].copy()), T(rows[:, 4:5].copy())
    rgb2, op2, dp2, none8 = ops.scatter_pixels(T(dest), n, a, b, c)
    assert none8 is None and torch.equal(rgb2, rgb) and torch.equal(op2, op) and torch.equal(dp2, dp)


def test_render_video_equals_frames_rendered_alone(oracle):
    """video.render_video: a camera path with per-frame times streamed through frames in flight, rays generated on
    the device on each lane's stream; every frame equals render_image_test of that frame alone, and the uint8 frames
    are the numpy conversion of those floats.  5 frames over 3 lanes (the last step is ragged), then 1 lane."""
    from ced_nerf_amd import cameras, synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import render_image_test
    from ced_nerf_amd.video import render_video
    W, H = 96, 72
    sc = _scene("dnerf", W, H, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    f = DNGPradianceField.from_params(sc["params"], DEV).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    rk = dict(sc["render"]); rk["render_bkgd"] = T(rk["render_bkgd"])
    focal = 0.5 * W / np.tan(0.5 * cfg["camera_angle_x"])
    K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], np.float32)
    n_frames = 5
    poses = [S.look_at_c2w(cfg["radius"], 30.0, 15.0 + 20.0 * k, cfg["opengl"]) for k in range(n_frames)]
    times = [torch.tensor([[k / (n_frames - 1.0)]], device=DEV) for k in range(n_frames)]
    rays_of = lambda i: cameras.pinhole_rays(K, poses[i], W, H, cfg["opengl"], device=DEV)
    alone = [render_image_test(1024, f, est, rays_of(i), timestamps=times[i], **rk) for i in range(n_frames)]
    assert not torch.equal(alone[0][0], alone[1][0])
    for in_flight in (3, 1):
        frames = render_video(f, est, rays_of, lambda i: times[i], n_frames, render_kwargs=rk, frames_in_flight=in_flight,
                              keep_float=True)
        torch.cuda.synchronize()
        assert len(frames) == n_frames
        for fr, want in zip(frames, alone):
            assert fr["n_samples"] == want[3] and want[3] > 1000
            assert torch.equal(fr["rgb_f32"], want[0]) and torch.equal(fr["depth_f32"], want[2])
            assert_bitexact(N(fr["rgb"]), oracle.frame_to_rgb8(N(want[0])), "video rgb8")
            assert_bitexact(N(fr["depth"]), oracle.depth_to_u8(N(want[2])[..., 0]), "video depth8")
    # two frames per native call (5 frames -> 3 calls, the last one padded), 2 calls in flight
    frames = render_video(f, est, rays_of, lambda i: times[i], n_frames, render_kwargs=rk, frames_in_flight=2,
                          frames_per_call=2, keep_float=True)
    torch.cuda.synchronize()
    assert len(frames) == n_frames and frames[0]["n_samples"] == alone[0][3] + alone[1][3] and frames[1]["n_samples"] is None
    for fr, want in zip(frames, alone):
        assert torch.equal(fr["rgb_f32"], want[0]) and torch.equal(fr["opacity_f32"], want[1]) and torch.equal(fr["depth_f32"], want[2])
    host = render_video(f, est, rays_of, lambda i: times[i], 2, render_kwargs=rk, to_host=True)
    assert isinstance(host[0]["rgb"], np.ndarray) and host[1]["rgb"].shape == (H, W, 3)
    # a failure on a lane's thread surfaces as an exception instead of a hang
    def bad_rays(i):
        if i == 3:
            raise ValueError("no such pose")
        return rays_of(i)
    with pytest.raises(ValueError):
        render_video(f, est, bad_rays, lambda i: times[i], n_frames, render_kwargs=rk)


def test_occupancy_grid_update_parity(oracle):
    """OccGridEstimator._update (positions -> density*step -> EMA max -> threshold) with
    the random draws injected; occs bit-exact against the oracle, binaries equal away from the threshold."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    res, levels = 48, 2
    roi = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0]
    p = S.init_field_params(S.enlarge_aabb(roi, 2), 1.0 / 4096, 1024, 15, regime="trained", use_time_embedding=True)
    of = oracle.OracleField(p)
    f = DNGPradianceField.from_params(p, DEV).eval()
    est = OccGridEstimator(roi, res, levels).to(DEV)
    rng = np.random.default_rng(0)
    cells = res ** 3
    step = 5e-3
    occs_o = np.zeros(levels * cells, np.float32)
    aabbs = oracle.make_aabbs(roi, levels)
    for it in range(2):
        idx = [np.arange(cells, dtype=np.int64), rng.permutation(cells)[: cells // 3].astype(np.int64)]
        noise = [rng.uniform(0, 1, size=(len(i), 3)).astype(np.float32) for i in idx]
        tt = [rng.uniform(0, 1, size=len(i)).astype(np.float32) for i in idx]
        cursor = {"k": 0}

        def occ_o(x):
            k = cursor["k"]; cursor["k"] += 1
            return of.forward(x, tt[k])["density"] * np.float32(step)
        occs_o, bin_o = oracle.occ_grid_update(occs_o, aabbs, res, idx, noise, occ_o, occ_thre=0.01, ema_decay=0.95)
        gcur = {"k": 0}

        def occ_g(x):
            k = gcur["k"]; gcur["k"] += 1
            return f.query_density(x, T(tt[k])[:, None])["density"] * step
        est._update(step=it, occ_eval_fn=occ_g, occ_thre=0.01, ema_decay=0.95, _lvl_indices=[T(i) for i in idx],
                    _noise=[T(n_) for n_ in noise])
        assert_bitexact(N(est.occs), occs_o, f"occs after update {it}")
        thre = min(float(occs_o[occs_o >= 0].astype(np.float64).mean()), 0.01)
        away = np.abs(occs_o - thre) > 1e-6
        assert np.array_equal(N(est.binaries).reshape(-1)[away], bin_o[away])
        assert 0.0 < bin_o.mean() < 1.0 and est.binaries.shape == (levels, res, res, res)
    # the sampled (post-warm-up) path and the training guard
    est.train()
    from ced_nerf_amd.model import make_occ_eval_fn
    est.update_every_n_steps(step=512, occ_eval_fn=make_occ_eval_fn(f, torch.linspace(0, 1, 7, device=DEV)[:, None], step))
    assert torch.isfinite(est.occs).all() and est.occs.min().item() >= 0.0
    est.eval()
    with pytest.raises(RuntimeError):
        est.update_every_n_steps(step=0, occ_eval_fn=None)


def test_device_ray_generation(oracle):
    """rays generated on the device from camera parameters -- the HyperNeRF camera
    against the reference's own outputs (golden), the pinhole camera against the oracle."""
    import os
    from ced_nerf_amd import cameras, synthetic as S
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hypercam_rays.npz"))
    for tag in ("plain", "distorted"):
        kw = dict(orientation=g[tag + "_orientation"], position=g[tag + "_position"],
                  focal_length=float(g[tag + "_focal_length"]), principal_point=g[tag + "_principal_point"],
                  image_size=g[tag + "_image_size"], skew=float(g[tag + "_skew"]),
                  pixel_aspect_ratio=float(g[tag + "_pixel_aspect_ratio"]),
                  radial_distortion=g[tag + "_radial_distortion"], tangential_distortion=g[tag + "_tangential_distortion"])
        rays = cameras.hypercam_rays(device=DEV, **kw)
        assert rays.viewdirs.shape == (36, 48, 3)
        assert np.abs(N(rays.viewdirs) - g[tag + "_rays"]).max() <= 3e-7            # vs the reference itself
        wo, wd = oracle.hypercam_rays(**kw)
        assert np.abs(N(rays.viewdirs) - wd).max() <= 1.2e-7 and np.array_equal(N(rays.origins), wo)
    for opengl in (True, False):
        c2w = S.look_at_c2w(4.0, 25.0, 110.0, opengl)
        W, H = 800, 800
        focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
        K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], np.float32)
        rays = cameras.pinhole_rays(K, c2w, W, H, opengl, device=DEV)
        wo, wd = oracle.pinhole_rays(K, c2w, W, H, opengl)
        assert_bitexact(N(rays.origins), wo, "pinhole origins")
        assert_bitexact(N(rays.viewdirs), wd, "pinhole viewdirs")


def test_frame_renderer_edge_cases(oracle):
    """Empty / ragged / degenerate inputs of render_image_test against the oracle."""
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.utils import Rays, render_image_test
    sc = _scene("dnerf", 37, 23, "trained", log2_hashmap_size=14)          # 851 rays: not a multiple of 64 / 256
    of, oest, f, est, rays, rk = _setup(oracle, sc)
    ts = T(sc["timestamps"])

    def both(max_samples, o, d, est_g, est_o, **over):
        kw = dict(sc["render"]); kw.update(over)
        kg = dict(rk); kg.update({k: (T(v) if isinstance(v, np.ndarray) else v) for k, v in over.items()})
        w = oracle.render_image_test(max_samples, of, est_o, o, d, timestamps=sc["timestamps"], **kw)
        g = render_image_test(max_samples, f, est_g, Rays(T(o), T(d)), timestamps=ts, **kg)
        assert g[3] == w[3]
        for i, nm in enumerate(("rgb", "opacity", "depth")):
            assert g[i].shape == w[i].shape
            assert_bitexact(N(g[i]), w[i], f"{nm} (max_samples={max_samples})")
        return w
    o, d = sc["origins"], sc["viewdirs"]
    for ms in (1, 2, 7, 64):                    # sample budget binds: rays stop mid-object
        both(ms, o, d, est, oest)
    w = both(1024, o.reshape(-1, 3)[:5], d.reshape(-1, 3)[:5], est, oest)       # flat [n,3] rays, n < one wave
    assert w[0].shape == (5, 3)
    # empty occupancy grid: nothing to march, pixels = background
    empty_g = OccGridEstimator(sc["cfg"]["aabb"], 128, 1).to(DEV)
    empty_o = oracle.OracleEstimator(sc["cfg"]["aabb"], 128, 1, np.zeros_like(sc["binaries"]))
    w = both(1024, o, d, empty_g, empty_o)
    assert w[3] == 0 and np.all(w[0] == 1.0)
    # every ray misses the box; fully occupied grid; black background given as None
    away = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (64, 1))
    far_o = np.tile(np.array([[5.0, 5.0, 5.0]], np.float32), (64, 1))
    w = both(1024, far_o, away, est, oest)
    assert w[3] == 0
    full_g = OccGridEstimator(sc["cfg"]["aabb"], 128, 1).to(DEV); full_g.set_binaries(torch.ones_like(full_g.binaries))
    full_o = oracle.OracleEstimator(sc["cfg"]["aabb"], 128, 1, np.ones_like(sc["binaries"]))
    w = both(96, o[::3, ::3], d[::3, ::3], full_g, full_o)
    assert w[3] > 1000
    g = render_image_test(64, f, est, Rays(T(o), T(d)), timestamps=ts, render_step_size=5e-3, render_bkgd=None)
    wk = dict(sc["render"]); wk["render_bkgd"] = None
    w = oracle.render_image_test(64, of, oest, o, d, timestamps=sc["timestamps"], **wk)
    assert_bitexact(N(g[0]), w[0], "rgb without background")
    with pytest.raises(NotImplementedError):
        render_image_test(8, f, est, Rays(T(o), T(d)), timestamps=None)
    with pytest.raises(NotImplementedError, match="cuda"):
        render_image_test(8, f, est, Rays(torch.from_numpy(o), torch.from_numpy(d)), timestamps=ts)


def _two_rank_bench(extra):
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CED_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--width", "256", "--height", "256", "--no-cpu-baseline", "--min-seconds", "0.2", "--frames-per-call", "3"] + extra
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _assert_gather_is_the_single_rank_render(g, frames):
    """bench.py's gather_check: the frames two ranks rendered in shares (image-global schedule, survivor counts
    all-reduced per iteration), all-gathered and un-permuted, against the same frames rendered whole by one rank --
    the same bits and the same sample totals, not a tolerance (north-star: counts bit-exact, pixels <= 1e-4)."""
    assert g["frames"] == frames and g["ok"] and g["bitexact"] and g["samples_equal"], g
    assert g["pixels_over_1e-4"] == 0 and g["rgb_max_abs"] == 0.0 and g["depth_max_abs"] == 0.0 and g["opacity_max_abs"] == 0.0, g
    assert g["samples_gathered"] == g["samples_single_rank"] > 10000, g


def test_bench_two_rank_rehearsal():
    """bench.py's multi-rank path end to end (launch line of the driver, 2 ranks): sharding, the per-iteration schedule
    exchange, frames in flight, the asynchronous pixel gather, the other precision modes, rank 0's JSON line with BOTH
    scalings.  The ranks share this box's one card, so the collectives run over gloo (CED_BENCH_BACKEND=gloo); with
    RCCL only the backend differs."""
    d = _two_rank_bench(["--also", "f32+h16x2", "--scaling", "strong"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["mlp_precision"] == "f16x2"
    assert d["config"]["frames_per_step"] == 9 and d["config"]["frames_per_call"] == 3 and "f32+h16x2" in d["other_mlp_precisions"]
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert set(d["roofline"]["hash_lookup_hbm_frac"]) == {"f32_table_1024B", "f16_table_512B"}
    assert "one communicator, one issuing thread" in d["comm"]["design"] and d["comm"]["timeout_s"] > 0
    o = d["other_scaling"]
    assert o["scaling"] == "weak" and o["frames_per_step"] == 18 and o["value"] > 0
    assert d["comm"]["ranks"] == 2 and d["comm"]["schedule_allreduces_last_call"] >= 3
    _assert_gather_is_the_single_rank_render(d["gather_check"], 3)
    assert d["single_frame_latency_ms"] > 0
    w = d["windows"]
    assert w["n"] >= 5 and w["p10"] <= w["median"] <= w["p90"] and w["steps_each"] == 2


def test_bench_two_rank_rehearsal_weak_scaling():
    """The same launch line with --scaling weak: per-GPU work fixed (a call holds frames_per_call x ranks frames, a
    unit = one rank's share of ONE frame), the strong figure beside it."""
    d = _two_rank_bench(["--also", "", "--no-single-frame"])             # weak is the default
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["frames_per_step"] == 18
    assert d["config"]["frames_per_call"] == 6 and d["other_scaling"]["scaling"] == "strong"
    _assert_gather_is_the_single_rank_render(d["gather_check"], 3)


@pytest.mark.parametrize("case", [0, 3])
def test_trainable_field_matches_fused_kernel(oracle, case):
    """The training graph (torch GEMMs + HIP hash/compositing, ced_nerf_amd/train.py) and the fused inference kernel
    compute the same field from the same parameters."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.train import TrainableField
    kw = dict(FIELD_CASES[case])
    p = S.init_field_params([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 1e-3, 1024, 17, regime="init", seed=3 + case, **kw)
    p["hash"]["table"] = (p["hash"]["table"] * 3000.0).astype(np.float32)        # features of order 0.3
    tf = TrainableField(p, DEV)
    f = tf.to_inference(DEV)
    rng = np.random.default_rng(5)
    n = 4000
    pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)
    t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    with torch.no_grad():
        rgb_t, sig_t = tf(T(pos), T(t), T(d))
    rgb_f, res = f(T(pos), T(t), T(d))
    sig_f = res["density"][:, 0]
    assert torch.equal(sig_t == 0, sig_f == 0)
    assert (rgb_t - rgb_f).abs().max().item() <= 2e-5
    nz = sig_f != 0
    assert ((sig_t[nz] - sig_f[nz]).abs() / sig_f[nz]).max().item() <= 2e-4


WGRAD_SHAPES = [(64, 64), (64, 32), (6, 64), (16, 64), (64, 41), (64, 19), (3, 64), (1, 1), (33, 17), (48, 64), (64, 48)]


@pytest.mark.parametrize("n", [0, 1, 3, 5, 1000, 70001])
def test_weight_grad_matches_oracle(oracle, n):
    """ced_weight_grad (dW = dy^T x, fp32 MFMA accumulation over the sample stream) against the oracle's float64 sum,
    for every layer shape of the model and some ragged ones; sample counts that are not a multiple of the MFMA step
    (4), fewer samples than one step, and none at all."""
    from ced_nerf_amd import ops
    rng = np.random.default_rng(100 + n)
    for n_out, n_in in WGRAD_SHAPES:
        x = rng.normal(size=(n, n_in)).astype(np.float32)
        dy = (rng.normal(size=(n, n_out)) * rng.uniform(0.1, 3.0, size=(1, n_out))).astype(np.float32)
        got = N(ops.weight_grad(T(x), T(dy)))
        want = oracle.weight_grad(x, dy)
        assert got.shape == (n_out, n_in)
        bound = 4e-6 * (np.abs(dy).astype(np.float64).T @ np.abs(x).astype(np.float64)) + 1e-30
        assert (np.abs(got - want) <= bound).all(), (n_out, n_in, np.abs(got - want).max())
        if n == 0:
            assert not got.any()


def test_weight_grad_full_size_reproducible_and_rejects_bad_input(oracle):
    """1.2 M samples x (64 x 64) -- the per-step size of a 262 k-ray training batch: two launches give the same bits
    (no float atomics), the result matches a float64 library product, and bad widths / host tensors are refused."""
    from ced_nerf_amd import ops
    g = torch.Generator(device=DEV).manual_seed(7)
    n = 1_200_003
    x = torch.randn(n, 64, device=DEV, generator=g); dy = torch.randn(n, 64, device=DEV, generator=g)
    a = ops.weight_grad(x, dy); b = ops.weight_grad(x, dy)
    assert torch.equal(a, b)
    want = dy.double().t() @ x.double()
    assert (a.double() - want).abs().max().item() <= 2e-6 * float(n) ** 0.5 * 16
    with pytest.raises(RuntimeError):
        ops.weight_grad(torch.zeros(8, 65, device=DEV), torch.zeros(8, 4, device=DEV))
    with pytest.raises(NotImplementedError):
        ops.weight_grad(torch.zeros(8, 4), torch.zeros(8, 4))
    odd = torch.randn(9, 19, device=DEV, generator=g)                           # a row-offset view: 76-byte offset
    got = ops.weight_grad(odd[1:], torch.ones(8, 3, device=DEV))
    assert torch.allclose(got, odd[1:].sum(0, keepdim=True).expand(3, 19), atol=1e-5)


LINEAR_SHAPES = [(64, 64), (32, 64), (64, 6), (64, 3), (41, 64), (19, 64), (64, 16), (64, 32), (64, 1), (1, 1), (17, 33), (48, 64)]


@pytest.mark.parametrize("n", [0, 1, 31, 33, 5000, 70001])
def test_linear_kernel_matches_float64(oracle, n):
    """ced_linear (csrc/linear.hip), the hand-written layer of the training path: forward with and without ReLU, the
    input gradient (transposed weights) with the fused ReLU mask, every layer shape of the model and ragged ones,
    against a float64 product."""
    from ced_nerf_amd import ops
    rng = np.random.default_rng(n + 3)
    for n_in, n_out in LINEAR_SHAPES:
        x = rng.normal(size=(n, n_in)).astype(np.float32)
        w = rng.normal(size=(n_out, n_in)).astype(np.float32)
        want = x.astype(np.float64) @ w.astype(np.float64).T
        scale = max(1.0, np.abs(want).max()) if n else 1.0
        for relu in (False, True):
            got = N(ops.linear(T(x), T(w), relu=relu))
            ref = np.maximum(want, 0) if relu else want
            assert got.shape == (n, n_out)
            if n:
                assert np.abs(got - ref).max() <= 2e-6 * scale * max(n_in, 8), (n_in, n_out, relu)
        # input gradient: dz [n, n_out] times W [n_out, n_in], masked by the layer input's sign
        dz = rng.normal(size=(n, n_out)).astype(np.float32)
        mask = rng.normal(size=(n, n_in)).astype(np.float32)
        got = N(ops.linear(T(dz), T(w), transpose_w=True, mask=T(mask)))
        ref = (dz.astype(np.float64) @ w.astype(np.float64)) * (mask > 0)
        if n:
            assert np.abs(got - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) * max(n_out, 8), (n_in, n_out, "dx")
    with pytest.raises(RuntimeError, match="widths must be 1..64"):
        ops.linear(torch.zeros(4, 65, device=DEV), torch.zeros(3, 65, device=DEV))


def test_mlp_autograd_node_matches_torch(oracle):
    """`_MlpFn` (ced_linear + ced_weight_grad, one autograd node per MLP) against the same MLP in float64 torch."""
    from ced_nerf_amd.train import _MlpFn
    g = torch.Generator(device=DEV).manual_seed(2)
    for dims in ((32, 64, 64, 64, 6), (41, 64, 16), (19, 64, 64, 3), (32, 64, 1)):
        n = 4097
        x = torch.randn(n, dims[0], device=DEV, generator=g, requires_grad=True)
        ws = [torch.randn(dims[i + 1], dims[i], device=DEV, generator=g).mul_(0.3).requires_grad_() for i in range(len(dims) - 1)]
        y = _MlpFn.apply(x, *ws)
        up = torch.randn(y.shape, device=DEV, generator=g)
        (y * up).sum().backward()
        xd = x.detach().double().requires_grad_()
        wd = [w.detach().double().requires_grad_() for w in ws]
        h = xd
        for i, w in enumerate(wd):
            h = h @ w.t()
            if i < len(wd) - 1:
                h = torch.relu(h)
        (h * up.double()).sum().backward()
        assert (y.detach().double() - h.detach()).abs().max().item() <= 1e-4 * max(1.0, h.abs().max().item())
        assert (x.grad.double() - xd.grad).abs().max().item() <= 1e-4 * max(1.0, xd.grad.abs().max().item()), dims
        for a, b in zip(ws, wd):
            assert (a.grad.double() - b.grad).abs().max().item() <= 2e-4 * max(1.0, b.grad.abs().max().item()), dims


def test_rendering_train_extras_and_their_gradients(oracle):
    """The training extras of cednerf/render.py:101-124 (latent / weight prediction losses reduced per ray) and the
    differentiable (weights, trans) they use, against the same graph in plain float64 torch."""
    from ced_nerf_amd.render import rendering_train
    n_rays, seed = 300, 12
    packed, t0, t1, sig, rgbs = _packed_problem(n_rays, seed)
    ri = np.repeat(np.arange(n_rays), packed[:, 1]).astype(np.int64)
    S = ri.shape[0]
    rng = np.random.default_rng(seed)
    lat = rng.uniform(0, 1, size=(S, 32)).astype(np.float32)
    pw = rng.uniform(0, 1, size=(S, 1)).astype(np.float32)
    sel = rng.uniform(size=S) < 0.9

    def graph(dt, dev):
        cv = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        s_ = cv(sig).to(dt).requires_grad_(); c_ = cv(rgbs).to(dt).requires_grad_()
        l_ = cv(lat).to(dt).requires_grad_(); p_ = cv(pw).to(dt).requires_grad_()
        return s_, c_, l_, p_, cv

    s_, c_, l_, p_, cv = graph(torch.float32, DEV)

    def fn(ts, te, rix):
        return c_, {"density": s_[:, None], "interal_output": {"selector": cv(sel), "latent_losses": l_, "weight_losses": p_}}
    colors, op, dp, ex = rendering_train(cv(t0), cv(t1), cv(ri), n_rays, fn, render_bkgd=torch.ones(3, device=DEV))
    loss = colors.sum() + ex["latent_losses"].mean() + ex["weight_losses"].mean() * 3.0
    loss.backward()
    # float64 reference
    sd, cd, ld, pd_, cvd = graph(torch.float64, "cpu")
    t0d, t1d, rid = cvd(t0).double(), cvd(t1).double(), cvd(ri)
    sdl = sd * (t1d - t0d)
    alpha = 1 - torch.exp(-sdl)
    cs = torch.cumsum(sdl, 0)
    starts = cvd(packed[:, 0]); first = cs[starts.clamp(max=S - 1)] - sdl[starts.clamp(max=S - 1)]
    trans = torch.exp(-(cs - sdl - first[rid]))
    w = trans * alpha
    col = torch.zeros(n_rays, 3, dtype=torch.float64).index_add_(0, rid, w[:, None] * cd)
    opd = torch.zeros(n_rays, 1, dtype=torch.float64).index_add_(0, rid, w[:, None])
    col = col + 1.0 * (1 - opd)
    latent = torch.zeros(n_rays, 32, dtype=torch.float64).index_add_(0, rid, w[:, None].detach() * ld)
    wl = torch.nn.functional.huber_loss(pd_, trans[:, None], reduction="none") * cvd(sel)[:, None]
    cnt = torch.bincount(rid, minlength=n_rays).double() + 1
    wloss = torch.zeros(n_rays, 1, dtype=torch.float64).index_add_(0, rid, w[:, None] * wl) / cnt[:, None]
    ref = col.sum() + latent.mean() + wloss.mean() * 3.0
    ref.backward()
    assert abs(loss.item() - ref.item()) <= 1e-4 * abs(ref.item())
    assert (N(ex["latent_losses"]) - latent.detach().numpy()).__abs__().max() <= 1e-4
    assert (N(ex["weight_losses"]) - wloss.detach().numpy()).__abs__().max() <= 1e-5
    for nm, a, b in (("sigma", s_, sd), ("rgb", c_, cd), ("latent", l_, ld), ("p_weight", p_, pd_)):
        err = (a.grad.cpu().double() - b.grad).abs().max().item()
        assert err <= 2e-4 * max(1e-3, b.grad.abs().max().item()), (nm, err, b.grad.abs().max().item())


def test_trainable_field_gradients_hip_vs_library(oracle):
    """The parameter gradients of a training loss on the all-HIP MLPs (`_MlpFn`: ced_linear + ced_weight_grad) equal
    the all-library ones (torch GEMMs), prediction heads included."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.train import TrainableField
    p = S.init_field_params([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 1e-3, 1024, 17, regime="init", seed=11, **dict(FIELD_CASES[0]))
    p["hash"]["table"] = (p["hash"]["table"] * 3000.0).astype(np.float32)
    tf = TrainableField(p, DEV, use_feat_predict=True, use_weight_predict=True)
    rng = np.random.default_rng(6)
    n = 30000
    pos = T(rng.uniform(-1.4, 1.4, size=(n, 3)).astype(np.float32)); t = T(rng.uniform(0, 1, size=(n, 1)).astype(np.float32))
    d = T(rng.normal(size=(n, 3)).astype(np.float32)); wr = T(rng.normal(size=(n, 3)).astype(np.float32))
    grads = {}
    for mode in (True, False):
        tf.hip_mlp = mode
        tf.hip_weight_grad = mode
        tf.zero_grad(set_to_none=True)
        rgb, res = tf(pos, t, d, return_internal=True)
        io = res["interal_output"]
        assert io["latent_losses"].shape == (n, 32) and io["weight_losses"].shape == (n, 1) and io["move"].shape == (n, 3)
        ((rgb * wr).sum() + res["density"].sum() * 0.1 + io["latent_losses"].sum() * 1e3 + io["weight_losses"].sum()).backward()
        grads[mode] = [q.grad.clone() for q in tf.parameters()]
    assert TrainableField.hip_mlp is True and TrainableField.hip_weight_grad is True
    assert any("mlp_feat_prediction" in nm for nm, _ in tf.named_parameters())
    for (name, _), a, b in zip(tf.named_parameters(), grads[True], grads[False]):
        assert a.abs().max().item() > 0, name
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item(), (name, (a - b).abs().max().item(), b.abs().max().item())


@pytest.mark.parametrize("case", range(4))          # the fp32, non-temporal tables (what training supports)
def test_trainable_field_fused_elementwise_pieces_match_the_torch_statements(oracle, case):
    """The HIP pieces between the MLPs (ced_train_inputs / _warp / _head_in, one launch per direction) against the torch
    statements they replaced (`fused_glue = False`): outputs, internal outputs and every parameter gradient, for all
    flag combinations, through both entry points (explicit positions; ray-packed samples), points outside the box
    included (selector 0, clamp's gradient rule)."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.train import TrainableField
    kw = dict(FIELD_CASES[case])
    p = S.init_field_params([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 1e-3, 1024, 17, regime="init", seed=21 + case, **kw)
    p["hash"]["table"] = (p["hash"]["table"] * 3000.0).astype(np.float32)
    tf = TrainableField(p, DEV, use_feat_predict=True, use_weight_predict=True)
    rng = np.random.default_rng(60 + case)
    n_rays, n = 512, 20000
    ro = T(rng.uniform(-0.4, 0.4, size=(n_rays, 3)).astype(np.float32))
    rd = rng.normal(size=(n_rays, 3)); rd = T((rd / np.linalg.norm(rd, axis=1, keepdims=True)).astype(np.float32))
    ri = T(np.sort(rng.integers(0, n_rays, size=n)).astype(np.int64))
    t0 = T(rng.uniform(0.0, 1.9, size=n).astype(np.float32)); t1 = t0 + 5e-3
    ts = T(rng.uniform(0, 1, size=(n_rays, 1)).astype(np.float32))
    wr = T(rng.normal(size=(n, 3)).astype(np.float32))
    out = {}
    for fused in (True, False):
        tf.fused_glue = fused
        for entry in ("rays", "explicit"):
            tf.zero_grad(set_to_none=True)
            if entry == "rays":
                rgb, res = tf.forward_rays(ro, rd, ri, t0, t1, ts, return_internal=True)
            else:
                pos = ro[ri] + rd[ri] * ((t0 + t1)[:, None] / 2.0)
                rgb, res = tf(pos, ts[ri], rd[ri], return_internal=True)
            io = res["interal_output"]
            loss = (rgb * wr).sum() + res["density"].sum() * 0.1 + io["latent_losses"].sum() * 1e3 + io["weight_losses"].sum() \
                + (io["move"] * wr).sum() * 1e2
            loss.backward()
            out[(fused, entry)] = (rgb.detach(), res["density"].detach(), io["move"].detach(), io["selector"],
                                   [q.grad.clone() for q in tf.parameters()])
    tf.fused_glue = True
    ref = out[(False, "explicit")]
    assert 0 < int(ref[3].sum()) < n                  # some samples inside the box, some outside
    for key in ((True, "rays"), (True, "explicit"), (False, "rays")):
        got = out[key]
        assert torch.equal(got[3], ref[3]), key
        assert (got[0] - ref[0]).abs().max().item() <= 2e-6, key
        assert ((got[1] - ref[1]).abs() <= 2e-5 * ref[1].abs() + 1e-12).all(), key
        assert (got[2] - ref[2]).abs().max().item() <= 1e-9 + 2e-6 * ref[2].abs().max().item(), key
        for (name, _), a, b in zip(tf.named_parameters(), got[4], ref[4]):
            assert b.abs().max().item() > 0, name
            # the Frequency terms differ in the last bit (torch.sin against the inference kernel's exact-reduction
            # sin(pi y)), and the position gradient of the fine hash levels is piecewise constant: a sample that moves
            # by 1e-8 across a cell face changes its contribution -- bounds in the norm, and a looser one on the maximum
            assert (a - b).norm().item() <= 1e-3 * b.norm().item(), (key, name, (a - b).norm().item(), b.norm().item())
            assert (a - b).abs().max().item() <= 5e-3 * b.abs().max().item(), (key, name, (a - b).abs().max().item(), b.abs().max().item())


DW_NETS = [[32, 64, 64, 64, 3], [32, 64, 64, 64, 6], [32, 64, 16], [41, 64, 16], [19, 64, 64, 3], [32, 64, 32], [32, 64, 1],
           [7, 64, 5], [48, 64, 64, 32]]


@pytest.mark.parametrize("n", [1, 31, 33, 1000, 70001])
def test_mlp_backward_with_weight_gradients_in_one_launch(oracle, n):
    """ced_mlp_backward_dw against the two-step path it replaces: the input gradient equals ced_mlp_chain's bit for bit
    (the same walk), every dW_l equals the oracle's float64 sum of dz_l^T a_l within the fp32 accumulation bound of
    ced_weight_grad's test; every network shape of the model and ragged ones, sample counts that are not a multiple of
    the 32-sample tile; two launches give the same bits (no float atomics)."""
    from ced_nerf_amd import ops
    rng = np.random.default_rng(300 + n)
    for widths in DW_NETS:
        L = len(widths) - 1
        assert ops.mlp_backward_dw_supported(widths)
        ws = [T((rng.normal(size=(widths[l + 1], widths[l])) / np.sqrt(widths[l])).astype(np.float32)) for l in range(L)]
        x = T(rng.normal(size=(n, widths[0])).astype(np.float32))
        acts = [x] + ops.mlp_chain(x, ws)[:-1]
        dy = T((rng.normal(size=(n, widths[L])) * rng.uniform(0.1, 3.0, size=(1, widths[L]))).astype(np.float32))
        g0, dws = ops.mlp_backward_dw(dy, ws, acts, want_g0=True)
        g0_b, dws_b = ops.mlp_backward_dw(dy, ws, acts, want_g0=True)
        g = ops.mlp_chain(dy, ws, backward=True, masks=[None] + acts[1:], want=[True] * L)
        assert torch.equal(g0, g[0]) and torch.equal(g0, g0_b), widths
        ups = g[1:] + [dy]
        for l in range(L):
            assert dws[l].shape == ws[l].shape and torch.equal(dws[l], dws_b[l])
            a, dz = N(acts[l]), N(ups[l])
            want = oracle.weight_grad(a, dz)
            bound = 4e-6 * (np.abs(dz).astype(np.float64).T @ np.abs(a).astype(np.float64)) + 1e-30
            assert (np.abs(N(dws[l]) - want) <= bound).all(), (widths, l, np.abs(N(dws[l]) - want).max())
        _, dws_n = ops.mlp_backward_dw(dy, ws, acts, want_g0=False)
        assert all(torch.equal(p_, q_) for p_, q_ in zip(dws, dws_n))
    assert not ops.mlp_backward_dw_supported([32, 64, 48, 3]) and not ops.mlp_backward_dw_supported([64, 64, 3])


def test_table_gradient_on_the_side_stream_is_the_same_gradient(oracle):
    """train_step's overlap_table_grad: the hash-table gradient launched on a stream of its own and handed to the
    parameter before the optimiser step equals the one autograd returns on the main stream (up to the order of the
    float atomics), every other gradient is unchanged, and nothing is left deferred afterwards."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd import train as TR
    p = S.init_field_params([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 1e-3, 1024, 17, regime="init", seed=31)
    p["hash"]["table"] = (p["hash"]["table"] * 3000.0).astype(np.float32)
    tf = TR.TrainableField(p, DEV)
    rng = np.random.default_rng(8)
    n = 50000
    pos = T(rng.uniform(-1.4, 1.4, size=(n, 3)).astype(np.float32)); t = T(rng.uniform(0, 1, size=(n, 1)).astype(np.float32))
    d = T(rng.normal(size=(n, 3)).astype(np.float32)); wr = T(rng.normal(size=(n, 3)).astype(np.float32))
    grads = {}
    for deferred in (False, True):
        tf.zero_grad(set_to_none=True)
        rgb, sigma = tf(pos, t, d)
        loss = (rgb * wr).sum() + sigma.sum() * 0.1
        if deferred:
            TR.begin_deferred_table_grad(DEV)
        loss.backward()
        if deferred:
            assert tf.hash_table.grad is None and len(TR._HashFn.deferred["pending"]) == 1
            TR.join_deferred_table_grad(tf.hash_table)
        assert TR._HashFn.deferred is None
        torch.cuda.synchronize()
        grads[deferred] = {nm: q.grad.clone() for nm, q in tf.named_parameters()}
    for nm in grads[False]:
        a, b = grads[True][nm], grads[False][nm]
        assert b.abs().max().item() > 0, nm
        assert (a - b).abs().max().item() <= 1e-5 * b.abs().max().item(), (nm, (a - b).abs().max().item())


@pytest.mark.parametrize("name,alpha_thre", [("dnerf", 0.0), ("dnerf", 0.004), ("hypernerf", 0.0)])
def test_sampling_on_the_native_visibility_pass(oracle, name, alpha_thre):
    """OccGridEstimator.sampling as the training step calls it (stratified near planes, per-ray timestamps): the
    sampling-only mode of ced_render_image (density front to back, rays stopped at the threshold) returns the survivors
    of the filter over every marched sample, bit for bit."""
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.train import TrainableField
    sc = _scene(name, 96, 72, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    fused = TrainableField(sc["params"], DEV).shared_inference()
    fused.train()
    o = T(sc["origins"]).reshape(-1, 3).contiguous(); d = T(sc["viewdirs"]).reshape(-1, 3).contiguous()
    g = torch.Generator(device=DEV).manual_seed(4)
    ts = torch.rand(o.shape[0], 1, device=DEV, generator=g)

    def sigma_fn(t_starts, t_ends, ray_indices):
        return fused.query_rays(o, d, ray_indices, t_starts, t_ends, ts, want_rgb=False)[1]

    kw = dict(sigma_fn=sigma_fn, near_plane=cfg["near_plane"], far_plane=cfg["far_plane"],
              render_step_size=cfg["render_step_size"], stratified=True, cone_angle=cfg["cone_angle"], alpha_thre=alpha_thre)
    torch.manual_seed(9)
    a = est.sampling(o, d, sigma_field=(fused, ts, True), **kw)
    torch.manual_seed(9)
    b = est.sampling(o, d, **kw)
    assert b[0].numel() > 1000
    for x, y, nm in zip(a, b, ("ray_indices", "t_starts", "t_ends")):
        assert x.dtype == y.dtype and torch.equal(x, y), nm
    # the second batch of a size marches in one pass (capacity from the first); one that does not fit is redone exactly
    n = o.shape[0]
    marched = est._march_totals[n]
    torch.manual_seed(9)
    c = est.sampling(o, d, sigma_field=(fused, ts, True), **kw)
    est._march_totals[n] = 1
    torch.manual_seed(9)
    e = est.sampling(o, d, sigma_field=(fused, ts, True), **kw)
    assert est._march_totals[n] == marched and (marched > 65537 + 1 or name != "dnerf")
    for other in (c, e):
        for x, y, nm in zip(other, b, ("ray_indices", "t_starts", "t_ends")):
            assert torch.equal(x, y), nm


@pytest.mark.parametrize("widths,n", [((32, 64, 64, 64, 3), 100003), ((32, 64, 16), 4097), ((19, 64, 64, 3), 33),
                                      ((41, 64, 16), 1), ((7, 5), 1000), ((64, 64, 64, 64, 64, 64, 64), 20011)])
def test_fused_mlp_chain_equals_layerwise(oracle, widths, n):
    """ced_mlp_chain (a whole MLP per launch, activations in registers from layer to layer) against the layer-by-layer
    ced_linear calls it replaces: every activation and every gradient bit for bit; and through autograd (_MlpFn)."""
    from ced_nerf_amd import ops
    from ced_nerf_amd.train import _MlpFn
    g = torch.Generator(device=DEV).manual_seed(len(widths) * 1000 + n)
    x = torch.randn(n, widths[0], device=DEV, generator=g)
    ws = [torch.randn(widths[l + 1], widths[l], device=DEV, generator=g) * (1.5 / widths[l] ** 0.5) for l in range(len(widths) - 1)]
    L = len(ws)
    acts = ops.mlp_chain(x, ws)
    h = x
    for l, w in enumerate(ws):
        h = ops.linear(h, w, relu=l < L - 1)
        assert torch.equal(acts[l], h), f"forward layer {l}"
    dy = torch.randn(n, widths[-1], device=DEV, generator=g)
    masks = [None] + acts[:-1]
    gs = ops.mlp_chain(dy, ws, backward=True, masks=masks)
    dz = dy
    for l in reversed(range(L)):
        dz = ops.linear(dz, ws[l], transpose_w=True, mask=masks[l])
        assert torch.equal(gs[l], dz), f"backward layer {l}"
    some = ops.mlp_chain(dy, ws, backward=True, masks=masks, want=[l == 0 for l in range(L)])
    assert torch.equal(some[0], gs[0]) and all(t is None for t in some[1:])
    # autograd node: fused and layer-wise give the same output and the same parameter / input gradients
    # (fused_dw: the weight gradients inside the backward walk -- another summation order, compared below)
    outs = {}
    for fused, fused_dw in ((True, False), (False, False), (True, True)):
        _MlpFn.fused, _MlpFn.fused_dw = fused, fused_dw
        xi = x.clone().requires_grad_(True)
        wi = [w.clone().requires_grad_(True) for w in ws]
        y = _MlpFn.apply(xi, *wi)
        (y * dy).sum().backward()
        outs[(fused, fused_dw)] = (y.detach(), xi.grad, [w.grad for w in wi])
    _MlpFn.fused, _MlpFn.fused_dw = True, True
    ref = outs[(False, False)]
    assert torch.equal(outs[(True, False)][0], ref[0]) and torch.equal(outs[(True, False)][1], ref[1])
    for a, b in zip(outs[(True, False)][2], ref[2]):
        assert torch.equal(a, b)
    one = outs[(True, True)]
    assert torch.equal(one[0], ref[0]) and torch.equal(one[1], ref[1])          # the same walk: the same input gradient
    for a, b in zip(one[2], ref[2]):
        assert (a - b).abs().max().item() <= 2e-5 * b.abs().max().item() + 1e-30


@pytest.mark.parametrize("kind", ["f16_table", "temporal", "temporal_f16"])
def test_training_on_the_reference_table_types(oracle, kind):
    """train_real.py trains fp16 tiny-cuda-nn parameters under autocast (:330; fp32 master, fp16 evaluated copy) and the
    temporal table has a backward of its own (hash_encoder_inter.py:202-275).  TrainableField on those table types: the
    differentiable forward equals the fused inference kernel on the same parameters (which reads the fp16 / temporal table
    natively), the table gradient of a loss equals the oracle's float64 gradient of the same encode, and a student with a
    damaged table relearns a teacher's renders (the evaluated fp16 copy following its fp32 master step by step)."""
    from ced_nerf_amd import ops, synthetic as S
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.train import TrainableField, train_step
    from ced_nerf_amd.utils import Rays, render_image
    temporal, f16 = kind.startswith("temporal"), kind.endswith("f16") or kind == "f16_table"
    kw = dict(table_dtype=np.float16 if f16 else np.float32)
    if temporal:
        kw.update(temporal_hash=True, use_time_embedding=True)
    cfg = S.CONFIGS["dnerf"]
    params = S.init_field_params(cfg["aabb"], cfg["moving_step"], cfg["hash_max_res"], 14, regime="trained", seed=3, **kw)
    tf = TrainableField(params, DEV)
    assert tf.temporal == temporal and tf.table_f16 == f16 and tf.hash_table.dtype == torch.float32
    assert tf.hash_table.shape[1] == (8 if temporal else 2)
    f = tf.to_inference(DEV)
    assert f.hash_table.dtype == (torch.float16 if f16 else torch.float32)
    rng = np.random.default_rng(5)
    n = 6000
    pos = rng.uniform(-1.6, 1.6, size=(n, 3)).astype(np.float32)
    t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    rgb_t, sig_t = tf(T(pos), T(t), T(d))
    rgb_f, res = f(T(pos), T(t), T(d))
    assert (rgb_t - rgb_f).abs().max().item() <= 5e-5
    sig_f = res["density"][:, 0]
    assert torch.equal(sig_t == 0, sig_f == 0)
    assert ((sig_t - sig_f).abs() <= 2e-4 * sig_f.abs() + 1e-12).all()
    # the table gradient through autograd == the oracle's float64 gradient of the encode for the same upstream gradient
    tf.zero_grad(set_to_none=True)
    xn = torch.rand(3000, 3, device=DEV)
    tt = torch.rand(3000, device=DEV)
    from ced_nerf_amd.train import _HashFn
    up = torch.randn(3000, 32, device=DEV)
    feats = _HashFn.apply(xn, tf.hash_table, tf.hash_cfg, tt, tf._table_eval())
    (feats * up).sum().backward()
    of = oracle.OracleField({"hash": dict(params["hash"])})
    if temporal:
        want = of.hash_encode_backward_temporal(N(xn), N(tt), N(up))
    else:
        want, _ = of.hash_encode_backward(N(xn), N(up))
    got = N(tf.hash_table.grad).astype(np.float64)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-5 * np.abs(want).max()
    # a short training run: teacher = these parameters, student = a damaged table
    sc = _scene("dnerf", 64, 48, "trained", log2_hashmap_size=14)
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    bk = T(sc["render"]["render_bkgd"])
    rk = dict(sc["render"], render_bkgd=bk)
    rays = Rays(T(sc["origins"]), T(sc["viewdirs"]))
    ts = T(sc["timestamps"])
    target = render_image(f, est, rays, timestamps=ts, **rk)[0].reshape(-1, 3)
    sp = dict(params); sp["hash"] = dict(params["hash"])
    tab = params["hash"]["table"].astype(np.float32)
    sp["hash"]["table"] = (tab * 0.5 + rng.normal(size=tab.shape) * 0.05).astype(params["hash"]["table"].dtype)
    student = TrainableField(sp, DEV)
    opt = torch.optim.Adam([student.hash_table], lr=2e-2)
    o = rays.origins.reshape(-1, 3); dd = rays.viewdirs.reshape(-1, 3)
    idx_all = ((target - bk).abs().sum(dim=1) > 1e-3).nonzero().flatten()
    assert idx_all.numel() > 200
    g = torch.Generator(device=DEV).manual_seed(1)
    losses = []
    for step in range(30):
        idx = idx_all[torch.randint(0, idx_all.numel(), (1024,), device=DEV, generator=g)]
        out = train_step(student, est, opt, o[idx].contiguous(), dd[idx].contiguous(), ts, target[idx].contiguous(),
                         cfg["render_step_size"], near_plane=cfg["near_plane"], far_plane=cfg["far_plane"], render_bkgd=bk)
        assert out["n_samples"] > 0 and np.isfinite(out["loss"])
        losses.append(out["loss"])
    print(kind, "losses", [round(x, 5) for x in losses[::5]])
    assert np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), losses
    if f16:       # the evaluated copy is the master rounded to fp16, after every step
        assert torch.equal(student.hash_table_half, student.hash_table.detach().half())


def test_training_steps_reduce_the_loss(oracle):
    """train.train_step end to end: HIP sampling, HIP hash forward/backward, HIP MLPs (ced_linear / ced_weight_grad),
    HIP compositing forward/backward, Adam.  A student whose hash table was damaged relearns a teacher's renders."""
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    from ced_nerf_amd.train import TrainableField, train_step
    from ced_nerf_amd.utils import Rays, render_image
    sc = _scene("dnerf", 96, 72, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    params = sc["params"]
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    est.set_binaries(T(sc["binaries"]))
    rk = dict(sc["render"]); bk = T(rk["render_bkgd"])
    teacher = TrainableField(params, DEV).to_inference(DEV)
    rays = Rays(T(sc["origins"]), T(sc["viewdirs"]))
    ts = T(sc["timestamps"])
    target = render_image(teacher, est, rays, timestamps=ts, **dict(rk, render_bkgd=bk))[0].reshape(-1, 3)
    student_p = dict(params); student_p["hash"] = dict(params["hash"])
    rng = np.random.default_rng(0)
    student_p["hash"]["table"] = (params["hash"]["table"] * 0.5 + rng.normal(size=params["hash"]["table"].shape) * 0.05).astype(np.float32)
    student = TrainableField(student_p, DEV)
    opt = torch.optim.Adam([student.hash_table], lr=2e-2)
    o = rays.origins.reshape(-1, 3); d = rays.viewdirs.reshape(-1, 3)
    hit = (target - bk).abs().sum(dim=1) > 1e-3                         # rays that see the object
    idx_all = hit.nonzero().flatten()
    assert idx_all.numel() > 500
    g = torch.Generator(device=DEV).manual_seed(1)
    losses = []
    for step in range(30):
        idx = idx_all[torch.randint(0, idx_all.numel(), (2048,), device=DEV, generator=g)]
        out = train_step(student, est, opt, o[idx].contiguous(), d[idx].contiguous(), ts, target[idx].contiguous(),
                         cfg["render_step_size"], near_plane=cfg["near_plane"], far_plane=cfg["far_plane"],
                         cone_angle=cfg["cone_angle"], alpha_thre=0.0, render_bkgd=bk)
        assert out["n_samples"] > 0 and np.isfinite(out["loss"])
        losses.append(out["loss"])
    first, last = np.mean(losses[:5]), np.mean(losses[-5:])
    print("training losses", [round(x, 5) for x in losses[::5]])
    assert last < 0.6 * first, (first, last)
    # the other loop pieces of train_real.py:324-420: GradScaler, dynamic ray batch, occupancy refresh on the current density
    from ced_nerf_amd.train import next_num_rays, refresh_occupancy
    scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 10)
    idx = idx_all[:1024]
    out = train_step(student, est, opt, o[idx].contiguous(), d[idx].contiguous(), ts, target[idx].contiguous(),
                     cfg["render_step_size"], near_plane=cfg["near_plane"], far_plane=cfg["far_plane"], render_bkgd=bk,
                     grad_scaler=scaler)
    assert np.isfinite(out["loss"]) and next_num_rays(1024, out["n_samples"], 1 << 16) == int(1024 * (65536 / out["n_samples"]))
    est2 = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(DEV)
    for step in range(0, 64, 16):
        refresh_occupancy(student, est2, step, ts, cfg["render_step_size"])
    frac = est2.binaries.float().mean().item()
    assert 0.0 < frac < 1.0, frac          # the synthetic field is dense almost everywhere; the refresh marks cells
