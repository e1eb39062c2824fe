"""CPU suite, part 1: pin the oracle.

(1) golden vectors captured from the importable reference modules (tests/golden/*.npz);
(2) analytic known-answer tests; (3) the independent PyTorch restatement (oracle/torch_oracle.py);
(4) self-consistency of the two render drivers.
"""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---- (1) golden vectors of the reference -------------------------------------------------------
def test_time_encoders_match_reference_golden(oracle):
    g = np.load(os.path.join(GOLD, "time_encoders.npz"))
    got = oracle.time_encode(g["t"])
    assert got.shape == g["plain"].shape == (g["t"].shape[0], int(g["plain_latent_dim"]))
    assert np.abs(got - g["plain"]).max() <= 1e-6
    got = oracle.time_encode(g["t_damped"], g["move"], with_exp=True)
    assert got.shape == g["damped"].shape == (g["t_damped"].shape[0], int(g["damped_latent_dim"]))
    assert np.abs(got - g["damped"]).max() <= 1e-6
    # SURVEY Appendix B.6 spot values
    assert np.allclose(oracle.time_encode([0.3])[0], [0.3000, 0.2955, 0.5646, 0.9320, 0.6755, 0.9553, 0.8253, 0.3624,
                                                     -0.7374], atol=5e-5)


def test_product_encoder_modules_match_reference_golden():
    from ced_nerf_amd.encoder import SinusoidalEncoder, SinusoidalEncoderWithExp
    g = np.load(os.path.join(GOLD, "time_encoders.npz"))
    plain, damped = SinusoidalEncoder(1, 0, 4, True), SinusoidalEncoderWithExp(1, 0, 4, True)
    assert plain.latent_dim == int(g["plain_latent_dim"]) and damped.latent_dim == int(g["damped_latent_dim"])
    assert np.abs(plain(torch.from_numpy(g["t"])).numpy() - g["plain"]).max() <= 1e-6
    out = damped(torch.from_numpy(g["t_damped"]), torch.from_numpy(g["move"])).numpy()
    assert np.abs(out - g["damped"]).max() <= 1e-6


def test_rays_namedtuple_matches_reference_golden():
    from ced_nerf_amd.utils import Rays, namedtuple_map
    g = np.load(os.path.join(GOLD, "rays_namedtuple.npz"))
    assert list(Rays._fields) == list(g["fields"])
    rays = Rays(origins=torch.arange(24.0).reshape(2, 4, 3), viewdirs=torch.ones(2, 4, 3))
    flat = namedtuple_map(lambda r: r.reshape([8] + list(r.shape[2:])), rays)
    assert isinstance(flat, Rays)
    assert np.array_equal(flat.origins.numpy(), g["flat_origins"]) and np.array_equal(flat.viewdirs.numpy(), g["flat_viewdirs"])


def test_hypercam_rays_match_reference_golden(oracle):
    """datasets/hyper_cam.py Camera.pixels_to_rays (imported by tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "hypercam_rays.npz"))
    for tag in ("plain", "distorted"):
        o, d = oracle.hypercam_rays(g[tag + "_orientation"], g[tag + "_position"], g[tag + "_focal_length"],
                                    g[tag + "_principal_point"], g[tag + "_image_size"], g[tag + "_skew"],
                                    g[tag + "_pixel_aspect_ratio"], g[tag + "_radial_distortion"],
                                    g[tag + "_tangential_distortion"])
        assert d.shape == g[tag + "_rays"].shape == (36, 48, 3)
        assert np.abs(d - g[tag + "_rays"]).max() <= 3e-7
        assert np.allclose(o, g[tag + "_position"])
    assert np.abs(g["plain_rays"] - g["distorted_rays"]).max() > 1e-3      # the distortion does something


def test_pinhole_rays_match_scene_generator(oracle):
    from ced_nerf_amd import synthetic as S
    for opengl in (True, False):
        c2w = S.look_at_c2w(4.0, 30.0, 40.0, opengl)
        W, H, ang = 50, 30, 0.6911112070083618
        focal = 0.5 * W / np.tan(0.5 * ang)
        K = np.array([[focal, 0, W / 2.0], [0, focal, H / 2.0], [0, 0, 1]], np.float32)
        o, d = oracle.pinhole_rays(K, c2w, W, H, opengl)
        so, sd = S.make_camera_rays(W, H, ang, c2w, opengl)
        assert np.abs(o - so).max() == 0 and np.abs(d - sd).max() <= 3e-7


# ---- scalar math kernels vs float64 ------------------------------------------------------------
def test_math_kernels(oracle):
    L = oracle.lib()
    xs = np.concatenate([np.linspace(-87, 88, 4001), [-103.5, -100.0, 0.0, 1e-8, -1e-8]]).astype(np.float32)
    got = np.array([L.ced_o_expf(float(x)) for x in xs], np.float64)
    ref = np.exp(xs.astype(np.float64))
    ok = ref > 1e-37                      # normal range: relative error ~1 ulp
    assert np.max(np.abs(got[ok] - ref[ok]) / ref[ok]) < 3e-7
    assert np.max(np.abs(got[~ok] - ref[~ok])) < 1e-44 * 4
    assert L.ced_o_expf(100.0) == np.inf and L.ced_o_expf(-110.0) == 0.0 and np.isnan(L.ced_o_expf(float("nan")))
    xs = np.linspace(-40, 40, 8001).astype(np.float32)
    got = np.array([L.ced_o_sinf(float(x)) for x in xs])
    assert np.abs(got - np.sin(xs.astype(np.float64))).max() < 1.5e-7
    ys = np.linspace(-13, 13, 8001).astype(np.float32)
    for ph, fn in ((0, np.sin), (1, np.cos)):
        got = np.array([L.ced_o_sinpi_phase(float(y), ph) for y in ys])
        assert np.abs(got - fn(np.pi * ys.astype(np.float64))).max() < 1.5e-7
    assert L.ced_o_sinpi_phase(1.0, 0) == 0.0 and L.ced_o_sinpi_phase(0.5, 0) == 1.0 and L.ced_o_sinpi_phase(2.0, 1) == 1.0


# ---- (2) known-answer tests, SURVEY Appendix B -------------------------------------------------
def test_kat_ray_aabb(oracle):
    o = np.array([[-2, 0.01, 0.01], [-2, 3, 0], [0, 0, 0]], np.float32)
    d = np.array([[1, 0, 0], [1, 0, 0], [0, 0, -1]], np.float32)
    tmin, tmax, hit = oracle.ray_aabb_intersect(o, d, np.array([[-1, -1, -1, 1, 1, 1]], np.float32))
    assert hit[:, 0].tolist() == [True, False, True]
    assert tmin[0, 0] == 1.0 and tmax[0, 0] == 3.0 and np.isinf(tmin[1, 0]) and tmax[2, 0] == 1.0


def test_kat_traverse_2x2x2(oracle):
    b = np.zeros((1, 2, 2, 2), bool); b[0, 1, 0, 0] = True
    r = oracle.traverse_grids(np.array([[-2, -.5, -.5]], np.float32), np.array([[1, 0, 0]], np.float32), b,
                              np.array([[-1, -1, -1, 1, 1, 1]], np.float32), np.zeros(1), np.full(1, 1e10), 0.25, 0.0)
    assert r["t_starts"].tolist() == [2.0, 2.25, 2.5, 2.75] and r["t_ends"].tolist() == [2.25, 2.5, 2.75, 3.0]
    assert r["packed_info"].tolist() == [[0, 4]] and r["termination_planes"].tolist() == [3.0]
    # limit: two samples per call, resuming from the termination plane gives the same sample set
    r1 = oracle.traverse_grids(np.array([[-2, -.5, -.5]], np.float32), np.array([[1, 0, 0]], np.float32), b,
                               np.array([[-1, -1, -1, 1, 1, 1]], np.float32), np.zeros(1), np.full(1, 1e10), 0.25, 0.0, 2, True)
    assert r1["t_starts"].tolist() == [2.0, 2.25] and r1["termination_planes"].tolist() == [2.5]
    r2 = oracle.traverse_grids(np.array([[-2, -.5, -.5]], np.float32), np.array([[1, 0, 0]], np.float32), b,
                               np.array([[-1, -1, -1, 1, 1, 1]], np.float32), r1["termination_planes"], np.full(1, 1e10),
                               0.25, 0.0, 2, True)
    assert r2["t_starts"].tolist() == [2.5, 2.75] and r2["t_ends"].tolist() == [2.75, 3.0]
    # masked ray: no samples
    r3 = oracle.traverse_grids(np.array([[-2, -.5, -.5]], np.float32), np.array([[1, 0, 0]], np.float32), b,
                               np.array([[-1, -1, -1, 1, 1, 1]], np.float32), np.zeros(1), np.full(1, 1e10), 0.25, 0.0, 2, True,
                               np.array([False]))
    assert r3["packed_info"][0, 1] == 0 and r3["t_starts"].shape == (0,)


def test_kat_constant_sigma_weights(oracle):
    K, sigma, delta = 12, np.float32(3.0), np.float32(0.05)
    t0 = (np.arange(K) * delta).astype(np.float32); t1 = (t0 + delta).astype(np.float32)
    packed = np.array([[0, K]], np.int64)
    w, tr, al = oracle.render_weight_from_density(t0, t1, np.full(K, sigma, np.float32), packed)
    k = np.arange(K)
    assert np.allclose(w, np.exp(-sigma * delta * k) * (1 - np.exp(-sigma * delta)), rtol=2e-6)
    acc = oracle.accumulate_along_rays_(w, None, packed, np.zeros((1, 1), np.float32))
    assert np.isclose(acc[0, 0], 1 - np.exp(-sigma * delta * K), rtol=1e-6)
    w2, _, _ = oracle.render_weight_from_density(t0, t1, np.full(K, sigma, np.float32), packed, np.full(K, 0.25, np.float32))
    assert np.allclose(w2, 0.25 * w, rtol=1e-6)


def test_kat_hash_affine_exactness_and_index(oracle):
    from ced_nerf_amd import synthetic as S
    # B.5: spot value of the hashed index
    assert (((1 * 1) ^ (1 * 2654435761 & 0xFFFFFFFF) ^ (1 * 805459861 & 0xFFFFFFFF)) & 0xFFFFFFFF) == 2922720805
    assert 2922720805 % 2 ** 21 == 1388069
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-4, 1024, 21, regime="init")
    lv = oracle.hash_levels(16, 1024, 16, 21)
    assert lv["total"] == 19263424 and list(lv["hashed"]) == [0] * 8 + [1] * 8 and int(lv["res"][15]) == 1024
    table = np.zeros_like(p["hash"]["table"])
    # B.4: dense level filled with an affine function of the vertex -> trilinear interpolation returns it
    l = 3; res = int(lv["res"][l]); off = int(lv["offset"][l]); sc = float(lv["scale"][l])
    zz, yy, xx = np.meshgrid(np.arange(res), np.arange(res), np.arange(res), indexing="ij")
    idx = (xx + yy * res + zz * res * res).reshape(-1)
    table[off + idx, 0] = (0.5 * xx - 0.25 * yy + 2.0 * zz + 1.0).reshape(-1)
    table[off + idx, 1] = (xx + yy + zz).reshape(-1)
    p["hash"]["table"] = table
    f = oracle.OracleField({"hash": p["hash"]})
    x = np.random.default_rng(0).uniform(0.02, 0.98, size=(500, 3)).astype(np.float32)
    out = f.hash_encode(x)
    pos = x.astype(np.float64) * sc + 0.5
    assert np.allclose(out[:, 2 * l], 0.5 * pos[:, 0] - 0.25 * pos[:, 1] + 2.0 * pos[:, 2] + 1.0, atol=2e-4)
    assert np.allclose(out[:, 2 * l + 1], pos.sum(1), atol=2e-4)
    assert np.all(out[:, :2 * l] == 0) and np.all(out[:, 2 * l + 2:] == 0)
    # integer part of the lookup on a hashed level
    ind = f.hash_indices(np.array([[1.0 / 1023, 1.0 / 1023, 1.0 / 1023]], np.float32))
    assert ind[0, 15, 7] == lv["offset"][15] + ((2 * 1) ^ (2 * 2654435761 & 0xFFFFFFFF) ^ (2 * 805459861 & 0xFFFFFFFF)) % 2 ** 21


def test_kat_composite_test(oracle):
    """composite_test (volume_render_test.py) with alpha_threshold=0 equals weights+accumulate with prefix."""
    import ctypes as C
    rng = np.random.default_rng(3)
    K = 9
    t0 = np.sort(rng.uniform(0, 1, K)).astype(np.float32); t1 = (t0 + 0.01).astype(np.float32)
    sig = rng.uniform(0.1, 20, K).astype(np.float32); rgbs = rng.uniform(0, 1, (K, 3)).astype(np.float32)
    packed = np.array([[0, K]], np.int64); alive = np.array([0], np.int64)
    op = np.array([[0.2]], np.float32); dp = np.zeros((1, 1), np.float32); rgb = np.zeros((1, 3), np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    oracle.lib().ced_o_composite_test(C.c_int64(1), p(sig), p(rgbs), p(t0), p(t1), p(packed), p(alive), C.c_float(0.0),
                                      C.c_float(0.0), p(op), p(dp), p(rgb))
    w, _, _ = oracle.render_weight_from_density(t0, t1, sig, packed, np.full(K, 0.8, np.float32))
    assert np.allclose(rgb[0], (w[:, None] * rgbs).sum(0), rtol=1e-5)
    assert np.isclose(op[0, 0], 0.2 + w.sum(), rtol=1e-5)


# ---- (3) independent PyTorch restatement --------------------------------------------------------
@pytest.mark.parametrize("flags", [dict(), dict(use_div_offsets=True, use_time_embedding=True, use_time_attenuation=True),
                                   dict(use_time_embedding=True, table_dtype=np.float16),
                                   dict(temporal_hash=True)])
@pytest.mark.parametrize("regime", ["init", "trained"])
def test_c_oracle_matches_torch_oracle_field(oracle, flags, regime):
    from ced_nerf_amd import synthetic as S
    from oracle.torch_oracle import TorchField
    p = S.init_field_params([-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], 1.0 / 64 if regime == "trained" else 1e-4, 1024, 15,
                            regime=regime, seed=3, **flags)
    rng = np.random.default_rng(5)
    n = 3000
    pos = rng.uniform(-1.55, 1.55, size=(n, 3)).astype(np.float32)
    t = rng.uniform(0, 1, size=(n, 1)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    a = oracle.OracleField(p).forward(pos, t, d, want_geo=True, want_xnorm=True)
    b = TorchField(p).forward(pos, t, d)
    assert np.abs(a["x_norm"] - b["x_norm"]).max() < 1e-5
    inside = np.all((b["x_norm"] > 1e-4) & (b["x_norm"] < 1 - 1e-4), axis=1)
    assert np.abs(a["base_mlp_out"] - b["base_mlp_out"])[inside].max() < 2e-4 * max(1.0, np.abs(b["base_mlp_out"]).max())
    rel = np.abs(a["density"] - b["density"]) / (np.abs(b["density"]) + 1e-3)
    assert rel[inside].max() < 2e-3
    assert np.abs(a["rgb"] - b["rgb"])[inside].max() < 1e-4
    assert (a["density"][~np.all((b["x_norm"] > -1e-4) & (b["x_norm"] < 1 + 1e-4), axis=1)] == 0).all()


def test_c_oracle_matches_torch_oracle_hash_and_compositing(oracle):
    from ced_nerf_amd import synthetic as S
    from oracle import torch_oracle as TO
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-4, 4096, 16, regime="trained")
    x = np.random.default_rng(0).uniform(0, 1, size=(4000, 3)).astype(np.float32)
    a = oracle.OracleField({"hash": p["hash"]}).hash_encode(x)
    b = TO.TorchField(p).hash_encode(torch.from_numpy(x)).numpy()
    assert np.abs(a - b).max() < 2e-6
    rng = np.random.default_rng(2)
    counts = rng.integers(0, 30, size=400); base = np.cumsum(counts) - counts
    packed = np.stack([base, counts], -1).astype(np.int64); S_ = int(counts.sum())
    t0 = np.sort(rng.uniform(0, 4, S_)).astype(np.float32); t1 = (t0 + 5e-3).astype(np.float32)
    sig = (rng.uniform(0, 1, S_) ** 3 * 500).astype(np.float32); rgbs = rng.uniform(0, 1, (S_, 3)).astype(np.float32)
    pf = rng.uniform(0, 1, S_).astype(np.float32)
    for prefix in (None, pf):
        wa = oracle.render_weight_from_density(t0, t1, sig, packed, prefix)
        wb = TO.render_weight_from_density(t0, t1, sig, packed, prefix)
        for x_, y_ in zip(wa, wb):
            assert np.abs(x_ - y_).max() < 2e-6
    acc_a = oracle.accumulate_along_rays_(wa[0], rgbs, packed, np.zeros((400, 3), np.float32))
    assert np.abs(acc_a - TO.accumulate_along_rays(wa[0], rgbs, packed)).max() < 1e-5


# ---- (4) drivers -------------------------------------------------------------------------------
def test_render_drivers_self_consistent(oracle):
    """Appendix B.7: render_image and render_image_test agree to 1e-4 except on the few rays where
    a restarted DDA flips a float comparison at a cell boundary (one sample gained or lost)."""
    from ced_nerf_amd import synthetic as S
    from oracle import torch_oracle as TO
    sc = S.make_scene("dnerf", 72, 54, "trained", log2_hashmap_size=15)
    cfg = sc["cfg"]
    f = oracle.OracleField(sc["params"]); est = oracle.OracleEstimator(cfg["aabb"], 128, 1, sc["binaries"])
    a = oracle.render_image_test(1024, f, est, sc["origins"], sc["viewdirs"], timestamps=sc["timestamps"], **sc["render"])
    b = oracle.render_image(f, est, sc["origins"], sc["viewdirs"], timestamps=sc["timestamps"], **sc["render"])
    assert a[0].shape == (54, 72, 3) and a[1].shape == (54, 72, 1) and b[3] < b[5]
    bad = (np.abs(a[0] - b[0]).max(-1) > 1e-4).mean()
    assert bad < 2e-3, bad
    assert 0.05 < a[1].mean() < 0.6 and a[0].std() > 0.05
    # the PyTorch path renders the same image
    c = TO.render_image_test(1024, TO.TorchField(sc["params"]), est, sc["origins"], sc["viewdirs"],
                             timestamps=sc["timestamps"], **sc["render"])
    assert (np.abs(a[0] - c[0]).max(-1) > 2e-4).mean() < 5e-3
    assert abs(a[3] - c[3]) <= 0.01 * a[3]


def test_empty_and_ragged_inputs(oracle):
    from ced_nerf_amd import synthetic as S
    sc = S.make_scene("dnerf", 8, 6, "init", log2_hashmap_size=12)
    cfg = sc["cfg"]
    f = oracle.OracleField(sc["params"])
    empty = oracle.OracleEstimator(cfg["aabb"], 128, 1, np.zeros_like(sc["binaries"]))
    out = oracle.render_image_test(64, f, empty, sc["origins"], sc["viewdirs"], timestamps=sc["timestamps"], **sc["render"])
    assert out[3] == 0 and np.all(out[0] == 1.0) and np.all(out[1] == 0) and np.all(out[2] == 0)
    out = oracle.render_image(f, empty, sc["origins"], sc["viewdirs"], timestamps=sc["timestamps"], **sc["render"])
    assert out[3] == 0 and np.all(out[0] == 1.0)


def test_oracle_fp16_rounding_matches_numpy(oracle):
    """The fp16-operand MLP mode of the oracle rests on this rounding: check it against numpy's
    float16 conversion (ties to even, subnormals), plus the saturation at 65504."""
    rng = np.random.default_rng(1)
    x = (rng.normal(size=300000) * 10.0 ** rng.uniform(-9, 5, 300000)).astype(np.float32)
    x = np.concatenate([x, np.float32([0, 65504, 65519.9, 6.1e-5, 6.103515625e-05, 5.96e-8, 2.98e-8, 2.9802322e-8, 3e-8,
                                       8.9e-8, 1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11])])
    y = oracle.round_f16(x)
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(np.float32)
    ref = np.where(np.isinf(ref), np.sign(x) * np.float32(65504), ref)
    assert np.array_equal(y, ref)
    assert np.array_equal(oracle.round_f16(np.float32([65520, 1e9, -1e9])), np.float32([65504, 65504, -65504]))
    assert np.signbit(oracle.round_f16(np.float32([-0.0]))[0])


def test_oracle_hash_backward_is_the_adjoint_of_the_forward(oracle):
    """hash_encoder_backward_kernel restated (hash_encoder_half.py:164-226): the table gradient is the adjoint of the
    (table-linear) forward, and the position gradient matches finite differences of the forward in the scaled
    position (the reference omits the `scale` factor) away from cell faces."""
    from ced_nerf_amd import synthetic as S
    p = S.init_field_params([-1, -1, -1, 1, 1, 1], 1e-4, 256, 12, regime="trained", seed=3)
    of = oracle.OracleField({"hash": p["hash"]})
    rng = np.random.default_rng(2)
    n = 400
    x = rng.uniform(0.05, 0.95, size=(n, 3)).astype(np.float32)
    dy = rng.normal(size=(n, 32)).astype(np.float32)
    grad, dx = of.hash_encode_backward(x, dy)
    # adjoint: <dy, enc(T + dT) - enc(T)> == <grad, dT>
    dT = rng.normal(size=p["hash"]["table"].shape).astype(np.float32) * 0.1
    h2 = dict(p["hash"]); h2["table"] = (p["hash"]["table"] + dT).astype(np.float32)
    of2 = oracle.OracleField({"hash": h2})
    lhs = ((of2.hash_encode(x).astype(np.float64) - of.hash_encode(x).astype(np.float64)) * dy).sum()
    rhs = (grad * dT).sum()
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), abs(rhs))
    # position gradient per level l is d/d(pos_l); finite differences in x give sum_l scale_l * (that); check the
    # single-level case by zeroing all but one level's dy
    for l in (0, 5, 11):
        dyl = np.zeros_like(dy); dyl[:, 2 * l:2 * l + 2] = dy[:, 2 * l:2 * l + 2]
        _, dxl = of.hash_encode_backward(x, dyl)
        sc = float(of.levels["scale"][l])
        frac = (x * np.float32(sc) + np.float32(0.5)) % 1.0
        ok = ((frac > 0.1) & (frac < 0.9)).all(axis=1)
        eps = 1e-3 / sc
        for a in range(3):
            xp = x.copy(); xp[:, a] += eps
            xm = x.copy(); xm[:, a] -= eps
            fd = ((of.hash_encode(xp).astype(np.float64) - of.hash_encode(xm).astype(np.float64)) * dyl).sum(axis=1) / (2 * eps * sc)
            err = np.abs(fd[ok] - dxl[ok, a]).max() / max(np.abs(dxl[ok, a]).max(), 1e-6)
            assert err < 2e-2, (l, a, err)
        _, dxs = of.hash_encode_backward(x, dyl, dx_scaled=True)          # times d pos / d x = scale
        assert np.allclose(dxs, dxl * np.float32(sc), rtol=1e-4, atol=1e-6 * float(np.abs(dxs).max()))


def test_weight_grad_oracle_is_the_plain_sum(oracle):
    """ced_o_weight_grad (checker of the training path's dW kernel) equals numpy's float64 dy^T x; empty input -> zeros."""
    rng = np.random.default_rng(3)
    for n, n_out, n_in in [(0, 4, 5), (1, 1, 1), (257, 6, 64), (1000, 64, 19)]:
        x = rng.normal(size=(n, n_in)).astype(np.float32); dy = rng.normal(size=(n, n_out)).astype(np.float32)
        got = oracle.weight_grad(x, dy)
        want = dy.astype(np.float64).T @ x.astype(np.float64)
        assert got.shape == (n_out, n_in)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)


def test_frame_conversion_oracle_known_answers(oracle):
    """The 8-bit frame conversions of the video step (train_real.py:38-41,556-557): truncation, the width flip, the
    min-max normalisation."""
    rgb = np.array([[[0.0, 0.5, 1.0], [0.999, 0.25, 0.003]]], np.float32)               # [1,2,3]
    got = oracle.frame_to_rgb8(rgb)
    assert got.tolist() == [[[254, 63, 0], [0, 127, 255]]]
    assert oracle.frame_to_rgb8(rgb, False).tolist() == [[[0, 127, 255], [254, 63, 0]]]
    d = np.array([[1.0, 2.0, 3.0, 5.0]], np.float32)
    assert oracle.depth_to_u8(d, False).tolist() == [[0, 63, 127, 255]]
    assert oracle.depth_to_u8(d).tolist() == [[255, 127, 63, 0]]
