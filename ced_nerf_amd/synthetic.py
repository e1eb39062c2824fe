"""Synthetic stand-ins for the reference's scenes (no dataset is available offline).

Everything here is *input data* generation: cameras with the real scenes' geometry
constants, an analytic occupancy grid and deterministic field parameters.  numpy only, seeded, so
the CPU oracle and the HIP path see bit-identical inputs.  Citations are relative to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

# per-dataset constants, train_real.py:86-175
CONFIGS: Dict[str, Dict] = {
    # D-NeRF synthetic (lego, trex, ...): train_real.py:86-117, dnerf_synthetic.py:54-55,156
    "dnerf": dict(aabb=[-1.5, -1.5, -1.5, 1.5, 1.5, 1.5], near_plane=0.0, far_plane=1e10, moving_step=1e-4,
                  hash_max_res=1024, grid_resolution=128, grid_levels=1, render_step_size=5e-3,
                  alpha_thre=0.0, cone_angle=0.0, bkgd=[1.0, 1.0, 1.0], opengl=True,
                  camera_angle_x=0.6911112070083618, radius=4.0, flags=dict()),
    # HyperNeRF vrig: train_real.py:119-150, run_hyper.sh:1 (-te -ta -df)
    "hypernerf": dict(aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], near_plane=0.2, far_plane=1e10,
                      moving_step=1.0 / 4096, hash_max_res=4096, grid_resolution=128, grid_levels=2,
                      render_step_size=1e-3, alpha_thre=1e-2, cone_angle=0.004, bkgd=[0.0, 0.0, 0.0],
                      opengl=False, camera_angle_x=0.8, radius=1.6,
                      flags=dict(use_time_embedding=True, use_time_attenuation=True, use_div_offsets=True)),
    # DyNeRF (Plenoptic video): train_real.py:152-182
    "dynerf": dict(aabb=[-1.0, -1.0, -1.0, 1.0, 1.0, 1.0], near_plane=0.2, far_plane=1e10,
                   moving_step=1.0 / 8192, hash_max_res=8192, grid_resolution=128, grid_levels=4,
                   render_step_size=1e-3, alpha_thre=1e-2, cone_angle=0.004, bkgd=[0.0, 0.0, 0.0],
                   opengl=False, camera_angle_x=0.9, radius=2.5,
                   flags=dict(use_time_embedding=True, use_time_attenuation=True, use_div_offsets=True)),
}


def look_at_c2w(radius: float, elev_deg: float, azim_deg: float, opengl: bool = True) -> np.ndarray:
    """Camera-to-world [3,4] looking at the origin from a sphere of the given radius."""
    el, az = math.radians(elev_deg), math.radians(azim_deg)
    pos = np.array([radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az),
                    radius * math.sin(el)], np.float64)
    fwd = -pos / np.linalg.norm(pos)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up); right /= np.linalg.norm(right)
    true_up = np.cross(right, fwd)
    if opengl:   # camera looks down -z, y up
        rot = np.stack([right, true_up, -fwd], 1)
    else:        # OpenCV: camera looks down +z, y down
        rot = np.stack([right, -true_up, fwd], 1)
    return np.concatenate([rot, pos[:, None]], 1).astype(np.float32)


def make_camera_rays(width: int, height: int, camera_angle_x: float, c2w: np.ndarray,
                     opengl: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """Pinhole rays for one full image, float32 [H,W,3] (origins, viewdirs).

    Follows datasets/dnerf_synthetic.py:54-55 (focal), :191-221 (pixel grid "xy", +0.5 centre,
    OpenGL sign flips, rotate by c2w[:3,:3], normalise); gui.py:43-86 is the same formula."""
    f32 = np.float32
    focal = f32(0.5 * width / math.tan(0.5 * camera_angle_x))
    cx, cy = f32(width / 2.0), f32(height / 2.0)
    x, y = np.meshgrid(np.arange(width, dtype=f32), np.arange(height, dtype=f32), indexing="xy")
    x = x.reshape(-1); y = y.reshape(-1)
    sgn = f32(-1.0 if opengl else 1.0)
    cam = np.stack([(x - cx + f32(0.5)) / focal, (y - cy + f32(0.5)) / focal * sgn,
                    np.full_like(x, sgn)], -1).astype(f32)
    rot = c2w[:3, :3].astype(f32)
    dirs = (cam[:, None, :] * rot[None, :, :]).sum(-1).astype(f32)
    origins = np.broadcast_to(c2w[:3, 3].astype(f32), dirs.shape).copy()
    viewdirs = (dirs / np.linalg.norm(dirs, axis=-1, keepdims=True)).astype(f32)
    return origins.reshape(height, width, 3), viewdirs.reshape(height, width, 3)


def enlarge_aabb(aabb, factor) -> np.ndarray:
    aabb = np.asarray(aabb, np.float32)
    c = (aabb[:3] + aabb[3:]) / np.float32(2)
    e = (aabb[3:] - aabb[:3]) / np.float32(2)
    return np.concatenate([c - e * np.float32(factor), c + e * np.float32(factor)]).astype(np.float32)


# three spheres inside the unit roi (centre, radius) in roi-normalised [-1,1] coordinates
_SPHERES = [((0.05, -0.10, 0.00), 0.35), ((-0.45, 0.35, 0.15), 0.25), ((0.40, 0.45, -0.30), 0.20)]


def make_occupancy(roi_aabb: Sequence[float], resolution: int = 128, levels: int = 1,
                   spheres=None) -> np.ndarray:
    """Analytic lego-like occupancy: cells whose centre is inside a union of three spheres
    (radii 0.35/0.25/0.2 in roi half-extent units).  Returns bool [levels,R,R,R]; level i covers
    the roi enlarged 2**i (nerfacc OccGridEstimator layout, x-major)."""
    spheres = _SPHERES if spheres is None else spheres
    roi = np.asarray(roi_aabb, np.float64)
    half = (roi[3:] - roi[:3]) / 2.0
    centre = (roi[3:] + roi[:3]) / 2.0
    out = np.zeros((levels, resolution, resolution, resolution), bool)
    for lvl in range(levels):
        ab = enlarge_aabb(roi_aabb, 2 ** lvl).astype(np.float64)
        axes = [ab[a] + (np.arange(resolution) + 0.5) * (ab[3 + a] - ab[a]) / resolution for a in range(3)]
        X, Y, Z = np.meshgrid(*axes, indexing="ij")
        occ = np.zeros_like(X, bool)
        for (c, r) in spheres:
            cw = centre + np.asarray(c) * half
            rw = r * half.min()
            occ |= ((X - cw[0]) ** 2 + (Y - cw[1]) ** 2 + (Z - cw[2]) ** 2) <= rw * rw
        out[lvl] = occ
    return out


def hash_total_entries(base_res: int, max_res: int, n_levels: int, log2_hashmap_size: int) -> int:
    from .hashgrid import level_tables
    return int(level_tables(base_res, max_res, n_levels, log2_hashmap_size)["total"])


def _xavier(rng: np.random.Generator, n_out: int, n_in: int) -> np.ndarray:
    lim = math.sqrt(6.0 / (n_in + n_out))
    return rng.uniform(-lim, lim, size=(n_out, n_in)).astype(np.float32)


def init_field_params(aabb, moving_step: float, hash_max_res: int = 1024, log2_hashmap_size: int = 21,
                      use_div_offsets: bool = False, use_time_embedding: bool = False,
                      use_time_attenuation: bool = False, regime: str = "init", seed: int = 42,
                      table_dtype=np.float32, n_levels: int = 16, base_res: int = 16,
                      temporal_hash: bool = False) -> Dict:
    """Deterministic DNGPradianceField parameters (model.py:100-309 dims; SURVEY 8d regimes).

    regime "init": hash U(-1e-4,1e-4) (hash_encoder_half.py:313), Xavier-uniform bias-free MLPs.
    regime "trained": hash N(0,0.5^2), mlp_base output row 0 (density) scaled x32 so sigma spans
    0..1e3 and rays terminate early, as in a trained scene; mlp_head output scaled x16 so colours
    use the whole (0,1) range."""
    rng = np.random.default_rng(seed)
    total = hash_total_entries(base_res, hash_max_res, n_levels, log2_hashmap_size)
    width = 8 if temporal_hash else 2
    if regime == "init":
        table = rng.uniform(-1e-4, 1e-4, size=(total, width)).astype(np.float32)
    elif regime == "trained":
        table = (rng.standard_normal(size=(total, width), dtype=np.float32) * np.float32(0.5))
    else:
        raise ValueError(regime)
    table = table.astype(table_dtype)
    time_mode = 0
    if use_time_embedding:
        time_mode = 2 if use_time_attenuation else 1
    base_in = 32 + (9 if time_mode else 0)
    n_mo = 6 if use_div_offsets else 3
    xyz_wrap = [_xavier(rng, 64, 32), _xavier(rng, 64, 64), _xavier(rng, 64, 64), _xavier(rng, n_mo, 64)]
    mlp_base = [_xavier(rng, 64, base_in), _xavier(rng, 16, 64)]
    mlp_head = [_xavier(rng, 64, 19), _xavier(rng, 64, 64), _xavier(rng, 3, 64)]
    if regime == "trained":
        mlp_base[1][0, :] *= np.float32(32.0)
        mlp_head[2] *= np.float32(16.0)
    return dict(aabb=np.asarray(aabb, np.float32), moving_step=float(moving_step),
                use_div_offsets=bool(use_div_offsets), time_mode=time_mode,
                hash=dict(base_res=base_res, max_res=hash_max_res, n_levels=n_levels,
                          log2_hashmap_size=log2_hashmap_size, table=table, temporal=bool(temporal_hash)),
                xyz_wrap=xyz_wrap, mlp_base=mlp_base, mlp_head=mlp_head)


def make_scene(name: str = "dnerf", width: int = 800, height: int = 800, regime: str = "trained",
               azim_deg: float = 30.0, elev_deg: float = 30.0, seed: int = 42, table_dtype=np.float32,
               log2_hashmap_size: int = 21, timestamp: float = 0.5) -> Dict:
    """One synthetic frame of a named configuration: rays, occupancy grid, field parameters and
    the render options train_real.py uses for that dataset."""
    cfg = CONFIGS[name]
    c2w = look_at_c2w(cfg["radius"], elev_deg, azim_deg, cfg["opengl"])
    origins, viewdirs = make_camera_rays(width, height, cfg["camera_angle_x"], c2w, cfg["opengl"])
    binaries = make_occupancy(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"])
    field_aabb = enlarge_aabb(cfg["aabb"], 2 ** (cfg["grid_levels"] - 1))   # estimator.aabbs[-1], train_real.py:254
    params = init_field_params(field_aabb, cfg["moving_step"], cfg["hash_max_res"], log2_hashmap_size,
                               regime=regime, seed=seed, table_dtype=table_dtype, **cfg["flags"])
    return dict(name=name, cfg=cfg, origins=origins, viewdirs=viewdirs, binaries=binaries, params=params,
                timestamps=np.array([[timestamp]], np.float32),
                render=dict(near_plane=cfg["near_plane"], far_plane=cfg["far_plane"],
                            render_step_size=cfg["render_step_size"], cone_angle=cfg["cone_angle"],
                            alpha_thre=cfg["alpha_thre"], render_bkgd=np.asarray(cfg["bkgd"], np.float32)))
