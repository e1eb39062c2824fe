"""Reference checkpoints (`model.pth`, train_real.py:433-441, read back at :524-529) -> this package's modules.

The reference saves `{"radiance_field": DNGPradianceField.state_dict(), "occupancy_grid": OccGridEstimator.state_dict()}`.
The estimator half are plain tensors under nerfacc's own names (`resolution, aabbs, occs, binaries`) and loads into
`nerfacc_api.OccGridEstimator` as it is.  The field half is tiny-cuda-nn's: every tcnn module exposes ONE flat
float32 `params` tensor (`hash_encoder.params`, `xyz_wrap.params`, `mlp_base.params`, `mlp_head.params`; the
SphericalHarmonics `direction_encoding.params` is empty), whose internal layout is tiny-cuda-nn's business.

**The layout below is a hypothesis.**  tiny-cuda-nn is not vendored by the reference, not pinned ("git master",
cednerf/model.py:19-21), not installed here and cannot be fetched; no `model.pth` exists in /root/reference.  What is
written down is the recollection of tiny-cuda-nn v1.6 recorded in SURVEY.md Appendix A.6-A.8 -- it has never been checked
against a file written by the real thing, and a wrong guess renders garbage without any error.  Hence
`assume_tcnn_layout=TCNN_LAYOUT` is a REQUIRED argument: the caller states the assumption, it is not a default.

TCNN_LAYOUT = "tcnn-v1.6:grid-level-major,mlp-rowmajor-out-in,pad16,ones":
  * HashGrid `params`: float32, levels back to back, level l holding `size[l] * 2` values (entry-major, the two
    features of an entry adjacent); sizes / resolutions / dense-vs-hashed as `hashgrid.level_tables` (the Taichi spec
    cednerf/taichi_kernel/hash_encoder_half.py:12-35,268-292 mirrors tcnn's grid).
  * FullyFusedMLP `params`: the weight matrices back to back, first layer first; every matrix row-major `[out][in]`;
    the first layer's input width padded to a multiple of 16, the last layer's output width padded to a multiple of 16
    (extra rows are dropped here); hidden width 64.  `NetworkWithInputEncoding` (xyz_wrap): the Frequency encoding has
    no parameters, so its `params` are the network's.
  * Padded INPUT columns see the constant 1 (tcnn pads encodings with ones), i.e. they are a learnt bias.  The kernels
    here are bias-free with zero padding.  mlp_head (19 -> 32): its first input is the constant SH coefficient
    Y00 = 0.28209479177387814 (tcnn SphericalHarmonics), so the bias is folded into that column exactly
    (in real arithmetic): W[:, 0] += sum(W[:, 19:32]) / Y00.  xyz_wrap (32) and mlp_base without the time embedding (32)
    have no padding.  mlp_base WITH the time embedding (41 -> 48, run_hyper.sh's `-te`) has seven such columns and no
    constant input to fold them into: such a checkpoint is refused unless those columns cancel.
  * fp16: tcnn evaluates in fp16 with fp16 accumulators; the master `params` in the checkpoint are fp32 and are taken
    as they are (use mlp_precision="f16" for the reference's precision class).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from .hashgrid import level_tables

TCNN_LAYOUT = "tcnn-v1.6:grid-level-major,mlp-rowmajor-out-in,pad16,ones"
SH_Y00 = 0.28209479177387814
WIDTH = 64


def _pad16(n: int) -> int:
    return (n + 15) // 16 * 16


def _require_layout(assume_tcnn_layout: Optional[str]) -> None:
    if assume_tcnn_layout != TCNN_LAYOUT:
        raise ValueError(
            "reference checkpoints hold tiny-cuda-nn's flat `params` tensors, whose layout cannot be verified offline; "
            f"pass assume_tcnn_layout={TCNN_LAYOUT!r} to state that the file follows the layout documented in "
            "ced_nerf_amd/checkpoint.py (unverified against a real model.pth)")


def split_tcnn_mlp(params: torch.Tensor, n_in: int, n_out: int, n_hidden_layers: int) -> Tuple[List[np.ndarray], np.ndarray]:
    """Flat FullyFusedMLP params -> ([W0 [64,n_in], W_hidden ..., W_last [n_out,64]], pad_bias [64]) where pad_bias is
    the contribution of the ones-padded input columns of the first layer (zero when n_in is a multiple of 16)."""
    p = params.detach().to(torch.float32).cpu().numpy().reshape(-1)
    in_pad, out_pad = _pad16(n_in), _pad16(n_out)
    want = WIDTH * in_pad + (n_hidden_layers - 1) * WIDTH * WIDTH + out_pad * WIDTH
    if p.size != want:
        raise ValueError(f"FullyFusedMLP {n_in}->{n_out} with {n_hidden_layers} hidden layer(s): expected {want} params "
                         f"({WIDTH}x{in_pad} + {n_hidden_layers - 1}x{WIDTH}x{WIDTH} + {out_pad}x{WIDTH}), got {p.size}")
    mats, off = [], 0
    w0 = p[off:off + WIDTH * in_pad].reshape(WIDTH, in_pad); off += WIDTH * in_pad
    mats.append(np.ascontiguousarray(w0[:, :n_in]))
    pad_bias = w0[:, n_in:].astype(np.float64).sum(axis=1).astype(np.float32)
    for _ in range(n_hidden_layers - 1):
        mats.append(np.ascontiguousarray(p[off:off + WIDTH * WIDTH].reshape(WIDTH, WIDTH))); off += WIDTH * WIDTH
    mats.append(np.ascontiguousarray(p[off:off + out_pad * WIDTH].reshape(out_pad, WIDTH)[:n_out]))
    return mats, pad_bias


def join_tcnn_mlp(mats: List[np.ndarray], pad_bias: Optional[np.ndarray] = None) -> torch.Tensor:
    """Inverse of split_tcnn_mlp: natural matrices -> flat params with 16-padding.  pad_bias (if given) is put into
    the first padded input column, the other padded entries are zero."""
    n_in, n_out = mats[0].shape[1], mats[-1].shape[0]
    in_pad, out_pad = _pad16(n_in), _pad16(n_out)
    w0 = np.zeros((WIDTH, in_pad), np.float32); w0[:, :n_in] = mats[0]
    if pad_bias is not None:
        if in_pad == n_in:
            raise ValueError("no padded input column to carry a bias")
        w0[:, n_in] = pad_bias
    wl = np.zeros((out_pad, WIDTH), np.float32); wl[:n_out] = mats[-1]
    parts = [w0.reshape(-1)] + [np.asarray(m, np.float32).reshape(-1) for m in mats[1:-1]] + [wl.reshape(-1)]
    return torch.from_numpy(np.concatenate(parts))


def field_params_from_reference_state(state: Dict[str, torch.Tensor], *, log2_hashmap_size: int, dst_resolution: int,
                                      base_resolution: int = 16, n_levels: int = 16, use_div_offsets: bool = False,
                                      use_time_embedding: bool = False, assume_tcnn_layout: Optional[str] = None) -> Dict:
    """The `radiance_field` state dict of a reference checkpoint -> {hash_table [E,2], xyz_wrap [4], mlp_base [2],
    mlp_head [3], aabb} in this package's natural layout (numpy float32)."""
    _require_layout(assume_tcnn_layout)
    tabs = level_tables(base_resolution, dst_resolution, n_levels, log2_hashmap_size)
    grid = state["hash_encoder.params"].detach().to(torch.float32).cpu().numpy().reshape(-1)
    if grid.size != tabs["total"] * 2:
        raise ValueError(f"hash_encoder.params holds {grid.size} values; a grid with dst_resolution={dst_resolution}, "
                         f"log2_hashmap_size={log2_hashmap_size} has {tabs['total']} entries x 2")
    n_mo = 6 if use_div_offsets else 3
    base_in = 41 if use_time_embedding else 32
    xyz, b_xyz = split_tcnn_mlp(state["xyz_wrap.params"], 32, n_mo, 3)
    base, b_base = split_tcnn_mlp(state["mlp_base.params"], base_in, 16, 1)
    head, b_head = split_tcnn_mlp(state["mlp_head.params"], 19, 3, 2)
    assert not b_xyz.any()
    if np.abs(b_base).max(initial=0.0) > 1e-6 * max(1.0, float(np.abs(base[0]).max())):
        raise NotImplementedError(
            "mlp_base with the time embedding has 41 inputs padded to 48 with ones: the seven padded columns of this "
            "checkpoint act as a bias (|sum| up to %.3g) that the bias-free kernels cannot express" % np.abs(b_base).max())
    # mlp_head: the ones-padded columns are a bias; input 0 is the constant Y00, so the bias rides on that column
    head[0] = head[0].copy()
    head[0][:, 0] = (head[0][:, 0].astype(np.float64) + b_head.astype(np.float64) / SH_Y00).astype(np.float32)
    out = dict(hash_table=np.ascontiguousarray(grid.reshape(-1, 2)), xyz_wrap=xyz, mlp_base=base, mlp_head=head)
    if "aabb" in state:
        out["aabb"] = state["aabb"].detach().to(torch.float32).cpu().numpy()
    return out


def reference_state_from_field(field, head_bias: Optional[np.ndarray] = None) -> Dict[str, torch.Tensor]:
    """A `radiance_field` state dict in TCNN_LAYOUT from a DNGPradianceField of this package (the inverse map: what a
    reference checkpoint of these weights would look like under the hypothesis).  head_bias [64]: put a bias into
    mlp_head's ones-padded input column (and take the same amount out of the Y00 column, so that the function is
    unchanged) -- exercises the folding of `field_params_from_reference_state`."""
    g = lambda p: p.detach().to(torch.float32).cpu().numpy()
    if field.hash_cfg.get("temporal"):
        raise NotImplementedError("the temporal hash table has no tiny-cuda-nn counterpart")
    head = [g(p) for p in field.mlp_head]
    if head_bias is not None:
        head[0] = head[0].copy()
        head[0][:, 0] = (head[0][:, 0].astype(np.float64) - np.asarray(head_bias, np.float64) / SH_Y00).astype(np.float32)
    sd = {
        "aabb": field.aabb.detach().cpu().clone(),
        "hash_encoder.params": torch.from_numpy(g(field.hash_table).reshape(-1).copy()),
        "xyz_wrap.params": join_tcnn_mlp([g(p) for p in field.xyz_wrap]),
        "mlp_base.params": join_tcnn_mlp([g(p) for p in field.mlp_base]),
        "mlp_head.params": join_tcnn_mlp(head, None if head_bias is None else np.asarray(head_bias, np.float32)),
        "direction_encoding.params": torch.zeros((0,), dtype=torch.float32),
    }
    return sd


@torch.no_grad()
def load_reference_checkpoint(path_or_state, radiance_field, estimator=None, *, assume_tcnn_layout: Optional[str] = None,
                              map_location="cpu") -> None:
    """`radiance_field.load_state_dict(checkpoint["radiance_field"]); estimator.load_state_dict(checkpoint[
    "occupancy_grid"])` of train_real.py:524-529 for a checkpoint written by the REFERENCE (tiny-cuda-nn parameters),
    into modules of this package constructed with the same flags (train_real.py:252-265).  See the module docstring:
    the tcnn layout is an unverified hypothesis and must be named explicitly."""
    _require_layout(assume_tcnn_layout)
    # weights_only: the reference's checkpoints hold tensors and python scalars only (train_real.py:433-441); never run a
    # pickle's code for them
    ckpt = torch.load(path_or_state, map_location=map_location, weights_only=True) if isinstance(path_or_state, (str, bytes)) \
        or hasattr(path_or_state, "read") else path_or_state
    sd = ckpt["radiance_field"] if "radiance_field" in ckpt else ckpt
    cfg = radiance_field.hash_cfg
    if radiance_field.hash_table.dtype != torch.float32 and radiance_field.hash_table.dtype != torch.float16:
        raise ValueError("unsupported hash table dtype")
    p = field_params_from_reference_state(
        sd, log2_hashmap_size=cfg["log2_hashmap_size"], dst_resolution=cfg["max_res"], base_resolution=cfg["base_res"],
        n_levels=cfg["n_levels"], use_div_offsets=radiance_field.use_div_offsets,
        use_time_embedding=radiance_field.use_time_embedding, assume_tcnn_layout=assume_tcnn_layout)
    dev = radiance_field.hash_table.device
    radiance_field.hash_table.copy_(torch.from_numpy(p["hash_table"]).to(dev, radiance_field.hash_table.dtype))
    for dst, src in ((radiance_field.xyz_wrap, p["xyz_wrap"]), (radiance_field.mlp_base, p["mlp_base"]),
                     (radiance_field.mlp_head, p["mlp_head"])):
        for q, w in zip(dst, src):
            if tuple(q.shape) != tuple(w.shape):
                raise ValueError(f"layer shape {tuple(w.shape)} does not fit the module's {tuple(q.shape)}")
            q.copy_(torch.from_numpy(w).to(dev))            # in place through the parameter: its version counter advances
    if "aabb" in p:
        radiance_field.aabb.copy_(torch.from_numpy(p["aabb"]).to(dev))
    if estimator is not None and "occupancy_grid" in ckpt:
        occ = {k: v for k, v in ckpt["occupancy_grid"].items() if k in ("resolution", "aabbs", "occs", "binaries")}
        estimator.load_state_dict(occ, strict=False)


# ----------------------------------------------------------------------------------------------------------------------
# A falsifiability gate for the layout hypothesis (tools/verify_checkpoint.py)
# ----------------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def checkpoint_consistency(radiance_field, estimator, timestamps: Optional[torch.Tensor] = None, render_step_size: float = 5e-3,
                           occ_thre: float = 1e-2, n_cells: int = 20000, seed: int = 0, points_per_cell: int = 4) -> Dict:
    """Does the loaded FIELD agree with the loaded OCCUPANCY GRID?  The two halves of a `model.pth` are stored
    independently (train_real.py:433-441): the grid's `binaries` were thresholded from the field's own density during
    training (`occ_eval_fn`, train_real.py:324-336: density(x, t) * render_step_size against `occ_thre`).  Under the right
    parameter layout the field is dense where the grid says occupied and empty where it says free; under a wrong one
    (levels permuted, a matrix transposed, features swapped) the density has no relation to the grid and the two rates
    below coincide.  Nothing here needs the training data.

    Per grid level, `n_cells` occupied and `n_cells` free cells are drawn; the density is evaluated at `points_per_cell`
    points of each cell (the centre and jittered points, as the trainer's `_update` samples cells at random positions) at
    each of `timestamps` (default 0, 0.25, .., 1), and the maximum over points and time is compared with `occ_thre` as the
    trainer did.
      hit_rate_occupied   fraction of occupied cells whose density * step exceeds the threshold
      hit_rate_free       the same for free cells
      separation          hit_rate_occupied - hit_rate_free     (1: perfect agreement, ~0: unrelated)
      auc                 P(density of a random occupied cell > density of a random free cell)   (0.5: unrelated)
    plus range statistics of sigma and rgb at the sampled points (a dead or saturated network shows here).
    `verdict`: "consistent" (separation >= 0.5 and auc >= 0.8), "inconsistent" (separation < 0.2 or auc < 0.65), else "unclear"."""
    dev = radiance_field.hash_table.device
    g = torch.Generator(device="cpu").manual_seed(seed)
    ts = timestamps if timestamps is not None else torch.linspace(0.0, 1.0, 5)
    ts = ts.reshape(-1).to(dev, torch.float32)
    binaries = estimator.binaries                       # [levels, R, R, R] bool
    aabbs = estimator.aabbs.to(dev)
    levels, R = binaries.shape[0], binaries.shape[1]
    out_levels, occ_all, free_all, sig_all, rgb_all = [], [], [], [], []
    for lvl in range(levels):
        flat = binaries[lvl].reshape(-1).to(dev)
        idx_occ = torch.nonzero(flat).reshape(-1)
        idx_free = torch.nonzero(~flat).reshape(-1)
        if idx_occ.numel() == 0 or idx_free.numel() == 0:
            out_levels.append({"level": lvl, "occupied_cells": int(idx_occ.numel()), "skipped": "one class is empty"})
            continue
        pick = lambda idx: idx[torch.randint(0, idx.numel(), (min(n_cells, idx.numel()),), generator=g).to(dev)]  # noqa: E731
        dens = {}
        for name, idx in (("occ", pick(idx_occ)), ("free", pick(idx_free))):
            ijk = torch.stack([idx // (R * R), (idx // R) % R, idx % R], dim=-1).to(torch.float32)
            best = torch.zeros(ijk.shape[0], device=dev)
            for k in range(max(1, points_per_cell)):
                off = 0.5 if k == 0 else torch.rand(ijk.shape[0], 3, generator=g).to(dev)
                x = aabbs[lvl, :3] + (ijk + off) / R * (aabbs[lvl, 3:] - aabbs[lvl, :3])
                for t in ts:
                    tt = t.expand(x.shape[0], 1).contiguous()
                    d = radiance_field.query_density(x.contiguous(), tt)["density"].reshape(-1)
                    best = torch.maximum(best, d)
            dens[name] = best
            sig_all.append(best)
        dirs = torch.nn.functional.normalize(torch.randn(dens["occ"].shape[0], 3, generator=g), dim=-1).to(dev)
        idx = pick(idx_occ)[:dirs.shape[0]]
        ijk = torch.stack([idx // (R * R), (idx // R) % R, idx % R], dim=-1).to(torch.float32)
        x = aabbs[lvl, :3] + (ijk + 0.5) / R * (aabbs[lvl, 3:] - aabbs[lvl, :3])
        rgb, _ = radiance_field(x[:dirs.shape[0]].contiguous(), ts[:1].expand(dirs.shape[0], 1).contiguous(), dirs[:x.shape[0]].contiguous())
        rgb_all.append(rgb)
        hit_o = float((dens["occ"] * render_step_size > occ_thre).float().mean())
        hit_f = float((dens["free"] * render_step_size > occ_thre).float().mean())
        # AUC by ranks (ties count half)
        both = torch.cat([dens["occ"], dens["free"]])
        ranks = torch.empty_like(both)
        order = torch.argsort(both)
        ranks[order] = torch.arange(1, both.numel() + 1, device=dev, dtype=both.dtype)
        n_o, n_f = dens["occ"].numel(), dens["free"].numel()
        auc = float((ranks[:n_o].sum() - n_o * (n_o + 1) / 2.0) / (n_o * n_f))
        out_levels.append({"level": lvl, "occupied_cells": int(idx_occ.numel()), "free_cells": int(idx_free.numel()),
                           "hit_rate_occupied": hit_o, "hit_rate_free": hit_f, "separation": hit_o - hit_f, "auc": auc})
        occ_all.append(dens["occ"]); free_all.append(dens["free"])
    scored = [l for l in out_levels if "separation" in l]
    if not scored:
        return {"levels": out_levels, "verdict": "unclear", "reason": "no grid level has both occupied and free cells"}
    sep = min(l["separation"] for l in scored)
    auc = min(l["auc"] for l in scored)
    sig = torch.cat(sig_all); rgb = torch.cat(rgb_all)
    q = lambda t, p: float(torch.quantile(t.float().reshape(-1)[:1_000_000], p))  # noqa: E731
    verdict = "consistent" if (sep >= 0.5 and auc >= 0.8) else ("inconsistent" if (sep < 0.2 or auc < 0.65) else "unclear")
    return {"levels": out_levels, "separation_min": sep, "auc_min": auc, "verdict": verdict,
            "sigma": {"min": float(sig.min()), "p50": q(sig, 0.5), "p99": q(sig, 0.99), "max": float(sig.max()),
                      "fraction_zero": float((sig == 0).float().mean()), "finite": bool(torch.isfinite(sig).all())},
            "rgb": {"min": float(rgb.min()), "mean": float(rgb.mean()), "max": float(rgb.max()), "std": float(rgb.std()),
                    "finite": bool(torch.isfinite(rgb).all())},
            "render_step_size": render_step_size, "occ_thre": occ_thre, "timestamps": [float(v) for v in ts]}
