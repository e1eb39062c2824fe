"""nerfacc-shaped operators backed by the HIP kernels.

The reference calls an un-vendored CUDA package for these (import sites cednerf/utils.py:12-18,
cednerf/render.py:6); this module keeps the names, argument order and return conventions its call
sites rely on (cednerf/utils.py:115-125,215,241-299; cednerf/render.py:52-54,81-87,158-169).
"""
from __future__ import annotations

from typing import Callable, NamedTuple, Optional, Tuple

import torch

from . import ops


class RayIntervals(NamedTuple):
    """vals[is_left] / vals[is_right] are the per-sample t_starts / t_ends (cednerf/utils.py:265-266).
    Layout: vals is [S, 2] flattened, is_left marks the even slots, is_right the odd ones."""
    vals: torch.Tensor
    packed_info: Optional[torch.Tensor]
    is_left: torch.Tensor
    is_right: torch.Tensor


class RaySamples(NamedTuple):
    vals: torch.Tensor                   # sample mid-points
    packed_info: torch.Tensor            # [n_rays, 2] (start, count), int64
    ray_indices: torch.Tensor            # int64
    is_valid: torch.Tensor               # bool


def _enlarge_aabb(aabb: torch.Tensor, factor: float) -> torch.Tensor:
    center = (aabb[:3] + aabb[3:]) / 2
    extent = (aabb[3:] - aabb[:3]) / 2
    return torch.cat([center - extent * factor, center + extent * factor])


def ray_aabb_intersect(rays_o, rays_d, aabbs, near_plane: float = -float("inf"), far_plane: float = float("inf"),
                       miss_value: float = float("inf")):
    """cednerf/utils.py:215.  Returns t_mins, t_maxs [n_rays, m] and hits [n_rays, m] (bool)."""
    return ops.ray_aabb_intersect(rays_o.contiguous(), rays_d.contiguous(), aabbs.contiguous(), near_plane, far_plane,
                                  miss_value)


def sort_intersections(t_mins, t_maxs):
    """The event list of cednerf/utils.py:219-225."""
    n_rays, n_grids = t_mins.shape
    if n_grids > 1 and t_mins.is_cuda and n_grids <= 8:
        return ops.sort_intersections(t_mins.contiguous(), t_maxs.contiguous())         # one HIP launch (torch.sort: 3 ms at C4)
    if n_grids > 1:
        t_sorted, t_indices = torch.sort(torch.cat([t_mins, t_maxs], -1), dim=-1, stable=True)
    else:
        t_sorted = torch.cat([t_mins, t_maxs], -1)
        t_indices = torch.arange(0, n_grids * 2, device=t_mins.device, dtype=torch.int64).expand(n_rays, n_grids * 2)
    return t_sorted.contiguous(), t_indices.contiguous()


def march_packed(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle, limit=0,
                 rays_mask=None, t_sorted=None, t_indices=None, hits=None, max_total: Optional[int] = None):
    """Count -> scan -> fill marching into ray-packed arrays.

    Returns (t_starts, t_ends, ray_indices, packed_info[n,2], termination_planes).  When `max_total`
    (a host-known upper bound of the sample count) is given the outputs are views of buffers of
    that size and NO device->host sync happens here beyond the one needed to slice them."""
    n = rays_o.shape[0]
    dev = rays_o.device
    if t_sorted is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    counts = torch.empty((n,), device=dev, dtype=torch.int64)
    term = torch.empty((n,), device=dev, dtype=torch.float32)
    common = (rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle, limit, rays_mask,
              t_sorted, t_indices, hits)
    ops.traverse_grids_raw(*common, 0, counts=counts, termination_planes=term)
    incl = torch.cumsum(counts, 0)
    base = incl - counts
    total = int(incl[-1].item()) if n > 0 else 0
    t_starts = torch.empty((total,), device=dev, dtype=torch.float32)
    t_ends = torch.empty((total,), device=dev, dtype=torch.float32)
    ray_indices = torch.empty((total,), device=dev, dtype=torch.int64)
    if total > 0:
        counts2 = torch.empty_like(counts)
        ops.traverse_grids_raw(*common, 1, base=base, counts=counts2, t_starts=t_starts, t_ends=t_ends,
                               ray_indices=ray_indices, termination_planes=term)
    packed_info = torch.stack([base, counts], -1)
    return t_starts, t_ends, ray_indices, packed_info, term


def traverse_grids(rays_o, rays_d, binaries, aabbs, near_planes=None, far_planes=None, step_size: float = 1e-3,
                   cone_angle: float = 0.0, traverse_steps_limit: Optional[int] = None, over_allocate: bool = False,
                   rays_mask=None, t_sorted=None, t_indices=None, hits=None):
    """nerfacc.traverse_grids with the positional order used at cednerf/utils.py:245-263.
    Returns (RayIntervals, RaySamples, termination_planes)."""
    rays_o = rays_o.contiguous(); rays_d = rays_d.contiguous()
    n = rays_o.shape[0]
    dev = rays_o.device
    if near_planes is None:
        near_planes = torch.zeros((n,), device=dev)
    if far_planes is None:
        far_planes = torch.full((n,), float("inf"), device=dev)
    near_planes = near_planes.contiguous(); far_planes = far_planes.contiguous()
    limit = -1 if traverse_steps_limit is None else int(traverse_steps_limit)
    if t_sorted is None or t_indices is None or hits is None:
        t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, aabbs)
        t_sorted, t_indices = sort_intersections(t_mins, t_maxs)
    if over_allocate:
        if limit <= 0:
            raise ValueError("traverse_steps_limit must be set if over_allocate is True.")
        total = n * limit
        t_starts = torch.zeros((total,), device=dev, dtype=torch.float32)
        t_ends = torch.zeros((total,), device=dev, dtype=torch.float32)
        ray_indices = torch.zeros((total,), device=dev, dtype=torch.int64)
        counts = torch.empty((n,), device=dev, dtype=torch.int64)
        term = torch.empty((n,), device=dev, dtype=torch.float32)
        ops.traverse_grids_raw(rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle, limit,
                               rays_mask, t_sorted.contiguous(), t_indices.contiguous(), hits.contiguous(), 2,
                               counts=counts, t_starts=t_starts, t_ends=t_ends, ray_indices=ray_indices,
                               termination_planes=term)
        starts = torch.arange(n, device=dev, dtype=torch.int64) * limit
        packed_info = torch.stack([starts, counts], -1)
        slot = torch.arange(limit, device=dev, dtype=torch.int64)
        is_valid = (slot[None, :] < counts[:, None]).reshape(-1)
    else:
        t_starts, t_ends, ray_indices, packed_info, term = march_packed(
            rays_o, rays_d, binaries, aabbs, near_planes, far_planes, step_size, cone_angle, limit, rays_mask,
            t_sorted.contiguous(), t_indices.contiguous(), hits.contiguous())
        is_valid = torch.ones_like(ray_indices, dtype=torch.bool)
    vals = torch.stack([t_starts, t_ends], -1).reshape(-1)
    valid2 = torch.stack([is_valid, is_valid], -1)
    left = torch.zeros_like(valid2); left[:, 0] = True
    right = torch.zeros_like(valid2); right[:, 1] = True
    intervals = RayIntervals(vals=vals, packed_info=None, is_left=(left & valid2).reshape(-1),
                             is_right=(right & valid2).reshape(-1))
    samples = RaySamples(vals=(t_starts + t_ends) / 2.0, packed_info=packed_info, ray_indices=ray_indices,
                         is_valid=is_valid)
    return intervals, samples, term


def _packed_info_from(ray_indices: torch.Tensor, n_rays: int) -> torch.Tensor:
    counts = torch.bincount(ray_indices, minlength=n_rays)
    base = torch.cumsum(counts, 0) - counts
    return torch.stack([base, counts], -1).contiguous()


def _resolve_packed(packed_info, ray_indices, n_rays):
    if packed_info is not None:
        return packed_info.contiguous()
    if ray_indices is None or n_rays is None:
        raise ValueError("either packed_info or (ray_indices, n_rays) is required")
    return _packed_info_from(ray_indices, n_rays)


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                                      prefix_trans=None):
    """cednerf/render.py:52-54.  Returns (trans, alphas)."""
    packed = _resolve_packed(packed_info, ray_indices, n_rays)
    _, trans, alphas = ops.render_weights(packed, t_starts.contiguous(), t_ends.contiguous(), sigmas.contiguous(),
                                          None if prefix_trans is None else prefix_trans.contiguous(),
                                          want=(False, True, True))
    return trans, alphas


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                               prefix_trans=None):
    """cednerf/render.py:81-87, cednerf/utils.py:274-281.  Returns (weights, trans, alphas)."""
    packed = _resolve_packed(packed_info, ray_indices, n_rays)
    return ops.render_weights(packed, t_starts.contiguous(), t_ends.contiguous(), sigmas.contiguous(),
                              None if prefix_trans is None else prefix_trans.contiguous())


def render_visibility_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None,
                                   early_stop_eps: float = 1e-4, alpha_thre: float = 0.0):
    packed = _resolve_packed(packed_info, ray_indices, n_rays)
    return ops.visibility_mask(packed, t_starts.contiguous(), t_ends.contiguous(), sigmas.contiguous(),
                               early_stop_eps, alpha_thre)


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None, packed_info=None):
    """cednerf/render.py:158-169.  Returns [n_rays, C] (C = 1 when values is None)."""
    packed = _resolve_packed(packed_info, ray_indices, n_rays)
    c = 1 if values is None else values.shape[-1]
    out = torch.zeros((packed.shape[0], c), device=weights.device, dtype=torch.float32)
    return ops.accumulate_along_rays_(packed, weights.contiguous(), None if values is None else values.contiguous(),
                                      out)


def accumulate_along_rays_(weights, values=None, ray_indices=None, outputs=None, packed_info=None):
    """In-place variant, cednerf/utils.py:282-299."""
    assert outputs is not None
    packed = _resolve_packed(packed_info, ray_indices, outputs.shape[0])
    ops.accumulate_along_rays_(packed, weights.contiguous(), None if values is None else values.contiguous(), outputs)


class OccGridEstimator(torch.nn.Module):
    """Occupancy-grid state + sampling(), as built at train_real.py:185-187 and used at
    cednerf/utils.py:115-125,215-217,250-251.  Grid maintenance
    (update_every_n_steps / mark_invisible_cells) is a "next" row and not part of the hot path."""

    DIM: int = 3

    def __init__(self, roi_aabb, resolution=128, levels: int = 1) -> None:
        super().__init__()
        if isinstance(resolution, int):
            resolution = [resolution] * self.DIM
        resolution = torch.as_tensor(resolution, dtype=torch.int32)
        if not isinstance(roi_aabb, torch.Tensor):
            roi_aabb = torch.tensor(roi_aabb, dtype=torch.float32)
        assert roi_aabb.shape == (6,), f"Invalid shape: {roi_aabb.shape}!"
        aabbs = torch.stack([_enlarge_aabb(roi_aabb, 2 ** i) for i in range(levels)], dim=0)
        self.cells_per_lvl = int(resolution.prod().item())
        self.levels = levels
        self.register_buffer("resolution", resolution)
        self.register_buffer("aabbs", aabbs)
        self.register_buffer("occs", torch.zeros(self.levels * self.cells_per_lvl))
        self.one_pass_march = True       # sampling / render_image may march in one pass with a capacity from the last call
        self.register_buffer("binaries", torch.zeros([levels] + resolution.tolist(), dtype=torch.bool))
        res = resolution.tolist()
        grid_coords = torch.stack(torch.meshgrid([torch.arange(r) for r in res], indexing="ij"), dim=-1)
        self.register_buffer("grid_coords", grid_coords.reshape(self.cells_per_lvl, self.DIM), persistent=False)
        self.register_buffer("grid_indices", torch.arange(self.cells_per_lvl), persistent=False)

    def occupancy_accel(self) -> torch.Tensor:
        """The brick distance field of `binaries` for the frame renderer (ops.build_occupancy_accel), rebuilt when the
        grid has changed (set_binaries, _update) and kept otherwise: a video renders hundreds of frames per grid."""
        b = self.binaries
        # the generation counter is bumped by every writer of `binaries` in this class (set_binaries, _update,
        # load_state_dict); pointer / version catch in-place writes from outside (binaries.copy_(...)).  A writer
        # that goes through `.data` must call invalidate_accel() itself.
        key = (self.__dict__.get("_grid_generation", 0), b.data_ptr(), b._version, str(b.device))
        if getattr(self, "_accel_key", None) != key:
            with torch.cuda.device(b.device):
                self._accel = ops.build_occupancy_accel(b.contiguous())
            self._accel_key = key
        return self._accel

    def invalidate_accel(self) -> None:
        """Forget the cached distance fields: the next render rebuilds them from `binaries`."""
        self.__dict__["_grid_generation"] = self.__dict__.get("_grid_generation", 0) + 1
        self._accel_key = None

    def set_binaries(self, binaries: torch.Tensor, occs: Optional[torch.Tensor] = None) -> None:
        """Load a precomputed grid (e.g. from a checkpoint's 'occupancy_grid' state)."""
        assert binaries.shape == self.binaries.shape, (binaries.shape, self.binaries.shape)
        self.binaries.copy_(binaries.to(self.binaries.device, torch.bool))
        self.occs.copy_(self.binaries.reshape(-1).float() if occs is None else occs.to(self.occs.device))
        self.invalidate_accel()

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_accel()

    # ---- grid maintenance: nerfacc OccGridEstimator._update & friends, as driven at
    # train_real.py:202-211,324-336.  Index sampling is torch (it is in nerfacc too); the per-sample
    # position and EMA steps are HIP launches, the density query is the fused field kernel. ----
    @torch.no_grad()
    def _get_all_cells(self):
        """Per level: every cell not marked invisible (occs >= 0)."""
        out = []
        for lvl in range(self.levels):
            cell_ids = lvl * self.cells_per_lvl + self.grid_indices
            out.append(self.grid_indices[self.occs[cell_ids] >= 0.0])
        return out

    @torch.no_grad()
    def _sample_uniform_and_occupied_cells(self, n: int):
        """Per level: n uniformly drawn visible cells plus (at most n of) the occupied ones."""
        out = []
        dev = self.occs.device
        for lvl in range(self.levels):
            uniform = torch.randint(self.cells_per_lvl, (n,), device=dev)
            uniform = uniform[self.occs[lvl * self.cells_per_lvl + uniform] >= 0.0]
            occupied = torch.nonzero(self.binaries[lvl].flatten())[:, 0]
            if n < len(occupied):
                occupied = occupied[torch.randint(len(occupied), (n,), device=dev)]
            out.append(torch.cat([uniform, occupied], dim=0))
        return out

    @torch.no_grad()
    def _update(self, step: int, occ_eval_fn: Callable, occ_thre: float = 0.01, ema_decay: float = 0.95,
                warmup_steps: int = 256, _lvl_indices=None, _noise=None) -> None:
        """One EMA update of `occs` and re-thresholding of `binaries`.  `occ_eval_fn(x)` returns
        density * render_step_size at world positions x [n,3] (train_real.py:324-328).  `_lvl_indices` /
        `_noise` inject the random draws (tests)."""
        if _lvl_indices is None:
            if step < warmup_steps:
                _lvl_indices = self._get_all_cells()
            else:
                _lvl_indices = self._sample_uniform_and_occupied_cells(self.cells_per_lvl // 4)
        res = int(self.resolution[0].item())
        assert bool((self.resolution == res).all()), "cubic grids only"
        aabbs = self.aabbs.detach().cpu().numpy()
        for lvl, indices in enumerate(_lvl_indices):
            indices = indices.contiguous()
            if indices.numel() == 0:
                continue
            noise = torch.rand((indices.shape[0], 3), device=indices.device) if _noise is None else _noise[lvl]
            x = ops.occ_cell_points(indices, noise.contiguous(), res, aabbs[lvl])
            occ = occ_eval_fn(x).reshape(-1).float().contiguous()
            cell_ids = (lvl * self.cells_per_lvl + indices).contiguous()
            ops.occ_ema_update_(self.occs, cell_ids, occ, 1.0, ema_decay)     # occ already holds density * step
        thre = torch.clamp(self.occs[self.occs >= 0].mean(), max=occ_thre)
        # in place (nerfacc rebinds the attribute): the buffer keeps its address, its version counter advances, and
        # the cached distance fields of the old grid are dropped explicitly
        self.binaries.copy_((self.occs > thre).view(self.binaries.shape))
        self.invalidate_accel()

    @torch.no_grad()
    def update_every_n_steps(self, step: int, occ_eval_fn: Callable, occ_thre: float = 1e-2, ema_decay: float = 0.95,
                             warmup_steps: int = 256, n: int = 16) -> None:
        """train_real.py:332-336."""
        if not self.training:
            raise RuntimeError("You should only call this function only during training. "
                               "Please call _update() directly if you want to update the field during inference.")
        if step % n == 0 and self.training:
            self._update(step=step, occ_eval_fn=occ_eval_fn, occ_thre=occ_thre, ema_decay=ema_decay,
                         warmup_steps=warmup_steps)

    @torch.no_grad()
    def mark_invisible_cells(self, K: torch.Tensor, c2w: torch.Tensor, width: int, height: int,
                             near_plane: float = 0.0, chunk: int = 32 ** 3) -> None:
        """Cells no training camera sees get occs = -1 and are never sampled (train_real.py:205-211).
        A one-off projection test of cell corners (plain torch, as in nerfacc)."""
        assert K.dim() == 3 and K.shape[1:] == (3, 3)
        assert c2w.dim() == 3 and (c2w.shape[1:] == (3, 4) or c2w.shape[1:] == (4, 4))
        assert K.shape[0] == c2w.shape[0] or K.shape[0] == 1
        n_cams = c2w.shape[0]
        w2c_R = c2w[:, :3, :3].transpose(2, 1)
        w2c_T = -w2c_R @ c2w[:, :3, 3:]
        for lvl, indices in enumerate(self._get_all_cells()):
            coords = self.grid_coords[indices]
            for i in range(0, len(indices), chunk):
                x = coords[i:i + chunk] / (self.resolution - 1)
                idx = indices[i:i + chunk]
                xyz_w = (self.aabbs[lvl, :3] + x * (self.aabbs[lvl, 3:] - self.aabbs[lvl, :3])).T
                uvd = K @ (w2c_R @ xyz_w + w2c_T)
                uv = uvd[:, :2] / uvd[:, 2:]
                in_image = (uvd[:, 2] >= 0) & (uv[:, 0] >= 0) & (uv[:, 0] < width) & (uv[:, 1] >= 0) & (uv[:, 1] < height)
                covered = (uvd[:, 2] >= near_plane) & in_image
                too_near = ((uvd[:, 2] < near_plane) & in_image).any(0)
                valid = (covered.sum(0) / n_cams > 0) & (~too_near)
                self.occs[lvl * self.cells_per_lvl + idx] = torch.where(valid, 0.0, -1.0)

    @torch.no_grad()
    def march(self, rays_o, rays_d, near_plane: float = 0.0, far_plane: float = 1e10, t_min=None, t_max=None,
              render_step_size: float = 1e-3, stratified: bool = False, cone_angle: float = 0.0,
              want_ray_indices: bool = True, fast: Optional[bool] = None, near_planes: Optional[torch.Tensor] = None):
        """The marching half of `sampling`: every sample inside occupied cells, all rays to the far plane.
        Returns (t_starts, t_ends, ray_indices, packed_info).  fast: None = the accelerated walk when it applies,
        False = ced_traverse_grids.  near_planes: the per-ray near planes already drawn (`_near_planes`)."""
        rays_o = rays_o.contiguous(); rays_d = rays_d.contiguous()
        if near_planes is None:
            near_planes = self._near_planes(rays_o, near_plane, t_min, stratified, render_step_size)
        far_planes = torch.full_like(rays_o[..., 0], fill_value=far_plane)
        if t_max is not None:
            far_planes = torch.clamp(far_planes, max=t_max)
        if fast is None:
            fast = t_max is None and rays_o.is_cuda
        if fast:        # the accelerated walk of the frame renderer (same samples, ced_march_all)
            ev = (None, None, None)
            if self.binaries.shape[0] > 1:
                t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, self.aabbs)
                ev = sort_intersections(t_mins, t_maxs) + (hits.contiguous(),)
            return ops.march_all(rays_o, rays_d, self.binaries, self.aabbs, self.occupancy_accel(), near_planes.contiguous(),
                                 far_plane, render_step_size, cone_angle, want_ray_indices=want_ray_indices,
                                 t_sorted=ev[0], t_indices=ev[1], hits=ev[2])
        t_starts, t_ends, ray_indices, packed_info, _ = march_packed(
            rays_o, rays_d, self.binaries, self.aabbs, near_planes.contiguous(), far_planes.contiguous(),
            render_step_size, cone_angle)
        return t_starts, t_ends, ray_indices, packed_info

    @staticmethod
    def _near_planes(rays_o, near_plane, t_min, stratified, render_step_size):
        """Per-ray near planes of nerfacc's sampling: the constant, clamped by t_min, jittered by one step if stratified."""
        near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        if t_min is not None:
            near_planes = torch.clamp(near_planes, min=t_min)
        if stratified:
            near_planes = near_planes + torch.rand_like(near_planes) * render_step_size
        return near_planes

    @torch.no_grad()
    def march_onepass(self, rays_o, rays_d, near_plane: float, far_plane: float, render_step_size: float,
                      cone_angle: float, capacity: int, near_planes: Optional[torch.Tensor] = None):
        """`march` in one pass into arrays of `capacity` samples (ops.march_all_onepass): for a caller that can bound the
        total, e.g. by the previous frame's.  Returns (t_starts, t_ends, packed_info, total [device int64])."""
        rays_o = rays_o.contiguous(); rays_d = rays_d.contiguous()
        if near_planes is None:
            near_planes = torch.full_like(rays_o[..., 0], fill_value=near_plane)
        near_planes = near_planes.contiguous()
        ev = (None, None, None)
        if self.binaries.shape[0] > 1:
            t_mins, t_maxs, hits = ray_aabb_intersect(rays_o, rays_d, self.aabbs)
            ev = sort_intersections(t_mins, t_maxs) + (hits.contiguous(),)
        return ops.march_all_onepass(rays_o, rays_d, self.binaries, self.aabbs, self.occupancy_accel(), near_planes,
                                     far_plane, render_step_size, cone_angle, capacity, t_sorted=ev[0], t_indices=ev[1],
                                     hits=ev[2])

    @torch.no_grad()
    def sampling(self, rays_o, rays_d, sigma_fn: Optional[Callable] = None, alpha_fn: Optional[Callable] = None,
                 near_plane: float = 0.0, far_plane: float = 1e10, t_min=None, t_max=None,
                 render_step_size: float = 1e-3, early_stop_eps: float = 1e-4, alpha_thre: float = 0.0,
                 stratified: bool = False, cone_angle: float = 0.0,
                 sigma_field=None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """nerfacc OccGridEstimator.sampling (call sites cednerf/utils.py:115-125, train_real.py:339-350).
        sigma_field = (field, timestamps, per_ray): says that `sigma_fn` IS the density of that HIP field
        (`field.query_rays(..., want_rgb=False)`); the visibility filter then runs on ced_render_image's sampling-only
        mode -- density evaluated front to back, rays stopped at the threshold -- with the survivors of the filter over
        every marched sample, bit for bit (tests/test_gpu_parity.py)."""
        native = (sigma_field is not None and rays_o.is_cuda and (alpha_thre > 0.0 or early_stop_eps > 0.0)
                  and early_stop_eps > 0.0)
        if native:
            fld, ts, per_ray = sigma_field
            thre = float(alpha_thre)
            if thre > 0.0:
                thre = min(thre, self.occs.mean().item())
            rays_o = rays_o.contiguous(); rays_d = rays_d.contiguous()
            near_planes = self._near_planes(rays_o, near_plane, t_min, stratified, render_step_size)
            tq = ts.reshape(-1).float().contiguous()
            # one-pass march into arrays sized by the last batch of this size (batches of a training run march about the
            # same number of samples); a batch that does not fit is redone with the exact two-pass march
            hints = self.__dict__.setdefault("_march_totals", {})
            n = rays_o.shape[0]
            hint = hints.get(n)
            if hint is not None and t_max is None and self.one_pass_march:
                cap = int(hint * 1.25) + 65536
                t0, t1, packed, total_dev = self.march_onepass(rays_o, rays_d, near_plane, far_plane, render_step_size,
                                                               cone_angle, cap, near_planes=near_planes)
                out = ops.sampling_native(fld._descriptor(), rays_o, rays_d, packed, t0, t1, early_stop_eps, thre, tq, per_ray)
                total = int(total_dev.item())
                hints[n] = total
                if total <= cap:
                    return out
            t_starts, t_ends, _, packed_info = self.march(rays_o, rays_d, near_plane, far_plane, t_min, t_max, render_step_size,
                                                          stratified, cone_angle, want_ray_indices=False,
                                                          near_planes=near_planes)
            hints[n] = int(t_starts.shape[0])
            return ops.sampling_native(fld._descriptor(), rays_o, rays_d, packed_info, t_starts, t_ends, early_stop_eps, thre,
                                       tq, per_ray)
        t_starts, t_ends, ray_indices, packed_info = self.march(rays_o, rays_d, near_plane, far_plane, t_min, t_max,
                                                                render_step_size, stratified, cone_angle)
        if (alpha_thre > 0.0 or early_stop_eps > 0.0) and (sigma_fn is not None or alpha_fn is not None):
            alpha_thre = min(alpha_thre, self.occs.mean().item())
            if alpha_fn is not None:
                raise NotImplementedError("alpha_fn is not used by Ced-NeRF; pass sigma_fn")
            if t_starts.shape[0] != 0:
                sigmas = sigma_fn(t_starts, t_ends, ray_indices)
            else:
                sigmas = torch.empty((0,), device=t_starts.device)
            assert sigmas.shape == t_starts.shape, "sigmas must have shape of (N,)! Got {}".format(sigmas.shape)
            masks = ops.visibility_mask(packed_info, t_starts, t_ends, sigmas.contiguous(), early_stop_eps, alpha_thre)
            ray_indices, t_starts, t_ends = ray_indices[masks], t_starts[masks], t_ends[masks]
        return ray_indices, t_starts, t_ends
