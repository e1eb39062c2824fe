"""The video step: render a camera path frame by frame and hand back the 8-bit frames
the reference writes to `rgb_render.mp4` / `depth_render.mp4` (train_real.py:531-558; the viewer's `render_frame`,
gui.py:205-237, is the same call for one frame).

The reference renders the frames one after another and converts each on the host.  Here the frames of the path are a
stream through `dist.PipelinedRenderer` (several frames in flight on one GPU, each a complete `render_image_test`
call, so every frame's pixels are those of rendering it alone), rays are generated on the device from the camera
(`cameras.pinhole_rays`) or supplied, and the float images are turned into uint8 on the device
(`ops.frame_to_rgb8`, `ops.depth_to_u8`).  Encoding to mp4 (imageio) and cv2's TURBO colour map for the depth video
are host-side table/codec work and are left to the caller: `depth` frames are the normalised 8-bit depths that
`cv2.applyColorMap` takes as input.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .dist import PipelinedRenderer, ShardedRenderer
from .utils import Rays


def frame_to_uint8(rgb: torch.Tensor, depth: torch.Tensor, flip_w: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(rgb [H,W,3] f32, depth [H,W,1] f32) -> (uint8 [H,W,3], uint8 [H,W]) as train_real.py:556-557 builds them."""
    return ops.frame_to_rgb8(rgb.contiguous(), flip_w), ops.depth_to_u8(depth.contiguous(), flip_w)


@torch.no_grad()
def render_video(radiance_field, estimator, rays_of_frame: Callable[[int], Rays], timestamps_of_frame: Callable[[int], torch.Tensor],
                 n_frames: int, max_samples: int = 1024, render_kwargs: Optional[Dict] = None, frames_in_flight: int = 3,
                 flip_w: bool = True, to_host: bool = False, keep_float: bool = False, frames_per_call: int = 1) -> List[Dict]:
    """Render frames 0..n_frames-1 of a path.

    rays_of_frame(i) -> Rays with [H,W,3] device tensors (e.g. `cameras.pinhole_rays(K, c2w_i, W, H)`; it is called on
    the rendering thread of the frame's lane with that lane's stream current, so device ray generation overlaps the
    other frames); timestamps_of_frame(i) -> the frame's [1,1] time.  Returns one dict per frame: `rgb` uint8 [H,W,3],
    `depth` uint8 [H,W] (torch tensors on the device, or numpy arrays with to_host=True), `n_samples`; with
    keep_float=True also the float images `rgb_f32`, `opacity_f32`, `depth_f32`.
    frames_per_call > 1 (up to 8): that many consecutive frames go through one native call (ced_render_frames_test:
    shared launches, each frame on its own schedule -- same pixels, larger launches); `n_samples` is then reported
    per call on its first frame's entry and the others carry None."""
    if n_frames <= 0:
        return []
    device = radiance_field.aabb.device if hasattr(radiance_field, "aabb") else torch.device("cuda")
    per_call = max(1, min(8, int(frames_per_call)))
    n_calls = (n_frames + per_call - 1) // per_call
    n_lanes = max(1, min(int(frames_in_flight), n_calls))
    lanes = [ShardedRenderer(radiance_field, estimator, 1, 0, device, max_samples=max_samples,
                             render_kwargs=render_kwargs, tile_order=True, units=per_call) for _ in range(n_lanes)]
    pipe = PipelinedRenderer(lanes)
    n_steps = (n_calls + n_lanes - 1) // n_lanes
    # frames of the call (lane, step); the tail of the path repeats the last frame
    frames_of = lambda lane, step: [min((step * n_lanes + lane) * per_call + k, n_frames - 1) for k in range(per_call)]

    def before_frame(lane, step):
        rs = [rays_of_frame(i) for i in frames_of(lane, step)]
        lanes[lane].set_rays(torch.stack([r.origins for r in rs]), torch.stack([r.viewdirs for r in rs]))

    def times(lane, step):
        return torch.cat([timestamps_of_frame(i).reshape(-1)[:1] for i in frames_of(lane, step)])

    steps = pipe.render_steps(times, n_steps, before_frame)
    frames = []
    for step, row in enumerate(steps):
        for lane, out in enumerate(row):
            for k in range(per_call):
                i = (step * n_lanes + lane) * per_call + k
                if i >= n_frames:
                    break
                rgb, depth = out["rgb"][k], out["depth"][k]
                rgb8, d8 = frame_to_uint8(rgb, depth, flip_w)
                f = {"rgb": rgb8.cpu().numpy() if to_host else rgb8, "depth": d8.cpu().numpy() if to_host else d8,
                     "n_samples": int(out["total_samples"]) if (per_call == 1 or k == 0) else None}
                if keep_float:
                    f.update(rgb_f32=rgb, opacity_f32=out["opacity"][k], depth_f32=depth)
                frames.append(f)
    return frames
