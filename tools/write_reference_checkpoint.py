"""Writes a `model.pth` in the layout a REFERENCE run would leave behind (train_real.py:433-441), under the
tiny-cuda-nn layout hypothesis of ced_nerf_amd/checkpoint.py (TCNN_LAYOUT), from known weights: the synthetic field and
occupancy grid of ced_nerf_amd.synthetic.  Used by the round-trip tests and as a template for converting real
checkpoints the other way.  No GPU needed.

    python tools/write_reference_checkpoint.py out.pth [--scene dnerf] [--width 64 --height 48] [--head-bias]
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--scene", default="dnerf")
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--height", type=int, default=48)
    ap.add_argument("--log2-hashmap-size", type=int, default=15)
    ap.add_argument("--head-bias", action="store_true",
                    help="carry part of mlp_head's first layer as a bias in its ones-padded input column")
    args = ap.parse_args()
    from ced_nerf_amd import synthetic as S
    from ced_nerf_amd.checkpoint import reference_state_from_field
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    sc = S.make_scene(args.scene, args.width, args.height, "trained", log2_hashmap_size=args.log2_hashmap_size)
    cfg = sc["cfg"]
    field = DNGPradianceField.from_params(sc["params"], "cpu")
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"])
    est.set_binaries(torch.from_numpy(sc["binaries"]))
    bias = np.random.default_rng(5).normal(scale=0.05, size=64).astype(np.float32) if args.head_bias else None
    torch.save({"radiance_field": reference_state_from_field(field, head_bias=bias), "occupancy_grid": est.state_dict()}, args.out)
    print(f"wrote {args.out}: " + ", ".join(f"{k} {tuple(v.shape)}" for k, v in reference_state_from_field(field, bias).items()))


if __name__ == "__main__":
    main()
