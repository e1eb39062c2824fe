#!/usr/bin/env python3
"""Falsifiability gate for `checkpoint.load_reference_checkpoint` (needs a GPU: the field has no CPU path).

The tiny-cuda-nn parameter layout assumed by the loader is a HYPOTHESIS (ced_nerf_amd/checkpoint.py): a wrong guess
renders *something* without any error.  A `model.pth` of the reference holds two independent halves -- the field and the
occupancy grid the trainer thresholded from that field's own density (train_real.py:324-336,433-441) -- so a correctly
loaded file must be dense where its grid says occupied and empty where it says free.  This tool loads both halves and
reports how well they agree; under a wrong layout the numbers are at chance.

    python tools/verify_checkpoint.py model.pth --scene dnerf [-te] [-ta] [-df] [--log2-hashmap-size 21 --max-res 1024]
                                       [--moving-step 0.0009765625] [--render-step-size 5e-3] [--json]

exit status: 0 consistent, 1 inconsistent (the layout hypothesis or the flags are wrong), 2 unclear
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint")
    ap.add_argument("--scene", default="dnerf", choices=["dnerf", "hypernerf", "dynerf"],
                    help="constants of train_real.py:97-175 (aabb, grid levels, hash resolution, step size)")
    ap.add_argument("-te", "--use-time-embedding", action="store_true")
    ap.add_argument("-ta", "--use-time-attenuation", action="store_true")
    ap.add_argument("-df", "--use-div-offsets", action="store_true")
    ap.add_argument("--log2-hashmap-size", type=int, default=None)
    ap.add_argument("--max-res", type=int, default=None)
    ap.add_argument("--moving-step", type=float, default=None)
    ap.add_argument("--render-step-size", type=float, default=None)
    ap.add_argument("--occ-thre", type=float, default=1e-2)
    ap.add_argument("--cells", type=int, default=20000)
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    from ced_nerf_amd import checkpoint as CK, synthetic as S
    from ced_nerf_amd.model import DNGPradianceField
    from ced_nerf_amd.nerfacc_api import OccGridEstimator
    assert torch.cuda.is_available(), "the field is evaluated by the HIP kernels: run this on the GPU box"
    cfg = S.CONFIGS[args.scene]
    dev = "cuda:0"
    max_res = args.max_res or cfg["hash_max_res"]
    field_aabb = S.enlarge_aabb(cfg["aabb"], 2 ** (cfg["grid_levels"] - 1))        # estimator.aabbs[-1], train_real.py:254
    field = DNGPradianceField(aabb=field_aabb, moving_step=args.moving_step or cfg["moving_step"],
                              dst_resolution=max_res, log2_hashmap_size=args.log2_hashmap_size or 21,
                              use_time_embedding=args.use_time_embedding, use_time_attenuation=args.use_time_attenuation,
                              use_div_offsets=args.use_div_offsets).to(dev).eval()
    est = OccGridEstimator(cfg["aabb"], cfg["grid_resolution"], cfg["grid_levels"]).to(dev)
    CK.load_reference_checkpoint(args.checkpoint, field, est, assume_tcnn_layout=CK.TCNN_LAYOUT, map_location="cpu")
    rep = CK.checkpoint_consistency(field, est, render_step_size=args.render_step_size or cfg["render_step_size"],
                                    occ_thre=args.occ_thre, n_cells=args.cells)
    if args.json:
        print(json.dumps(rep))
    else:
        for l in rep["levels"]:
            print("level {level}: ".format(**l) + ", ".join(f"{k} {v:.3f}" if isinstance(v, float) else f"{k} {v}" for k, v in l.items() if k != "level"))
        print("sigma:", rep.get("sigma")); print("rgb:  ", rep.get("rgb"))
        print(f"VERDICT: {rep['verdict']}  (separation {rep.get('separation_min', float('nan')):.3f}, auc {rep.get('auc_min', float('nan')):.3f}; "
              "consistent needs >= 0.5 and >= 0.8, chance is 0 and 0.5)")
    sys.exit({"consistent": 0, "inconsistent": 1}.get(rep["verdict"], 2))


if __name__ == "__main__":
    main()
