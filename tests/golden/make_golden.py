"""Generates tests/golden/*.npz by IMPORTING the reference (run in the build container only:
`python tests/golden/make_golden.py`; /root/reference does not exist on the GPU box).

Only data leaves this script: inputs and the reference's outputs.  Importable pieces of the
reference: cednerf/encoder.py (the two time encoders) and datasets/utils.py
(Rays, namedtuple_map).  Everything else on the hot path needs nerfacc / tiny-cuda-nn / taichi,
which are not installed and not installable offline.
"""
import importlib.util
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    enc = _load("ref_cednerf_encoder", os.path.join(REF, "cednerf", "encoder.py"))
    plain = enc.SinusoidalEncoder(1, 0, 4, True)
    damped = enc.SinusoidalEncoderWithExp(1, 0, 4, True)
    t = torch.linspace(0.0, 1.0, 33, dtype=torch.float32)[:, None]
    extra = torch.tensor([[0.3], [0.123456], [0.999], [1e-3]], dtype=torch.float32)
    t = torch.cat([t, extra], 0)
    out_plain = plain(t)
    moves = [0.0, 1e-4, 1e-2, 0.3, 1.0]
    tt = t.repeat(len(moves), 1)
    mv = torch.tensor(moves, dtype=torch.float32).repeat_interleave(t.shape[0])[:, None]
    out_damped = damped(tt, mv)
    np.savez(os.path.join(HERE, "time_encoders.npz"),
             t=t.numpy(), plain=out_plain.numpy(), t_damped=tt.numpy(), move=mv.numpy(), damped=out_damped.numpy(),
             plain_latent_dim=np.int64(plain.latent_dim), damped_latent_dim=np.int64(damped.latent_dim))
    du = _load("ref_datasets_utils", os.path.join(REF, "datasets", "utils.py"))
    rays = du.Rays(origins=torch.arange(24.0).reshape(2, 4, 3), viewdirs=torch.ones(2, 4, 3))
    flat = du.namedtuple_map(lambda r: r.reshape([8] + list(r.shape[2:])), rays)
    np.savez(os.path.join(HERE, "rays_namedtuple.npz"), fields=np.array(du.Rays._fields),
             flat_origins=flat.origins.numpy(), flat_viewdirs=flat.viewdirs.numpy())
    # HyperNeRF camera rays (datasets/hyper_cam.py): one undistorted and one distorted camera
    hc = _load("ref_hyper_cam", os.path.join(REF, "datasets", "hyper_cam.py"))
    rng = np.random.default_rng(0)
    cams = {}
    for tag, rad, tan, skew, aspect in (("plain", None, None, 0.0, 1.0),
                                        ("distorted", [0.05, -0.02, 0.004], [0.0015, -0.002], 0.3, 1.02)):
        a = rng.normal(size=(3, 3)); q, _ = np.linalg.qr(a)
        if np.linalg.det(q) < 0:
            q[:, 0] = -q[:, 0]
        cam = hc.Camera(orientation=q.astype(np.float32), position=np.array([0.3, -0.2, 1.1], np.float32),
                        focal_length=55.0, principal_point=np.array([23.5, 17.25], np.float32),
                        image_size=np.array([48, 36]), skew=skew, pixel_aspect_ratio=aspect,
                        radial_distortion=rad, tangential_distortion=tan)
        rays = cam.pixels_to_rays(cam.get_pixel_centers())
        cams[tag + "_rays"] = rays.astype(np.float32)
        for k, v in cam.get_parameters().items():
            cams[tag + "_" + k] = np.asarray(v)
    np.savez(os.path.join(HERE, "hypercam_rays.npz"), **cams)
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))


if __name__ == "__main__":
    sys.exit(main())
