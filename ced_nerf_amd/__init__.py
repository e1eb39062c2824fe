"""ced_nerf_amd: MI355X-native (gfx950) rendering hot path behind the Ced-NeRF Python API.

Module names follow the reference package (`cednerf.utils`, `cednerf.render`, `cednerf.model`,
`cednerf.encoder`) so that `import ced_nerf_amd as cednerf` is a drop-in for the rendering path.
"""
__version__ = "0.1.0"

from . import encoder, hashgrid, synthetic  # noqa: F401  (no GPU / native code needed)


def __getattr__(name):
    # torch-facing modules are imported lazily so that `import ced_nerf_amd` stays cheap
    if name in ("ops", "nerfacc_api", "model", "render", "utils", "dist", "_lib", "cameras", "train", "profiling", "video"):
        import importlib
        return importlib.import_module("." + name, __name__)
    raise AttributeError(name)
