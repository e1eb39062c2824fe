"""Loader of the C-ABI library ``libcednerf_hip.so`` (declared in ``include/cednerf_hip.h``).

There is deliberately no CPU or PyTorch fallback: if the library is missing or fails to load,
``lib()`` raises and every op of this package fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import List, Optional

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("CED_NERF_LIB", os.path.join(_PKG, "libcednerf_hip.so"))
SOURCES = ["runtime.hip", "march.hip", "composite.hip", "field.hip", "field_half.hip", "field_mixed.hip", "frame.hip", "occgrid.hip",
           "raygen.hip", "wgrad.hip", "pixels.hip", "accel.hip", "linear.hip", "mlp.hip", "train_glue.hip"]
MLP_F32, MLP_F16X2, MLP_F16, MLP_F32_HEAD16X2 = 0, 1, 2, 3          # ced_field_desc.mlp_precision
# "f32+h16x2": sigma chain exact fp32 (counts / opacity / depth bit-identical to "f32"), colour head on split-fp16 MFMAs
MLP_PRECISIONS = {"f32": MLP_F32, "f16x2": MLP_F16X2, "f16": MLP_F16, "f32+h16x2": MLP_F32_HEAD16X2}
# -fno-slp-vectorize: the SLP vectoriser forms packed-fp32 instructions with an op_sel swizzle (v_pk_mul_f32 ...
# op_sel:[0,1]); on gfx950 that form reads its swizzled operand as zero while another wave of the SIMD runs
# v_mfma_f32_16x16x32_f16 (tools/probes/pk_opsel_mfma.hip).  tools/isa_lint.py checks the built library.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17"]
# Per-source additions.  field_half.hip: the AMDGPU scheduler's "max-memory-clause" strategy instead of its default
# (round 4, alternating runs on one box: +1.0 % on the f16x2 kernel standalone, +0.4 % in bench.py, f16 unchanged, the fp32 /
# mixed kernels slightly slower with it; same results bit for bit: scheduling only -- profiles/r04_ab_sched_strategy.txt).  An option of the AMDGPU
# backend itself: the host half of the compilation accepts and ignores it.
PER_SOURCE_FLAGS = {"field_half.hip": ["-mllvm", "-amdgpu-sched-strategy=max-memory-clause"]}
MAX_LEVELS = 16


class HashDesc(C.Structure):
    """ced_hash_desc"""
    _fields_ = [
        ("n_levels", C.c_int32), ("table_dtype", C.c_int32), ("temporal", C.c_int32), ("reserved", C.c_int32),
        ("scale", C.c_float * MAX_LEVELS), ("res", C.c_uint32 * MAX_LEVELS), ("offset", C.c_uint32 * MAX_LEVELS),
        ("size", C.c_uint32 * MAX_LEVELS), ("hashed", C.c_uint32 * MAX_LEVELS),
        ("table", C.c_void_p), ("total_entries", C.c_uint64),
    ]


class FieldDesc(C.Structure):
    """ced_field_desc"""
    _fields_ = [
        ("aabb", C.c_float * 6), ("moving_step", C.c_float), ("use_div_offsets", C.c_int32),
        ("time_mode", C.c_int32), ("mlp_precision", C.c_int32), ("max_workgroups", C.c_int32),
        ("packed_weights", C.c_void_p), ("packed_floats", C.c_uint64),
        ("hash", HashDesc),
    ]


class FrameTrace(C.Structure):
    """ced_frame_trace"""
    _fields_ = [
        ("capacity", C.c_int32), ("n_iters", C.c_int32),
        ("field_begin", C.POINTER(C.c_void_p)), ("field_end", C.POINTER(C.c_void_p)),
        ("iter_alive", C.POINTER(C.c_int64)), ("iter_n_samples", C.POINTER(C.c_int64)),
        ("iter_samples", C.POINTER(C.c_int64)), ("field_stamps", C.c_void_p),
    ]


# ced_exchange_fn: int (*)(void *user, int64_t *counts, int32_t n_counts, int32_t iteration, void *stream)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p)


class ShardExchange(C.Structure):
    """ced_shard_exchange"""
    _fields_ = [
        ("global_rays_per_frame", C.c_int64), ("local_rays", C.c_void_p), ("counts", C.c_void_p),
        ("reduce", EXCHANGE_FN), ("user", C.c_void_p),
    ]


_vp, _i64, _i32, _f = C.c_void_p, C.c_int64, C.c_int32, C.c_float

# name -> (restype, argtypes); one entry per function declared in include/cednerf_hip.h
PROTOTYPES = {
    "ced_version": (C.c_int, []),
    "ced_last_error_string": (C.c_char_p, []),
    "ced_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "ced_packed_weight_floats": (_i64, [C.c_int, C.c_int]),
    "ced_pack_field_weights": (C.c_int, [C.c_int, C.c_int] + [_vp] * 10),
    "ced_packed_weight_words": (_i64, [C.c_int, C.c_int, C.c_int]),
    "ced_pack_field_weights_half": (C.c_int, [C.c_int, C.c_int, C.c_int] + [_vp] * 10),
    "ced_pack_field_weights_mixed": (C.c_int, [C.c_int, C.c_int] + [_vp] * 10),
    "ced_ray_aabb_intersect": (C.c_int, [_i64, _vp, _vp, _i32, _vp, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "ced_sort_intersections": (C.c_int, [_i64, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ced_traverse_grids": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _f, _f, _i32, _vp, _vp, _vp, _vp,
                                     _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ced_host_skip_march": (_f, [_f, _f, _f, _f]),
    "ced_occupancy_accel_bytes": (_i64, [_i32, _i32]),
    "ced_build_occupancy_accel": (C.c_int, [_vp, _i32, _i32, _vp, _i64, _vp]),
    "ced_host_build_occupancy_accel": (C.c_int, [_vp, _i32, _i32, _vp]),
    "ced_host_count_steps": (_i32, [C.POINTER(C.c_float), _f, _f, _i32, C.POINTER(C.c_float)]),
    "ced_host_march_frame": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f, _f, _f, _i32, _vp, _vp, _vp, _vp, _i32, _i32,
                                       _i32, _vp, _vp, _vp, _vp]),
    "ced_hash_encode": (C.c_int, [C.POINTER(HashDesc), _i64, _vp, _vp, _vp, _vp]),
    "ced_hash_encode_backward": (C.c_int, [C.POINTER(HashDesc), _i64, _vp, _vp, _vp, _vp, _i32, _vp]),
    "ced_hash_encode_backward_temporal": (C.c_int, [C.POINTER(HashDesc), _i64, _vp, _vp, _vp, _vp, _vp]),
    "ced_field_forward": (C.c_int, [C.POINTER(FieldDesc), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ced_field_forward_rays": (C.c_int, [C.POINTER(FieldDesc), _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                         _vp, _vp, _vp]),
    "ced_render_weights": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ced_accumulate_along_rays": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _vp, _vp]),
    "ced_reduce_along_rays": (C.c_int, [_i64, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _vp, _vp, _vp]),
    "ced_visibility_mask": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp]),
    "ced_composite_prefix": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ced_composite_backward": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ced_frame_to_rgb8": (C.c_int, [_i32, _i32, _vp, _i32, _vp, _vp]),
    "ced_depth_to_u8": (C.c_int, [_i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "ced_scatter_pixels": (C.c_int, [_i64, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp]),
    "ced_linear": (C.c_int, [_i64, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "ced_weight_grad_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "ced_weight_grad": (C.c_int, [_i64, _vp, _i32, _vp, _i32, _vp, _vp, _i64, _vp]),
    "ced_composite_step": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _i32, _vp, _vp, _vp]),
    "ced_composite_test": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp]),
    "ced_finalize_pixels": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp]),
    "ced_occ_cell_points": (C.c_int, [_i64, _vp, _vp, _i32, C.POINTER(C.c_float), _vp, _vp]),
    "ced_occ_ema_update": (C.c_int, [_i64, _vp, _vp, _f, _f, _vp, _vp]),
    "ced_generate_rays_pinhole": (C.c_int, [_i32, _i32, _f, _f, _f, _f, C.POINTER(C.c_float), _i32, _vp, _vp, _vp, _vp]),
    "ced_generate_rays_hypercam": (C.c_int, [_i32, _i32, C.POINTER(C.c_float), C.POINTER(C.c_float), _f, _f, _f, _f, _f,
                                             C.POINTER(C.c_float), C.POINTER(C.c_float), _vp, _vp, _vp]),
    "ced_render_image_test_workspace_bytes": (_i64, [_i64, _i32, _i32, _f, _i32]),
    "ced_render_image_test": (C.c_int, [C.POINTER(FieldDesc), _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f, _f, _f, _f, _f,
                                        _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i64, _vp, C.POINTER(_i64),
                                        C.POINTER(FrameTrace), _vp, _vp]),
    "ced_render_frames_test_workspace_bytes": (_i64, [_i32, _i64, _i32, _i32, _f, _i32]),
    "ced_render_frames_test": (C.c_int, [C.POINTER(FieldDesc), _i32, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f, _f, _f, _f,
                                         _f, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, C.POINTER(_i64),
                                         C.POINTER(FrameTrace), _vp, _vp]),
    "ced_render_frames_test_sharded_workspace_bytes": (_i64, [_i32, _i64, _i64, _i32, _i32, _f, _i32]),
    "ced_render_frames_test_iterations": (_i32, [_f, _i32]),
    "ced_render_frames_test_host_bytes": (_i64, [_f, _i32]),
    "ced_render_frames_test_sharded": (C.c_int, [C.POINTER(FieldDesc), _i32, _i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f, _f, _f,
                                                 _f, _f, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, C.POINTER(_i64),
                                                 C.POINTER(FrameTrace), _vp, _vp, C.POINTER(ShardExchange)]),
    "ced_wall_clock_khz": (_i64, []),
    "ced_mlp_chain": (C.c_int, [_i64, _i32, _i32, _vp, C.POINTER(_i32), C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp), _i32, _vp]),
    "ced_march_all": (C.c_int, [_i64, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                _vp, _i64, _vp, _vp]),
    "ced_render_image_workspace_bytes": (_i64, [_i64, _i64]),
    "ced_render_image": (C.c_int, [C.POINTER(FieldDesc), _i64, _vp, _vp, _i64, _vp, _vp, _vp, _f, _f, _vp, _i32, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _i64, _vp, C.POINTER(_i64), _vp, _vp]),
    "ced_mlp_backward_dw_workspace_bytes": (_i64, [_i64, _i32, C.POINTER(_i32)]),
    "ced_mlp_backward_dw": (C.c_int, [_i64, _i32, _vp, C.POINTER(_i32), C.POINTER(_vp), C.POINTER(_vp), _vp, _vp, _vp, _i64, _vp]),
    "ced_train_inputs": (C.c_int, [_i64] + [_vp] * 13),
    "ced_train_warp": (C.c_int, [_i64, _vp, _vp, _i32, _i32, _f, C.POINTER(C.c_float), _vp, _vp, _vp, _vp]),
    "ced_train_warp_backward": (C.c_int, [_i64, _vp, _vp, _i32, _i32, _f, C.POINTER(C.c_float), _vp, _vp, _vp, _vp]),
    "ced_train_head_in": (C.c_int, [_i64] + [_vp] * 6),
    "ced_train_head_in_backward": (C.c_int, [_i64] + [_vp] * 6),
    "ced_render_image_gather": (C.c_int, [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
}


def _includes(path: str, seen=None) -> List[str]:
    """Transitive closure of the quoted #include files of a source (its rebuild dependencies)."""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return []
    seen.add(path)
    out = [path]
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(path).read(), flags=re.M):
        out += _includes(os.path.normpath(os.path.join(os.path.dirname(path), inc)), seen)
    return out


def build(force: bool = False, verbose: bool = False, extra_flags: Optional[List[str]] = None) -> str:
    """Compile every HIP source for gfx950 into the in-tree shared library (hipcc cross-compiles
    without a GPU).  One object per source, compiled in parallel and re-made only when the source or
    one of the headers it includes (transitively) is newer; objects live in build/obj (git-ignored)."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(_PKG, "csrc", s) for s in SOURCES]
    hipcc = os.environ.get("HIPCC", "hipcc")
    flags = [f for f in HIPCC_FLAGS if f != "-shared"] + list(extra_flags or [])
    if os.environ.get("CED_HALF_MFMA_K32", "0") == "1":
        # opt-in: the half-precision MLP blocks on v_mfma_f32_16x16x32_f16 (+11 % in f16x2).  Only for processes in
        # which no foreign kernel (torch, RCCL) can be co-resident with a field kernel: field_half_device.hpp, mfma_k32
        flags.append("-DCED_HALF_MFMA_K32")
    obj_dir = os.path.join(_ROOT, "build", "obj")
    os.makedirs(obj_dir, exist_ok=True)
    stamp = os.path.join(obj_dir, "flags.txt")
    flag_text = " ".join([hipcc] + flags + [f"{k}:{' '.join(v)}" for k, v in sorted(PER_SOURCE_FLAGS.items())])
    if not os.path.exists(stamp) or open(stamp).read() != flag_text:
        force = True
    jobs, objs = [], []
    for src in srcs:
        obj = os.path.join(obj_dir, os.path.basename(src) + ".o")
        objs.append(obj)
        deps = _includes(src)
        if force or not os.path.exists(obj) or any(os.path.getmtime(obj) < os.path.getmtime(d) for d in deps):
            jobs.append([hipcc] + flags + PER_SOURCE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
    if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
        return LIB_PATH

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    workers = max(1, min(len(jobs), int(os.environ.get("CED_BUILD_JOBS", str(min(os.cpu_count() or 1, 8))))))
    if jobs:
        with ThreadPoolExecutor(max_workers=workers) as pool:
            list(pool.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    with open(stamp, "w") as f:
        f.write(flag_text)
    return LIB_PATH


_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                "g.build()'); ced_nerf_amd has no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)      # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
        # tuning knobs from the environment: CED_OPTIONS="field_spread_tiles=0,march_early_out=1" (ced_set_option)
        for item in filter(None, os.environ.get("CED_OPTIONS", "").split(",")):
            key, _, value = item.partition("=")
            if handle.ced_set_option(key.strip().encode(), int(value)) != 0:
                raise RuntimeError(f"CED_OPTIONS: {handle.ced_last_error_string().decode()}")
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ced_last_error_string()
        raise RuntimeError(f"libcednerf_hip {what} failed (code {rc}): {msg.decode() if msg else ''}")


def header_symbols() -> List[str]:
    """Function names declared in include/cednerf_hip.h (used by the CPU test-suite)."""
    import re
    text = open(os.path.join(_ROOT, "include", "cednerf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ced_[a-z0-9_]+)\s*\(", text)))
