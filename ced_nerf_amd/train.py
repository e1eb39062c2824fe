"""Training-time field and step: the training path.

What runs where:
  * sampling (occupancy-grid marching, stratified)            -> HIP  (nerfacc_api.OccGridEstimator.sampling)
  * hash-grid encode, forward and backward                    -> HIP  (ced_hash_encode / ced_hash_encode_backward)
  * compositing, forward and backward                         -> HIP  (render.rendering_train)
  * gradient-free densities of the sampling pass               -> HIP  (the fused inference kernel on the shared parameters)
  * the bias-free MLPs (motion, base, head, prediction heads) -> HIP  in both directions, no library GEMM:
    y = relu(x W^T) and dx = (dz W) * relu' through ced_linear (csrc/linear.hip), dW = dz^T x through ced_weight_grad
    (csrc/wgrad.hip); `_MlpFn` strings them into one autograd node per MLP.
  * the pieces between the MLPs (sample positions + Frequency / SH encodings; move / normalise / selector; trunc_exp and
    the head's input) -> HIP, one launch per direction each (ced_train_inputs / _warp / _head_in, csrc/train_glue.hip;
    `fused_glue = False` restores the torch statements they replaced, which the tests compare them with).  Only the
    sigmoid, the time encoders and the losses (huber, smooth-L1) remain torch element-wise kernels.
`TrainableField` keeps the parameter names and layout of `DNGPradianceField` (hash_table, xyz_wrap, mlp_base,
mlp_head as W[out][in]), so `to_inference()` hands the trained weights to the fused kernels unchanged, and
`tests/test_gpu_parity.py` checks that the two forwards agree.  Mirrors cednerf/model.py:354-488 (forward) and the
loss / optimiser lines of train_real.py:324-420 (occupancy refresh, smooth-L1 on colours, Adam, GradScaler, dynamic
ray batch).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib, ops
from .hashgrid import level_tables
from .render import rendering_train
from .utils import trunc_exp


class _HashFn(torch.autograd.Function):
    """Hash-grid encode with the gradient of hash_encoder_half.py:164-226 (table and positions), or -- temporal tables --
    of hash_encoder_inter.py:202-275 (table only: the reference's temporal encoder gives positions no gradient).
    `table` is the fp32 master parameter; `table_eval` (optional) is what the kernels READ: the fp16 copy of an fp16-table
    field (tiny-cuda-nn keeps fp32 master parameters and evaluates on their fp16 copy).  Gradients are accumulated in
    fp32 either way."""

    @staticmethod
    def forward(ctx, x, table, cfg, t=None, table_eval=None):
        tab = (table_eval if table_eval is not None else table).detach().contiguous()
        temporal = bool(cfg.get("temporal", False))
        desc, _ = ops.make_hash_desc(tab, cfg["base_res"], cfg["max_res"], cfg["n_levels"], cfg["log2_hashmap_size"], temporal)
        xc = x.detach().contiguous()
        tc = t.detach().reshape(-1).float().contiguous() if temporal else None
        out = ops.hash_encode(desc, xc, tc)
        ctx.save_for_backward(xc, tab, tc if tc is not None else xc.new_zeros(0))
        ctx.cfg = cfg
        return out

    # The table gradient is bound by the memory side's atomic request rate and nothing downstream of it in the
    # backward pass needs it -- only the optimiser does.  With `side` set (train_step does, `_HashFn.deferred`), it is
    # launched on a stream of its own, beside the position gradient and the motion MLP's backward that follow on the
    # main stream (MFMA / HBM-bound), and handed to the parameter by `join_deferred()` before the optimiser step.
    deferred = None            # None: everything on the current stream, the gradient returned through autograd
    side_blocks_per_level = 32
    _side_cache = None         # the side stream of the last step, kept for the next

    @staticmethod
    def backward(ctx, dy):
        x, tab, tc = ctx.saved_tensors
        cfg = ctx.cfg
        temporal = bool(cfg.get("temporal", False))
        desc, _ = ops.make_hash_desc(tab, cfg["base_res"], cfg["max_res"], cfg["n_levels"], cfg["log2_hashmap_size"], temporal)
        dy = dy.float().contiguous()
        d = _HashFn.deferred
        width = 8 if temporal else 2

        def table_grad(into=None):
            if temporal:
                return ops.hash_encode_backward_temporal(desc, x, tc, dy, grad_table=into)
            return ops.hash_encode_backward(desc, x, dy, grad_table=into, want_dx=False, dx_scaled=True)[0]

        want_dx = ctx.needs_input_grad[0] and not temporal
        if d is None or not ctx.needs_input_grad[1] or x.shape[0] == 0:
            grad_table = table_grad() if ctx.needs_input_grad[1] else None
            dx = ops.hash_encode_backward(desc, x, dy, want_dx=True, dx_scaled=True, want_table=False)[1] if want_dx else None
            return dx, grad_table, None, None, None
        main = torch.cuda.current_stream()
        grad_table = torch.zeros((int(desc.total_entries), width), device=x.device, dtype=torch.float32)    # main stream's pool
        d["side"].wait_stream(main)
        with torch.cuda.stream(d["side"]):
            # a few workgroups per CU (the atomics are fire-and-forget): room for the MLP kernels of the main stream.
            # Uncapped, the launch's 400 k workgroups crowd them out and the overlap is worth 0.3 ms instead of 1.2
            # (262 k rays: 12.4 -> 11.2 ms with 24-32 workgroups per level, 11.6 with 16, 11.9 with 48).
            _lib.lib().ced_set_option(b"hash_grad_blocks", _HashFn.side_blocks_per_level)
            try:
                table_grad(grad_table)
            finally:
                _lib.lib().ced_set_option(b"hash_grad_blocks", 0)
        for t_ in (x, dy, tab, tc):
            t_.record_stream(d["side"])                # their memory must not be handed out again before the kernel is over
        d["pending"].append(grad_table)
        dx = None
        if want_dx:
            _, dx = ops.hash_encode_backward(desc, x, dy, want_dx=True, dx_scaled=True, want_table=False)
        return dx, None, None, None, None


def begin_deferred_table_grad(device) -> None:
    """From now on `_HashFn.backward` runs the table gradient on a side stream; `join_deferred_table_grad` ends it."""
    d = _HashFn._side_cache
    if d is None or d["device"] != torch.device(device):
        d = {"side": torch.cuda.Stream(device=device), "pending": [], "device": torch.device(device)}
    d["pending"].clear()
    _HashFn.deferred = d


def join_deferred_table_grad(param: torch.nn.Parameter) -> None:
    """Waits (stream-wise, not the host) for the side stream and adds the deferred gradients into param.grad."""
    d = _HashFn.deferred
    _HashFn.deferred = None
    if d is None:
        return
    torch.cuda.current_stream().wait_stream(d["side"])
    for g in d["pending"]:
        param.grad = g if param.grad is None else param.grad + g
    d["pending"].clear()
    _HashFn._side_cache = d            # keep the stream for the next step


class _MlpFn(torch.autograd.Function):
    """A bias-free ReLU MLP (linear last layer) as ONE autograd node on the HIP kernels: forward y_l = relu(y_{l-1} W_l^T)
    through ced_linear; backward dW_l = dz_l^T y_{l-1} through ced_weight_grad and dz_{l-1} = (dz_l W_l) * [y_{l-1} > 0]
    through ced_linear with the ReLU derivative fused.  No library GEMM (tcnn FullyFusedMLP forward/backward,
    cednerf/model.py:200-222,280-344)."""

    fused = True          # one launch per direction (ced_mlp_chain); False: layer by layer (ced_linear), same bits
    fused_dw = True       # backward: the weight gradients inside the walk (ced_mlp_backward_dw); False: ced_weight_grad

    @staticmethod
    def forward(ctx, x, *weights):
        h = x.detach().float().contiguous()
        acts = [h]
        ws = [w.detach().float().contiguous() for w in weights]
        if _MlpFn.fused and h.shape[0] > 0:
            acts += ops.mlp_chain(h, ws)
            h = acts[-1]
        else:
            for i, w in enumerate(ws):
                h = ops.linear(h, w, relu=i < len(ws) - 1)
                acts.append(h)
        ctx.save_for_backward(*acts[:-1], *ws)
        ctx.n_layers = len(ws)
        return h

    @staticmethod
    def backward(ctx, dy):
        n = ctx.n_layers
        acts, ws = ctx.saved_tensors[:n], ctx.saved_tensors[n:]
        dz = dy.float().contiguous()
        grads = [None] * n
        widths = [ws[0].shape[1]] + [w.shape[0] for w in ws]
        if (_MlpFn.fused and _MlpFn.fused_dw and dz.shape[0] > 0 and all(ctx.needs_input_grad[1:1 + n])
                and ops.mlp_backward_dw_supported(widths)):
            # the whole backward in one launch: every dW_l from the registers of the walk (ced_mlp_backward_dw)
            g0, dws = ops.mlp_backward_dw(dz, list(ws), list(acts), want_g0=bool(ctx.needs_input_grad[0]))
            return (g0, *dws)
        if _MlpFn.fused and dz.shape[0] > 0:
            # g_l = gradient at layer l's input: needed by dW_{l-1} (l >= 1) and, for l = 0, by the caller
            need_w = [bool(ctx.needs_input_grad[1 + l]) for l in range(n)]
            want = [bool(ctx.needs_input_grad[0])] + [need_w[l - 1] for l in range(1, n)]    # what is stored (the walk is whole)
            g = ops.mlp_chain(dz, list(ws), backward=True, masks=[None] + list(acts[1:]), want=want)
            ups = g[1:] + [dz]                                   # gradient at layer l's OUTPUT = g_{l+1}
            for l in range(n):
                if need_w[l]:
                    grads[l] = ops.weight_grad(acts[l], ups[l])
            return (g[0] if ctx.needs_input_grad[0] else None, *grads)
        for l in reversed(range(n)):
            if ctx.needs_input_grad[1 + l]:
                grads[l] = ops.weight_grad(acts[l], dz)
            if l > 0 or ctx.needs_input_grad[0]:
                # acts[l] is the (post-ReLU) output of layer l-1: its positive entries are where the gradient passes
                dz = ops.linear(dz, ws[l], transpose_w=True, mask=acts[l] if l > 0 else None)
        return (dz if ctx.needs_input_grad[0] else None, *grads)


class _WarpFn(torch.autograd.Function):
    """move / normalise / clamp / selector of cednerf/model.py:356-383 in one launch per direction (ced_train_warp)."""

    @staticmethod
    def forward(ctx, pos, mo, aabb6, moving_step, use_div):
        mo_c = mo.detach().float().contiguous()
        xn, move, sel = ops.train_warp(pos, mo_c, aabb6, moving_step, use_div)
        ctx.save_for_backward(pos, mo_c)
        ctx.cfg = (aabb6, moving_step, use_div)
        ctx.mark_non_differentiable(sel)
        ctx.set_materialize_grads(False)
        return xn, move, sel

    @staticmethod
    def backward(ctx, d_xn, d_move, _d_sel):
        pos, mo = ctx.saved_tensors
        aabb6, moving_step, use_div = ctx.cfg
        if d_xn is None:
            d_xn = torch.zeros_like(pos)
        d_mo = ops.train_warp_backward(pos, mo, aabb6, moving_step, use_div, d_xn.float().contiguous(),
                                       None if d_move is None else d_move.float().contiguous())
        return None, d_mo, None, None, None


class _HeadInFn(torch.autograd.Function):
    """density = trunc_exp(raw - 1) * selector and the colour head's input [SH, geo] (cednerf/model.py:414-417,455,
    utils.py:27-43) in one launch per direction (ced_train_head_in)."""

    @staticmethod
    def forward(ctx, bout, sh, sel):
        b = bout.detach().float().contiguous()
        head_in, sigma = ops.train_head_in(b, sh, sel)
        ctx.save_for_backward(b, sel)
        ctx.set_materialize_grads(False)
        return head_in, sigma

    @staticmethod
    def backward(ctx, d_head_in, d_sigma):
        b, sel = ctx.saved_tensors
        d_bout = ops.train_head_in_backward(b, sel, None if d_head_in is None else d_head_in.float().contiguous(),
                                            None if d_sigma is None else d_sigma.float().contiguous())
        return d_bout, None, None


class _LinearFn(torch.autograd.Function):
    """One bias-free layer through the library (torch -> rocBLAS) for y and dx: kept ONLY as the A/B reference of the
    tests (`TrainableField.hip_mlp = False`); the training path itself uses `_MlpFn`."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = dy @ w if ctx.needs_input_grad[0] else None
        dw = ops.weight_grad(x.contiguous(), dy.float().contiguous()) if ctx.needs_input_grad[1] else None
        return dx, dw


def _frequency4(v: torch.Tensor) -> torch.Tensor:
    """tcnn Frequency(n_frequencies=4) on [N, D]: [dim][freq][sin, cos] of pi * 2^k * v."""
    ang = math.pi * v[:, :, None] * (2.0 ** torch.arange(4, device=v.device, dtype=torch.float32))
    return torch.stack([torch.sin(ang), torch.sin(ang + 0.5 * math.pi)], dim=-1).reshape(v.shape[0], 8 * v.shape[1])


class TrainableField(torch.nn.Module):
    """Differentiable DNGPradianceField (cednerf/model.py:97-488) with the inference module's parameters."""

    def __init__(self, params: Dict, device="cuda", use_feat_predict: bool = False, use_weight_predict: bool = False,
                 seed: int = 0):
        super().__init__()
        h = params["hash"]
        # The reference's table types (round 4): an fp16 table trains as tiny-cuda-nn does -- an fp32 MASTER parameter, an
        # fp16 copy that the kernels evaluate (refreshed after every optimiser step, `sync_half_table`), gradients in fp32
        # (cednerf/model.py:262-276 under train_real.py:330's autocast); the temporal table ([E, 8]: 4 key-frames x 2
        # features) has the reference's table-only backward (hash_encoder_inter.py:202-275).
        self.temporal = bool(h.get("temporal", False))
        self.table_f16 = np.asarray(h["table"]).dtype == np.float16
        self.hash_cfg = dict(base_res=h["base_res"], max_res=h["max_res"], n_levels=h["n_levels"],
                             log2_hashmap_size=h["log2_hashmap_size"])
        if self.temporal:
            self.hash_cfg["temporal"] = True
        assert level_tables(h["base_res"], h["max_res"], h["n_levels"], h["log2_hashmap_size"])["total"] == h["table"].shape[0]
        assert h["table"].shape[1] == (8 if self.temporal else 2)
        T = lambda a: torch.nn.Parameter(torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device))
        self.register_buffer("aabb", torch.from_numpy(np.asarray(params["aabb"], np.float32)).to(device))
        self.moving_step = float(params["moving_step"])
        self.use_div_offsets = bool(params["use_div_offsets"])
        self.time_mode = int(params["time_mode"])
        self.hash_table = T(h["table"])                       # fp32 master (converted exactly from an fp16 table)
        if self.table_f16:
            self.register_buffer("hash_table_half", self.hash_table.detach().half(), persistent=False)
        self.xyz_wrap = torch.nn.ParameterList([T(w) for w in params["xyz_wrap"]])
        self.mlp_base = torch.nn.ParameterList([T(w) for w in params["mlp_base"]])
        self.mlp_head = torch.nn.ParameterList([T(w) for w in params["mlp_head"]])
        if self.time_mode:
            from .encoder import SinusoidalEncoder, SinusoidalEncoderWithExp
            self.time_encoder = SinusoidalEncoder(1, 0, 4, True)
            self.time_encoder_feat = SinusoidalEncoderWithExp(1, 0, 4, True)
        # the training-only prediction heads (cednerf/model.py:312-344): Frequency(4) on (x_move, t) -> 64 -> 32 / 1
        self.use_feat_predict, self.use_weight_predict = bool(use_feat_predict), bool(use_weight_predict)
        gen = torch.Generator().manual_seed(seed)

        def xavier(o, i):
            lim = math.sqrt(6.0 / (i + o))
            return torch.nn.Parameter(torch.empty(o, i).uniform_(-lim, lim, generator=gen).to(device))
        if self.use_feat_predict:
            self.mlp_feat_prediction = torch.nn.ParameterList([xavier(64, 32), xavier(2 * h["n_levels"], 64)])
        if self.use_weight_predict:
            self.mlp_weight_prediction = torch.nn.ParameterList([xavier(64, 32), xavier(1, 64)])

    hip_mlp = True                  # False: the layers' y and dx through the library (A/B reference of the tests)
    hip_weight_grad = True          # (with hip_mlp False) False: dW through the library as well

    def _mlp(self, x, weights):
        if self.hip_mlp:
            return _MlpFn.apply(x, *weights)
        lin = _LinearFn.apply if self.hip_weight_grad else (lambda a, w: a @ w.t())
        for w in weights[:-1]:
            x = torch.relu(lin(x, w))
        return lin(x, weights[-1])

    fused_glue = True               # False: the element-wise pieces as torch statements (the A/B reference of the tests)

    def forward(self, positions: torch.Tensor, t: torch.Tensor, directions: torch.Tensor, return_internal: bool = False):
        """positions [N,3] world, t [N,1] in [0,1], directions [N,3] -> (rgb [N,3], sigma [N]); with return_internal
        also the dict of cednerf/model.py:428-443 (`move`, `selector`, `latent_losses`, `weight_losses`)."""
        if not self.fused_glue:
            return self._forward_torch(positions, t, directions, return_internal)
        n = positions.shape[0]
        tt = t.reshape(-1).float()
        if tt.shape[0] == 1 and n != 1:
            tt = tt.expand(n)
        pos, enc, sh, tt = ops.train_inputs(n, positions=positions.detach().float().contiguous(),
                                            directions=directions.detach().float().contiguous(), timestamps=tt.contiguous())
        return self._forward_core(pos, enc, sh, tt, return_internal)

    def forward_rays(self, rays_o, rays_d, ray_indices, t_starts, t_ends, ts_per_ray, return_internal: bool = False):
        """The same on ray-packed samples (what `rendering`'s rgb_sigma_fn hands over, cednerf/utils.py:86-104): positions
        rays_o[r] + rays_d[r] * (t_start + t_end) / 2, directions rays_d[r], time ts_per_ray[r] -- gathered inside the
        input kernel instead of by three torch index kernels."""
        if not self.fused_glue:
            pos = rays_o[ray_indices] + rays_d[ray_indices] * ((t_starts + t_ends)[:, None] / 2.0)
            return self._forward_torch(pos, ts_per_ray.reshape(-1, 1)[ray_indices], rays_d[ray_indices], return_internal)
        pos, enc, sh, tt = ops.train_inputs(ray_indices.shape[0], rays_o.contiguous(), rays_d.contiguous(),
                                            ray_indices.contiguous(), t_starts.contiguous(), t_ends.contiguous(),
                                            ts_per_ray.reshape(-1).float().contiguous())
        return self._forward_core(pos, enc, sh, tt, return_internal)

    def _forward_core(self, pos, enc, sh, tt, return_internal):
        # the box as python floats for the fused warp kernel, re-read whenever the buffer is replaced or written in place
        # (load_state_dict, a checkpoint load, aabb.copy_): keyed like DNGPradianceField._descriptor
        key = (self.aabb.data_ptr(), self.aabb._version)
        if getattr(self, "_aabb6_key", None) != key:
            object.__setattr__(self, "_aabb6", [float(v) for v in self.aabb.detach().cpu().tolist()])
            object.__setattr__(self, "_aabb6_key", key)
        mo = self._mlp(enc, list(self.xyz_wrap))
        xn, move, sel = _WarpFn.apply(pos, mo, self._aabb6, self.moving_step, self.use_div_offsets)   # model.py:356-383
        hash_feat = feat = _HashFn.apply(xn, self.hash_table, self.hash_cfg, tt, self._table_eval())
        if self.time_mode:                                                          # model.py:386-403 (no gradient there)
            with torch.no_grad():
                t1 = tt[:, None]
                te = self.time_encoder(t1) if self.time_mode == 1 else \
                    self.time_encoder_feat(t1, move.detach().norm(dim=-1, keepdim=True))
            feat = torch.cat([feat, te], dim=-1)
        bout = self._mlp(feat, list(self.mlp_base))
        head_in, sigma = _HeadInFn.apply(bout, sh, sel)                             # model.py:105,414-417,455
        rgb = torch.sigmoid(self._mlp(head_in, list(self.mlp_head)))
        if not return_internal:
            return rgb, sigma
        selector = sel > 0.5
        internal = {"move": move, "selector": selector}                              # model.py:428-443
        if self.use_feat_predict or self.use_weight_predict:
            x_move = (pos + move - self.aabb[:3]) / (self.aabb[3:] - self.aabb[:3])    # the unclamped normalised position
            temp = _frequency4(torch.cat([x_move, tt[:, None]], dim=-1))
            if self.use_feat_predict:
                predict_feat = self._mlp(temp, list(self.mlp_feat_prediction))
                internal["latent_losses"] = torch.nn.functional.huber_loss(predict_feat, hash_feat, reduction="none") \
                    * selector[:, None].to(predict_feat.dtype)
            if self.use_weight_predict:
                internal["weight_losses"] = self._mlp(temp, list(self.mlp_weight_prediction))
        return rgb, {"density": sigma[:, None], "interal_output": internal}

    def _forward_torch(self, positions, t, directions, return_internal: bool = False):
        """The element-wise pieces as torch statements (round 2's graph; kept as the reference the fused pieces are tested
        against)."""
        x, tt = positions.float(), t.reshape(-1, 1).float()
        enc = _frequency4(torch.cat([x, tt], dim=-1))                               # [N,32]
        mo = self._mlp(enc, list(self.xyz_wrap))
        move = mo[:, :3] * self.moving_step                                         # model.py:356-363
        if self.use_div_offsets:
            move = move + torch.tanh(mo[:, 3:6]) * self.moving_step
        xn = (x + move - self.aabb[:3]) / (self.aabb[3:] - self.aabb[:3])           # model.py:378-379
        selector = ((xn > 0.0) & (xn < 1.0)).all(dim=-1)                            # model.py:383
        hash_feat = feat = _HashFn.apply(xn.clamp(0.0, 1.0), self.hash_table, self.hash_cfg, tt, self._table_eval())
        if self.time_mode:                                                          # model.py:386-403 (no gradient there)
            with torch.no_grad():
                mn = move.detach().norm(dim=-1, keepdim=True)
                te = self.time_encoder(tt) if self.time_mode == 1 else self.time_encoder_feat(tt, mn)
            feat = torch.cat([feat, te], dim=-1)
        bout = self._mlp(feat, list(self.mlp_base))
        sigma = trunc_exp(bout[:, 0] - 1.0) * selector.to(bout.dtype)               # model.py:105,414-417
        d = directions.float()
        u = (d / d.norm(dim=-1, keepdim=True) + 1.0) / 2.0
        w = u * 2.0 - 1.0
        sh = torch.stack([torch.full_like(w[:, 0], 0.28209479177387814), -0.48860251190291987 * w[:, 1],
                          0.48860251190291987 * w[:, 2], -0.48860251190291987 * w[:, 0]], dim=-1)
        rgb = torch.sigmoid(self._mlp(torch.cat([sh, bout[:, 1:]], dim=-1), list(self.mlp_head)))
        if not return_internal:
            return rgb, sigma
        internal = {"move": move, "selector": selector}                              # model.py:428-443
        if self.use_feat_predict or self.use_weight_predict:
            temp = _frequency4(torch.cat([xn, tt], dim=-1))
            if self.use_feat_predict:
                predict_feat = self._mlp(temp, list(self.mlp_feat_prediction))
                internal["latent_losses"] = torch.nn.functional.huber_loss(predict_feat, hash_feat, reduction="none") \
                    * selector[:, None].to(predict_feat.dtype)
            if self.use_weight_predict:
                internal["weight_losses"] = self._mlp(temp, list(self.mlp_weight_prediction))
        return rgb, {"density": sigma[:, None], "interal_output": internal}

    def sync_half_table(self) -> None:
        """fp16-table fields: refresh the evaluated fp16 copy from the fp32 master (after an optimiser step or a load).
        In place, so that the shared inference module re-packs / re-reads it (its descriptor is keyed on tensor versions)."""
        if self.table_f16:
            with torch.no_grad():
                self.hash_table_half.copy_(self.hash_table)

    def _table_eval(self):
        return self.hash_table_half if self.table_f16 else None

    def export_params(self) -> Dict:
        g = lambda p: p.detach().cpu().numpy()
        return dict(aabb=g(self.aabb), moving_step=self.moving_step, use_div_offsets=self.use_div_offsets,
                    time_mode=self.time_mode,
                    hash=dict(table=g(self.hash_table).astype(np.float16) if self.table_f16 else g(self.hash_table), **self.hash_cfg),
                    xyz_wrap=[g(p) for p in self.xyz_wrap], mlp_base=[g(p) for p in self.mlp_base],
                    mlp_head=[g(p) for p in self.mlp_head])

    def to_inference(self, device="cuda", mlp_precision: str = "f32"):
        """The fused-kernel module with (a copy of) these weights."""
        from .model import DNGPradianceField
        return DNGPradianceField.from_params(self.export_params(), device, mlp_precision=mlp_precision).eval()

    def shared_inference(self):
        """A fused-kernel module on the SAME parameter tensors (no copy): its weight blob is re-packed whenever an
        optimiser step has changed them (the descriptor is keyed on the tensors' versions).  Used for the
        gradient-free density queries of the sampling pass and of the occupancy-grid refresh."""
        if getattr(self, "_shared", None) is None:
            from .model import DNGPradianceField
            h = self.hash_cfg
            m = DNGPradianceField(aabb=self.aabb, dst_resolution=h["max_res"], base_resolution=h["base_res"],
                                  n_levels=h["n_levels"], log2_hashmap_size=h["log2_hashmap_size"],
                                  moving_step=self.moving_step, use_div_offsets=self.use_div_offsets,
                                  use_time_embedding=self.time_mode != 0, use_time_attenuation=self.time_mode == 2,
                                  temporal_hash=self.temporal, hash_dtype=torch.float16 if self.table_f16 else torch.float32)
            m.hash_table = torch.nn.Parameter(self.hash_table_half, requires_grad=False) if self.table_f16 else self.hash_table
            m.xyz_wrap, m.mlp_base, m.mlp_head = self.xyz_wrap, self.mlp_base, self.mlp_head
            m.aabb = self.aabb
            object.__setattr__(self, "_shared", m.eval())          # not a sub-module: the parameters are ours
        return self._shared


def next_num_rays(num_rays: int, n_rendering_samples: int, target_sample_batch_size: int) -> int:
    """Dynamic ray batch of train_real.py:354-360: keep the number of rendered samples per step near the target."""
    if target_sample_batch_size <= 0 or n_rendering_samples <= 0:
        return num_rays
    return int(num_rays * (target_sample_batch_size / float(n_rendering_samples)))


def refresh_occupancy(field: TrainableField, estimator, step: int, timestamps: torch.Tensor, render_step_size: float,
                      occ_thre: float = 1e-2) -> None:
    """train_real.py:324-336: the occupancy grid's EMA update, every n steps, with the current density evaluated by
    the fused inference kernel on the shared parameters."""
    from .model import make_occ_eval_fn
    was_training = estimator.training
    estimator.train()
    estimator.update_every_n_steps(step=step, occ_eval_fn=make_occ_eval_fn(field.shared_inference(), timestamps, render_step_size),
                                   occ_thre=occ_thre)
    estimator.train(was_training)


def train_step(field: TrainableField, estimator, optimizer, rays_o: torch.Tensor, rays_d: torch.Tensor,
               timestamps: torch.Tensor, target_rgb: torch.Tensor, render_step_size: float, near_plane: float = 0.0,
               far_plane: float = 1e10, cone_angle: float = 0.0, alpha_thre: float = 0.0,
               render_bkgd: Optional[torch.Tensor] = None, grad_scaler=None, native_sampling: bool = True,
               overlap_table_grad: bool = True) -> Dict:
    """One optimisation step on a batch of rays (train_real.py:339-380): stratified occupancy-grid sampling with the
    current density (no gradient), differentiable field + compositing, smooth-L1 colour loss, optimiser step."""
    n_rays = rays_o.shape[0]
    ts = timestamps.reshape(-1, 1).float()
    if ts.shape[0] == 1:
        ts = ts.expand(n_rays, 1)

    field.sync_half_table()                               # fp16-table fields: the evaluated copy follows the master
    fused = field.shared_inference()

    def sigma_fn(t_starts, t_ends, ray_indices):
        # gradient-free density of every marched sample: the fused inference kernel on the current weights
        fused.train()                                     # per-ray timestamps, as in training (utils.py:86-104)
        _, sigma = fused.query_rays(rays_o, rays_d, ray_indices, t_starts, t_ends, ts, want_rgb=False)
        return sigma

    fused.train()
    ray_indices, t_starts, t_ends = estimator.sampling(rays_o, rays_d, sigma_fn=sigma_fn, near_plane=near_plane,
                                                       far_plane=far_plane, render_step_size=render_step_size,
                                                       stratified=True, cone_angle=cone_angle, alpha_thre=alpha_thre,
                                                       sigma_field=(fused, ts, True) if native_sampling else None)

    with_heads = field.use_feat_predict or field.use_weight_predict

    def rgb_sigma_fn(t_starts, t_ends, ray_indices):
        return field.forward_rays(rays_o, rays_d, ray_indices, t_starts, t_ends, ts, return_internal=with_heads)

    colors, opacities, depths, extras = rendering_train(t_starts, t_ends, ray_indices, n_rays, rgb_sigma_fn,
                                                         render_bkgd=render_bkgd)
    loss = torch.nn.functional.smooth_l1_loss(colors, target_rgb)
    if "latent_losses" in extras:                          # train_real.py:400-409
        loss = loss + extras["latent_losses"].mean()
    if "weight_losses" in extras:
        loss = loss + extras["weight_losses"].mean()
    optimizer.zero_grad(set_to_none=True)
    if overlap_table_grad:
        begin_deferred_table_grad(rays_o.device)
    try:
        if grad_scaler is not None:                        # train_real.py:252,414-419 (GradScaler(2**10))
            grad_scaler.scale(loss).backward()
        else:
            loss.backward()
    finally:
        if overlap_table_grad:
            join_deferred_table_grad(field.hash_table)
    if grad_scaler is not None:
        grad_scaler.step(optimizer)
        grad_scaler.update()
    else:
        optimizer.step()
    field.sync_half_table()
    return {"loss": float(loss.detach()), "n_samples": int(t_starts.shape[0])}
