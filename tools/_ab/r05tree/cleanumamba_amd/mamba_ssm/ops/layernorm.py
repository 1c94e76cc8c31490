"""Residual add + LayerNorm in one HIP kernel each way (csrc/layernorm.hip).

Arithmetic of mamba-ssm's ``Block.forward`` with ``fused_add_norm=False`` as the reference runs it
(src/network/CleanUMamba.py:156-189, 288-294):  ``residual = hidden + residual`` (fp32),
``hidden = LayerNorm(residual)``.  Upstream offers the same fusion as a Triton kernel (``fused_add_norm=True``,
never enabled by the reference); this is the MI355X counterpart: d_model <= 2048 (16-byte vector kernels when d_model
and the strides are multiples of 8, element-access kernels for the pruned checkpoints' odd widths).
"""
import os

import torch

from ... import hip

_ENABLED = os.environ.get("CUM_FUSED_LN", "1") != "0"      # "0": the separate torch ops (A/B timing)


def supported(hidden, norm):
    dim = hidden.shape[-1]
    return (_ENABLED and hidden.is_cuda and isinstance(norm, torch.nn.LayerNorm) and norm.elementwise_affine
            and norm.weight.dtype == torch.float32 and 1 <= dim <= 2048 and hidden.dim() == 3
            and hidden.stride(2) == 1 and hidden.dtype in hip.IO_TYPES)       # (any d_model / strides: element kernels)


class AddLayerNormFn(torch.autograd.Function):
    """(hidden (B, L, D) f32/bf16 any batch/time strides, residual (B, L, D) f32 or None) ->
    (normed (B, L, D) in ``out_dtype``, residual_out (B, L, D) f32)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, hidden, residual, weight, bias, eps, out_dtype):
        hip.require_gpu(hidden, any_dtype=True)
        hip.require_gpu(residual, weight, bias)
        bsz, L, dim = hidden.shape
        if residual is not None:
            residual = residual.contiguous()
        weight = weight.contiguous()
        bias = None if bias is None else bias.contiguous()
        dev = hidden.device
        res_out = torch.empty(bsz, L, dim, dtype=torch.float32, device=dev)
        y = torch.empty(bsz, L, dim, dtype=out_dtype, device=dev)
        mean = torch.empty(bsz * L, dtype=torch.float32, device=dev)
        rstd = torch.empty(bsz * L, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            hip.check(hip.lib().cum_add_layernorm_fwd(
                hip.dtype_code(hidden.dtype), hip.dtype_code(out_dtype), bsz, L, dim, hip.ptr(hidden), hidden.stride(0),
                hidden.stride(1), hip.ptr(residual), hip.ptr(weight), hip.ptr(bias), float(eps), hip.ptr(res_out),
                hip.ptr(y), hip.ptr(mean), hip.ptr(rstd), hip.stream_ptr()))
        ctx.save_for_backward(res_out, mean, rstd, weight)
        ctx.h_dtype, ctx.has_res, ctx.has_bias = hidden.dtype, residual is not None, bias is not None
        # the parameters themselves (identity only: gradient-sink lookup), when they are f32 leaves
        ctx.params = [q for q in (weight, bias) if q is not None]
        if not all(q.is_leaf and q.dtype == torch.float32 for q in ctx.params):
            ctx.params = None
        return y, res_out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy, dres_out):
        res_out, mean, rstd, weight = ctx.saved_tensors
        bsz, L, dim = res_out.shape
        dev = res_out.device
        lib = hip.lib()
        if dy is None:                                    # only the residual stream was used downstream
            dy = torch.zeros(bsz, L, dim, dtype=torch.float32, device=dev)
        if dy.dtype not in hip.IO_TYPES or (dy.dtype != torch.float32 and ctx.h_dtype != torch.float32
                                            and dy.dtype != ctx.h_dtype):
            dy = dy.float()
        dy = dy.contiguous()
        if dres_out is not None:
            dres_out = dres_out.float().contiguous()
        need_h, need_r = ctx.needs_input_grad[0], ctx.has_res and ctx.needs_input_grad[1]
        same = ctx.h_dtype == torch.float32
        dx32 = torch.empty(bsz, L, dim, dtype=torch.float32, device=dev) if (need_r or (need_h and same)) else None
        dxh = torch.empty(bsz, L, dim, dtype=ctx.h_dtype, device=dev) if (need_h and not same) else None
        # weight / bias gradients straight into the flat gradient buffer when it takes them (training/flat_optim.py): no
        # AccumulateGrad add per vector
        from ...network.convstack import grad_sink
        sink = grad_sink(ctx.params) if (ctx.params and ctx.needs_input_grad[2]
                                         and (not ctx.has_bias or ctx.needs_input_grad[3])) else None
        if sink is not None:
            flat, idx, offs = sink
            dw = flat.grad[offs[0]:offs[0] + dim]
            db = flat.grad[offs[1]:offs[1] + dim] if ctx.has_bias else None
        else:
            dw = torch.empty(dim, dtype=torch.float32, device=dev)
            db = torch.empty(dim, dtype=torch.float32, device=dev) if ctx.has_bias else None
        ws = torch.empty(lib.cum_add_layernorm_bwd_workspace_elems(dim), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            hip.check(lib.cum_add_layernorm_bwd(
                hip.dtype_code(dy.dtype), hip.dtype_code(ctx.h_dtype), bsz * L, dim, hip.ptr(dy), hip.ptr(dres_out),
                hip.ptr(res_out), hip.ptr(mean), hip.ptr(rstd), hip.ptr(weight), hip.ptr(dx32), hip.ptr(dxh),
                hip.ptr(dw), hip.ptr(db), hip.ptr(ws), hip.stream_ptr()))
        dh = (dx32 if same else dxh) if need_h else None
        if sink is not None:
            flat.wrote(idx)
            dw = db = None
        return dh, (dx32 if need_r else None), dw, db, None, None


def add_layer_norm(hidden, residual, norm):
    """LayerNorm(hidden + residual) and the fp32 sum; output in the autocast dtype when autocast is on (what the
    projection that follows would cast it to), else in the weight dtype as nn.LayerNorm returns it."""
    out_dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else norm.weight.dtype
    if out_dtype not in hip.IO_TYPES or (hidden.dtype != torch.float32 and out_dtype != torch.float32
                                         and out_dtype != hidden.dtype):
        out_dtype = torch.float32
    if residual is not None and residual.dtype != torch.float32:
        residual = residual.float()
    return AddLayerNormFn.apply(hidden, residual, norm.weight, norm.bias, norm.eps, out_dtype)
