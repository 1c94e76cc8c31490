"""causal_conv1d_fn / causal_conv1d_update over the HIP kernels.

Same signatures as causal-conv1d 1.1.0 (call pattern in the reference:
src/network/S4/MambaS4.py:454-463; torch equivalent ``act(conv1d(x)[..., :L])`` at
:455).  x: (B, D, L) logical, any strides (channel-contiguous preferred);
weight: (D, W); W <= 4.  No CPU path.

x may be float32, bfloat16 or float16 (what autocast hands over): the kernels read and write that element type
directly (cum_conv_shape.io_dtype) and compute in fp32; weights and their gradients are always fp32.
"""
import ctypes

import torch

from .. import hip


def _shape(x, y, width, silu):
    s = hip.ConvShape()
    s.batch, s.dim, s.len = x.shape
    s.width = width
    s.x_sb, s.x_sd, s.x_sl = x.stride()
    s.y_sb, s.y_sd, s.y_sl = y.stride()
    s.silu = int(silu)
    s.io_dtype = hip.dtype_code(x.dtype)
    return s


def _alloc_like(x):
    if x.stride(1) == 1 and x.shape[1] > 1:
        return torch.empty(x.shape[0], x.shape[2], x.shape[1], dtype=x.dtype, device=x.device).transpose(1, 2)
    return torch.empty(x.shape, dtype=x.dtype, device=x.device)


class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, weight, bias=None, activation=None):
        if activation not in (None, "silu", "swish"):
            raise NotImplementedError("activation must be None, silu, or swish")
        hip.require_gpu(x, any_dtype=True)
        hip.require_gpu(weight, bias)
        hip.dtype_code(x.dtype)
        if x.dim() != 3 or weight.dim() != 2 or weight.shape[0] != x.shape[1]:
            raise RuntimeError("causal_conv1d: x must be (B, D, L) and weight (D, W)")
        weight = weight.contiguous()
        bias = None if bias is None else bias.contiguous()
        silu = activation is not None
        y = _alloc_like(x)
        s = _shape(x, y, weight.shape[1], silu)
        with torch.cuda.device(x.device):
            hip.check(hip.lib().cum_causal_conv1d_fwd(ctypes.byref(s), hip.ptr(x), hip.ptr(weight), hip.ptr(bias),
                                                      hip.ptr(y), hip.stream_ptr()))
        ctx.silu = silu
        ctx.save_for_backward(x, weight, bias)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        x, weight, bias = ctx.saved_tensors
        dy = dy.to(x.dtype)
        lib = hip.lib()
        dx = _alloc_like(x)
        dw = torch.empty_like(weight)
        db = torch.empty_like(bias) if bias is not None else None
        bsz, dim, L = x.shape
        ws = torch.empty(max(lib.cum_conv_bwd_workspace_elems(bsz, dim, L, weight.shape[1]), 1),
                         dtype=torch.float32, device=x.device)
        s = _shape(x, dy, weight.shape[1], ctx.silu)
        with torch.cuda.device(x.device):
            hip.check(lib.cum_causal_conv1d_bwd(ctypes.byref(s), hip.ptr(x), hip.ptr(weight), hip.ptr(bias),
                                                hip.ptr(dy), hip.ptr(dx), dx.stride(0), dx.stride(1), dx.stride(2),
                                                hip.ptr(dw), hip.ptr(db), hip.ptr(ws), hip.stream_ptr()))
        return dx, dw, db, None


def causal_conv1d_fn(x, weight, bias=None, activation=None):
    in_dtype = x.dtype
    if in_dtype not in hip.IO_TYPES:
        x = x.float()
    return CausalConv1dFn.apply(x, weight.float(), None if bias is None else bias.float(), activation).to(in_dtype)


@torch.no_grad()
def causal_conv1d_update(x, conv_state, weight, bias=None, activation=None):
    """x: (B, D); conv_state: (B, D, W) contiguous, shifted in place; returns (B, D)."""
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")
    hip.require_gpu(x, conv_state, weight, bias)
    if not conv_state.is_contiguous():
        raise RuntimeError("causal_conv1d_update: conv_state must be contiguous")
    bsz, dim, W = conv_state.shape
    x = x.contiguous()
    y = torch.empty_like(x)
    # named locals: a contiguous() temporary must outlive the launch (ctypes passes bare addresses)
    wc = weight.contiguous()
    bc = None if bias is None else bias.contiguous()
    with torch.cuda.device(x.device):
        hip.check(hip.lib().cum_causal_conv1d_update(bsz, dim, W, hip.ptr(conv_state), hip.ptr(x), hip.ptr(wc),
                                                     hip.ptr(bc), int(activation is not None), hip.ptr(y),
                                                     hip.stream_ptr()))
    return y
