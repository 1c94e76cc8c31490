"""selective_scan_fn / selective_state_update over the HIP kernels.

Same call signatures as mamba-ssm 1.2.2's ``mamba_ssm.ops.selective_scan_interface``
and ``mamba_ssm.ops.triton.selective_state_update`` (the ops the reference reaches
through ``create_block`` at src/network/CleanUMamba.py:172-189 and Mamba.step at
:451-454).  Logical shapes are upstream's -- u, delta, z: (B, D, L); A: (D, N);
B, C: (B, N, L) or (B, 1, N, L) -- but any strides are accepted; the kernels are
tuned for channel-contiguous storage ((B, L, D) memory viewed as (B, D, L)), which
is what ``Mamba.forward`` here passes.  No CPU path: tensors must be on the GPU.

u, delta, z may be float32, bfloat16 or float16 (what autocast hands over, as upstream takes fp16/bf16): the kernels
read them and write out / du / ddelta / dz in that element type directly (cum_scan_shape.io_dtype); the
recurrence, A, B, C, D, the bias and their gradients are fp32.
"""
import ctypes
import os

import torch

from ... import hip


def _as3(t):
    """(B, 1, N, L) -> (B, N, L) view; rejects grouped B/C (the reference never uses groups)."""
    if t.dim() == 4:
        if t.shape[1] != 1:
            raise RuntimeError("selective_scan: only a single B/C group is supported")
        return t[:, 0]
    if t.dim() != 3:
        raise RuntimeError("selective_scan: B and C must be (B, N, L) or (B, 1, N, L)")
    return t


def _empty_like_layout(t):
    """Uninitialised (B, D, L) tensor with the channel-contiguous layout if t has it."""
    if t.stride(1) == 1 and t.shape[1] > 1:
        return torch.empty(t.shape[0], t.shape[2], t.shape[1], dtype=t.dtype, device=t.device).transpose(1, 2)
    return torch.empty(t.shape, dtype=t.dtype, device=t.device)


def _shape(u, delta, z, out, Bm, Cm, softplus):
    s = hip.ScanShape()
    s.batch, s.dim, s.len = u.shape
    s.dstate = Bm.shape[1]
    s.u_sb, s.u_sd, s.u_sl = u.stride()
    s.dt_sb, s.dt_sd, s.dt_sl = delta.stride()
    if z is not None:
        s.z_sb, s.z_sd, s.z_sl = z.stride()
    s.o_sb, s.o_sd, s.o_sl = out.stride()
    s.B_sb, s.B_sn, s.B_sl = Bm.stride()
    s.C_sb, s.C_sn, s.C_sl = Cm.stride()
    s.delta_softplus = int(bool(softplus))
    s.io_dtype = hip.dtype_code(u.dtype)
    return s


def scan_forward(s, u, delta, A, Bm, Cm, D, z, delta_bias, out, last, ckpt, time_parallel=True, y_pre=None):
    """Launch the forward scan described by ``s`` (a hip.ScanShape).  Where the sequential grid -- batch * ceil(dim / 64)
    * ceil(d_state / 8) waves -- would leave most of the chip idle (batch-1 file denoising, the 442K model, the pruned
    checkpoints), the library asks for a workspace and runs its time-parallel form (csrc/scan_seg.hip: segments walked
    from zero, composed with the scan's associative operator, re-walked from their true entering states).
    ``time_parallel=False`` pins the sequential kernels (tests compare the two)."""
    lib = hip.lib()
    n = lib.cum_scan_fwd_workspace_elems(s.batch, s.dim, s.dstate, s.len) if time_parallel else 0
    ws = torch.empty(n, dtype=torch.float32, device=u.device) if n > 0 else None
    with torch.cuda.device(u.device):
        hip.check(lib.cum_selective_scan_fwd_ws(ctypes.byref(s), hip.ptr(u), hip.ptr(delta), hip.ptr(A), hip.ptr(Bm),
                                                hip.ptr(Cm), hip.ptr(D), hip.ptr(z), hip.ptr(delta_bias), hip.ptr(out),
                                                hip.ptr(y_pre), hip.ptr(last), hip.ptr(ckpt), hip.ptr(ws), hip.stream_ptr()))


_KEEP_Y = os.environ.get("CUM_SCAN_KEEP_Y", "1") != "0"      # "0": the backward rebuilds y (A/B timing)


def keeps_y(s, time_parallel=True):
    """The forward of this shape can keep y before the gate for the backward (cum_scan_fwd_keeps_y): ask before
    allocating ``y_pre`` (a tensor of out's dtype and strides)."""
    return _KEEP_Y and bool(hip.lib().cum_scan_fwd_keeps_y(s.batch, s.dim, s.dstate, s.len, int(bool(time_parallel))))


TIME_PARALLEL = True        # module switch for tests / A-B timing: False pins the sequential forward AND backward kernels


def scan_backward_entry(bsz, dim, N, L, device):
    """(C entry point, workspace) of the selective-scan backward for this shape: the time-parallel form
    (cum_selective_scan_bwd_tp) where its plan segments the shape, else the sequential kernels."""
    lib = hip.lib()
    n = lib.cum_scan_bwd_tp_workspace_elems(bsz, dim, N, L) if TIME_PARALLEL else 0
    if n > 0:
        return lib.cum_selective_scan_bwd_tp, torch.empty(n, dtype=torch.float32, device=device)
    return lib.cum_selective_scan_bwd, torch.empty(max(lib.cum_scan_bwd_workspace_elems(bsz, dim, N, L), 1),
                                                   dtype=torch.float32, device=device)


class SelectiveScanFn(torch.autograd.Function):
    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False, save_ckpt=True):
        Bm, Cm = _as3(B), _as3(C)
        hip.require_gpu(u, delta, z, any_dtype=True)
        hip.require_gpu(A, Bm, Cm, D, delta_bias)
        if delta.dtype != u.dtype or (z is not None and z.dtype != u.dtype):
            raise RuntimeError("selective_scan: u, delta and z must share one element type")
        bsz, dim, L = u.shape
        N = A.shape[1]
        if A.shape[0] != dim or Bm.shape != (bsz, N, L) or Cm.shape != (bsz, N, L) or delta.shape != u.shape:
            raise RuntimeError("selective_scan: inconsistent shapes")
        A = A.contiguous()
        D = None if D is None else D.contiguous()
        delta_bias = None if delta_bias is None else delta_bias.contiguous()
        lib = hip.lib()
        out = _empty_like_layout(u)
        need_grad = bool(save_ckpt)      # decided by the caller: grad mode is always off inside forward()
        ckpt = None
        if need_grad:
            ckpt = torch.empty(max(lib.cum_scan_ckpt_elems(bsz, dim, N, L), 1), dtype=torch.float32, device=u.device)
        last = torch.empty(bsz, dim, N, dtype=torch.float32, device=u.device) if return_last_state else None
        s = _shape(u, delta, z, out, Bm, Cm, delta_softplus)
        # y before the gate, kept for the backward where the forward kernel can (d_state > 16, sequential form): the
        # backward then does not rebuild it (csrc/scan_bwd.hip YIN)
        y_pre = None
        if need_grad and z is not None and keeps_y(s, TIME_PARALLEL):
            y_pre = torch.empty_strided(out.shape, out.stride(), dtype=out.dtype, device=out.device)
        scan_forward(s, u, delta, A, Bm, Cm, D, z, delta_bias, out, last, ckpt, TIME_PARALLEL, y_pre=y_pre)
        ctx.delta_softplus = bool(delta_softplus)
        ctx.has_z = z is not None
        ctx.b4 = (B.dim() == 4, C.dim() == 4)
        ctx.save_for_backward(u, delta, A, Bm, Cm, D, z, delta_bias, ckpt, y_pre)
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dout, *unused):
        u, delta, A, Bm, Cm, D, z, delta_bias, ckpt, y_pre = ctx.saved_tensors
        if ckpt is None:
            raise RuntimeError("selective_scan backward called but forward saved no checkpoints")
        bsz, dim, L = u.shape
        N = A.shape[1]
        lib = hip.lib()
        dout = dout.to(u.dtype)
        du, ddelta = _empty_like_layout(u), _empty_like_layout(delta)
        dz = _empty_like_layout(z) if z is not None else None
        su = _shape(u, delta, z, dout, Bm, Cm, ctx.delta_softplus)     # o_* strides := dout's
        if y_pre is not None and y_pre.stride() != dout.stride():      # the kernel reads it with dout's strides
            y_pre = None
        gs = hip.ScanGradStrides()
        gs.du_sb, gs.du_sd, gs.du_sl = du.stride()
        gs.dd_sb, gs.dd_sd, gs.dd_sl = ddelta.stride()
        if dz is not None:
            gs.dz_sb, gs.dz_sd, gs.dz_sl = dz.stride()
        dA = torch.empty_like(A)
        dB = torch.empty(bsz, L, N, dtype=torch.float32, device=u.device)
        dC = torch.empty(bsz, L, N, dtype=torch.float32, device=u.device)
        dD = torch.empty_like(D) if D is not None else None
        dbias = torch.empty_like(delta_bias) if delta_bias is not None else None
        bwd, ws = scan_backward_entry(bsz, dim, N, L, u.device)
        with torch.cuda.device(u.device):
            hip.check(bwd(ctypes.byref(su), ctypes.byref(gs), hip.ptr(u), hip.ptr(delta), hip.ptr(A),
                          hip.ptr(Bm), hip.ptr(Cm), hip.ptr(D), hip.ptr(z),
                          hip.ptr(delta_bias), hip.ptr(dout), hip.ptr(y_pre), hip.ptr(ckpt), hip.ptr(du),
                          hip.ptr(ddelta), hip.ptr(dA), hip.ptr(dB), hip.ptr(dC),
                          hip.ptr(dD), hip.ptr(dz), hip.ptr(dbias), hip.ptr(ws),
                          hip.stream_ptr()))
        dB, dC = dB.transpose(1, 2), dC.transpose(1, 2)          # (B, N, L) views
        if ctx.b4[0]:
            dB = dB.unsqueeze(1)
        if ctx.b4[1]:
            dC = dC.unsqueeze(1)
        return du, ddelta, dA, dB, dC, dD, dz, dbias, None, None, None


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """out (and last_state (B, D, N) if requested); gate ``z`` is applied inside the kernel."""
    in_dtype = u.dtype
    io = in_dtype if in_dtype in hip.IO_TYPES else torch.float32
    u, delta = u.to(io), delta.to(io)
    z = None if z is None else z.to(io)
    A, B, C = A.float(), B.float(), C.float()
    D = None if D is None else D.float()
    delta_bias = None if delta_bias is None else delta_bias.float()
    # chunk-boundary states are written only when a backward can follow
    save = torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (u, delta, A, B, C, D, z, delta_bias))
    res = SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state, save)
    if return_last_state:
        return res[0].to(in_dtype), res[1]
    return res.to(in_dtype)


@torch.no_grad()
def selective_state_update(state, x, dt, A, B, C, D=None, z=None, dt_bias=None, dt_softplus=False):
    """One time step for a batch of streams; ``state`` (B, D, N) is updated in place.
    x, dt, z: (B, D); A: (D, N); B, C: (B, N).  Returns out (B, D)."""
    hip.require_gpu(state, x, dt, A, B, C, D, z, dt_bias)
    bsz, dim, N = state.shape
    if not state.is_contiguous():
        raise RuntimeError("selective_state_update: state must be contiguous")
    x, dt, A = x.contiguous(), dt.contiguous(), A.contiguous()
    z = None if z is None else z.contiguous()
    if B.stride(-1) != 1:
        B = B.contiguous()
    if C.stride(-1) != 1:
        C = C.contiguous()
    out = torch.empty_like(x)
    # named locals: a contiguous() temporary must outlive the launch (ctypes passes bare addresses)
    Dc = None if D is None else D.contiguous()
    bias_c = None if dt_bias is None else dt_bias.contiguous()
    with torch.cuda.device(x.device):
        hip.check(hip.lib().cum_selective_state_update(
            bsz, dim, N, hip.ptr(state), hip.ptr(x), hip.ptr(dt), hip.ptr(A), hip.ptr(B), B.stride(0),
            hip.ptr(C), C.stride(0), hip.ptr(Dc), hip.ptr(z), hip.ptr(bias_c), int(bool(dt_softplus)), hip.ptr(out),
            hip.stream_ptr()))
    return out
