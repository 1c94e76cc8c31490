"""Mamba mixer and Block over the HIP kernels.

Module contract of mamba-ssm 1.2.2's ``mamba_ssm.modules.mamba_simple`` as the
reference relies on it: class name ``Mamba`` (src/network/CleanUMamba.py:540),
sub-modules ``in_proj / x_proj / dt_proj / out_proj`` (nn.Linear), ``conv1d``
(nn.Conv1d, groups = d_inner), parameters ``A_log``, ``D`` and the mutable ints
``d_model, d_inner, d_state, dt_rank, d_conv, expand, layer_idx`` that the pruning
code edits in place (src/pruning/pruninggroup.py:340-352, CleanUMamba.py:336-349,
511-545).  Shapes are therefore always read from the tensors at call time.
Structure mirror inside the reference: src/network/S4/MambaS4.py:367-473.

Layout: activations stay channel-contiguous (B, L, C) from in_proj to out_proj --
the layout the projection GEMMs produce and the one the scan / conv kernels
coalesce on.  The (B, D, L) tensors of the upstream formulation appear only as
transposed views.
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ... import hip
from ..ops import layernorm as _ln
from ..ops.selective_scan_interface import selective_scan_fn, selective_state_update

# Module-level names that callers monkey-patch (src/examples/using_pruning_groups.py:26-27
# sets ``causal_conv1d_fn = None`` to force the nn.Conv1d path so that hooks fire).
from ...causal_conv1d import causal_conv1d_fn, causal_conv1d_update


class _ProjFn(torch.autograd.Function):
    """y = x @ w.T for the four bias-free projections of the Mamba block, all three GEMMs on the library's own MFMA
    kernels: the forward and the data gradient on cum_gemm_nt (csrc/gemm.hip; 256 x 256 / 128 x 128 tiles, the narrow
    ones -- x_proj forward, dt_proj data gradient -- on the 64 x 64 kernel that splits K over its four waves), the
    weight gradient dW[n][k] = sum_m dY[m][n] X[m][k] on cum_gemm_tn, whose split over the only 9984-long row axis is
    what keeps the chip busy there.  Packed (cast, zero-padded, for the data gradient transposed) weight operands come
    out of the model's per-step pack plan (network/convstack.py PackPlan) with the conv weights: no cast kernels."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, x, w, cd):
        from ...network import convstack as cs
        N, K = w.shape
        xc = x if x.dtype == cd else x.to(cd)
        x2 = xc.reshape(-1, K)
        if N % 8:        # odd widths (pruned checkpoints): rows of the result keep 16-byte alignment in a padded buffer
            y = cs.proj_fwd(x2, w, cd, out=torch.empty(x2.shape[0], cs.rup(N, 8), dtype=cd, device=x2.device)[:, :N])
        else:
            y = cs.proj_fwd(x2, w, cd)
        ctx.save_for_backward(x2, w)
        ctx.x_dtype, ctx.w_dtype, ctx.cd, ctx.x_shape = x.dtype, w.dtype, cd, x.shape
        # the parameter itself (identity only: gradient-sink lookup), when the weight is an f32 leaf
        ctx.weight = w if (w.is_leaf and w.dtype == torch.float32) else None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        from ...network import convstack as cs
        x2, w = ctx.saved_tensors
        N, K = w.shape
        cd = ctx.cd
        d2 = (dy if dy.dtype == cd else dy.to(cd)).reshape(-1, N)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            if K % 8:
                dx = cs.proj_dgrad(d2, w, cd, out=torch.empty(d2.shape[0], cs.rup(K, 8), dtype=cd, device=d2.device)[:, :K])
                dx = dx.reshape(ctx.x_shape).to(ctx.x_dtype)
            else:
                dx = cs.proj_dgrad(d2, w, cd).view(ctx.x_shape).to(ctx.x_dtype)
        if ctx.needs_input_grad[1] and (N % 8 or K % 8):
            # cum_gemm_tn takes widths that are multiples of 8: zero columns add nothing to the real entries
            N8, K8 = cs.rup(N, 8), cs.rup(K, 8)
            dw, _ = cs.wgrad(F.pad(d2, (0, N8 - N)), 0, N8, N8, F.pad(x2, (0, K8 - K)), 0, K8, K8, x2.shape[0],
                             want_bias=False)
            dw = dw[:N, :K].to(ctx.w_dtype)
        elif ctx.needs_input_grad[1]:
            if x2.stride(1) != 1 or x2.stride(0) % 8 or x2.data_ptr() % 16:
                x2 = x2.contiguous()
            if d2.stride(1) != 1 or d2.stride(0) % 8 or d2.data_ptr() % 16:
                d2 = d2.contiguous()
            sink = cs.grad_sink([ctx.weight]) if ctx.weight is not None else None
            if sink is not None:           # straight into the flat gradient buffer (training/flat_optim.py)
                flat, idx, offs = sink
                cs.wgrad(d2, 0, d2.stride(0), N, x2, 0, x2.stride(0), K, x2.shape[0], want_bias=False,
                         out_w=flat.grad[offs[0]:offs[0] + N * K])
                flat.wrote(idx)
            else:
                dw, _ = cs.wgrad(d2, 0, d2.stride(0), N, x2, 0, x2.stride(0), K, x2.shape[0], want_bias=False)
                dw = dw.to(ctx.w_dtype)
        return dx, dw, None


class _MambaInnerFn(torch.autograd.Function):
    """xz (B, L, 2 D) -> y (B, L, D): everything of upstream Mamba.forward between in_proj and out_proj -- split, causal
    depthwise conv + SiLU, x_proj, dt_proj, selective scan with the z gate (reached from
    src/network/CleanUMamba.py:289-290) -- as ONE autograd node on the library's kernels.  The arithmetic is that of the
    separate Functions (causal_conv1d_fn, _ProjFn, selective_scan_fn); what the single node removes is autograd's glue
    between them: the slice-backward / cat pairs of the (x | z) and (dt | B | C) splits, the float casts of B and C and
    their backward, the -exp(A_log) chain, and one AccumulateGrad add per parameter -- the kernels write dx and dz
    straight into the two halves of ONE d(xz) buffer, d(dt) / dB / dC into one d(x_dbl) buffer, and the parameter
    gradients into the flat gradient buffer (training/flat_optim.py) when it is fresh."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda")
    def forward(ctx, xz, conv_w, conv_b, xw, dtw, dt_bias, A_log, Dp, cd, save):
        import ctypes
        from ...causal_conv1d import _shape as conv_shape
        from ...network import convstack as cs
        from ..ops import selective_scan_interface as ssi
        lib = hip.lib()
        xz = xz if xz.dtype == cd else xz.to(cd)
        if not xz.is_contiguous():
            xz = xz.contiguous()
        Bn, L, D2 = xz.shape
        Dn = D2 // 2
        M = Bn * L
        R = dtw.shape[1]
        N = (xw.shape[0] - R) // 2
        dev = xz.device
        xv, zv = xz[..., :Dn].transpose(1, 2), xz[..., Dn:].transpose(1, 2)        # (B, D, L) views, channel stride 1
        w2 = conv_w.detach().reshape(Dn, -1).contiguous().float()
        cb = conv_b.detach().float().contiguous()
        xc = torch.empty(Bn, L, Dn, dtype=cd, device=dev)
        xcT = xc.transpose(1, 2)
        sc = conv_shape(xv, xcT, w2.shape[1], True)
        with torch.cuda.device(dev):
            hip.check(lib.cum_causal_conv1d_fwd(ctypes.byref(sc), hip.ptr(xv), hip.ptr(w2), hip.ptr(cb), hip.ptr(xcT),
                                                hip.stream_ptr()))
        x_dbl = cs.proj_fwd(xc.view(M, Dn), xw, cd)                                    # (M, R + 2 N)
        dt = cs.proj_fwd(x_dbl[:, :R], dtw, cd)                                        # (M, D); the bias goes in the scan
        bc = x_dbl[:, R:].float().view(Bn, L, 2 * N)                                   # B | C in f32, as the scan reads them
        Bm, Cm = bc[..., :N].transpose(1, 2), bc[..., N:].transpose(1, 2)
        A = -torch.exp(A_log.detach().float())
        Df, bias = Dp.detach().float().contiguous(), dt_bias.detach().float().contiguous()
        dtT = dt.view(Bn, L, Dn).transpose(1, 2)
        y = torch.empty(Bn, L, Dn, dtype=cd, device=dev)
        yT = y.transpose(1, 2)
        ckpt = None
        if save:
            ckpt = torch.empty(max(lib.cum_scan_ckpt_elems(Bn, Dn, N, L), 1), dtype=torch.float32, device=dev)
        ss = ssi._shape(xcT, dtT, zv, yT, Bm, Cm, True)
        # y before the gate, kept for the backward where the forward kernel can (the E6 / E8 bottleneck): the backward scan
        # then reads it instead of rebuilding it
        ypre = torch.empty_like(y) if (save and ssi.keeps_y(ss, ssi.TIME_PARALLEL)) else None
        ssi.scan_forward(ss, xcT, dtT, A, Bm, Cm, Df, zv, bias, yT, None, ckpt, ssi.TIME_PARALLEL,
                         y_pre=ypre.transpose(1, 2) if ypre is not None else None)
        ctx.save_for_backward(xz, w2, cb, xc, x_dbl, dt, bc, A, Df, bias, ckpt, xw, dtw, ypre)
        ctx.params = (conv_w, conv_b, xw, dtw, dt_bias, A_log, Dp)
        ctx.cd, ctx.dims = cd, (Bn, L, Dn, N, R)
        return y

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, dy):
        import ctypes
        from ...causal_conv1d import _shape as conv_shape
        from ...network import convstack as cs
        from ..ops import selective_scan_interface as ssi
        xz, w2, cb, xc, x_dbl, dt, bc, A, Df, bias, ckpt, xw, dtw, ypre = ctx.saved_tensors
        if ckpt is None:
            raise RuntimeError("Mamba inner backward called but the forward saved no scan checkpoints")
        conv_w, conv_b, xw_p, dtw_p, dt_bias_p, A_log_p, D_p = ctx.params
        cd = ctx.cd
        Bn, L, Dn, N, R = ctx.dims
        M, S = Bn * L, R + 2 * N
        dev = xz.device
        lib = hip.lib()
        dy = dy if dy.dtype == cd else dy.to(cd)
        dyT = (dy if dy.is_contiguous() else dy.contiguous()).transpose(1, 2)
        xv, zv = xz[..., :Dn].transpose(1, 2), xz[..., Dn:].transpose(1, 2)
        xcT, dtT = xc.transpose(1, 2), dt.view(Bn, L, Dn).transpose(1, 2)
        Bm, Cm = bc[..., :N].transpose(1, 2), bc[..., N:].transpose(1, 2)
        dxz = torch.empty_like(xz)
        dzT = dxz[..., Dn:].transpose(1, 2)
        du = torch.empty(Bn, L, Dn, dtype=cd, device=dev)
        ddelta = torch.empty(Bn, L, Dn, dtype=cd, device=dev)
        duT, ddT = du.transpose(1, 2), ddelta.transpose(1, 2)
        dBC = torch.empty(2, Bn, L, N, dtype=torch.float32, device=dev)
        # parameter gradients: straight into the flat gradient buffer where it is fresh (training/flat_optim.py)
        small = [conv_w, conv_b, dt_bias_p, A_log_p, D_p]
        sink = cs.grad_sink(small) if all(p.is_leaf and p.dtype == torch.float32 for p in small) else None

        def slot(i, like):
            if sink is not None:
                flat, _, offs = sink
                return flat.grad[offs[i]:offs[i] + like.numel()].view(like.shape)
            return torch.empty(like.shape, dtype=torch.float32, device=dev)
        dcw, dcb, dbias, dA_log, dD = slot(0, w2), slot(1, cb), slot(2, bias), slot(3, A), slot(4, Df)
        dA = torch.empty_like(A)
        bwd, ws = ssi.scan_backward_entry(Bn, Dn, N, L, dev)
        su = ssi._shape(xcT, dtT, zv, dyT, Bm, Cm, True)                  # o_* strides := dout's
        gs = hip.ScanGradStrides()
        gs.du_sb, gs.du_sd, gs.du_sl = duT.stride()
        gs.dd_sb, gs.dd_sd, gs.dd_sl = ddT.stride()
        gs.dz_sb, gs.dz_sd, gs.dz_sl = dzT.stride()
        with torch.cuda.device(dev):
            hip.check(bwd(ctypes.byref(su), ctypes.byref(gs), hip.ptr(xcT), hip.ptr(dtT), hip.ptr(A),
                          hip.ptr(Bm), hip.ptr(Cm), hip.ptr(Df), hip.ptr(zv), hip.ptr(bias),
                          hip.ptr(dyT), hip.ptr(ypre.transpose(1, 2) if ypre is not None else None),
                          hip.ptr(ckpt), hip.ptr(duT), hip.ptr(ddT), hip.ptr(dA),
                          hip.ptr(dBC[0]), hip.ptr(dBC[1]), hip.ptr(dD), hip.ptr(dzT), hip.ptr(dbias),
                          hip.ptr(ws), hip.stream_ptr()))
        torch.mul(dA, A, out=dA_log)                                       # A = -exp(A_log): dA / dA_log = A
        # d(x_dbl) = (d dt | dB | dC): one buffer, rows readable 64 columns past their end (zero weight columns there)
        pad = cs.rup(S, cs.bk_of(cd)) - S
        dxd_flat = torch.empty(M * S + pad, dtype=cd, device=dev)
        if pad:
            dxd_flat[M * S:].zero_()
        dxd = dxd_flat[:M * S].view(M, S)
        cs.proj_dgrad(ddelta.view(M, Dn), dtw, cd, out=dxd[:, :R])                       # d dt = d delta W_dt
        dxd[:, R:].view(M, 2, N).copy_(dBC.view(2, M, N).transpose(0, 1))               # dB | dC, cast
        # weight gradients of the two projections (row-split GEMM, csrc/gemm_tn.hip)
        d_xw = d_dtw = None
        for w_p, w_s, dz2, x2 in ((dtw_p, dtw, ddelta.view(M, Dn), x_dbl[:, :R]), (xw_p, xw, dxd, xc.view(M, Dn))):
            Nw, Kw = w_s.shape
            sk = cs.grad_sink([w_p]) if (w_p.is_leaf and w_p.dtype == torch.float32) else None
            if sk is not None:
                flat, idx, offs = sk
                cs.wgrad(dz2, 0, dz2.stride(0), Nw, x2, 0, x2.stride(0), Kw, M, want_bias=False,
                         out_w=flat.grad[offs[0]:offs[0] + Nw * Kw])
                flat.wrote(idx)
            else:
                g, _ = cs.wgrad(dz2, 0, dz2.stride(0), Nw, x2, 0, x2.stride(0), Kw, M, want_bias=False)
                if w_p is dtw_p:
                    d_dtw = g.to(w_p.dtype)
                else:
                    d_xw = g.to(w_p.dtype)
        # d(conv output) = d(x_dbl) W_x + du, then the depthwise conv's backward writes dx into the first half of d(xz)
        dxc = cs.proj_dgrad(dxd, xw, cd, res=du.view(M, Dn), tail_ok=True)
        dxcT = dxc.view(Bn, L, Dn).transpose(1, 2)
        dxT = dxz[..., :Dn].transpose(1, 2)
        wsc = torch.empty(max(lib.cum_conv_bwd_workspace_elems(Bn, Dn, L, w2.shape[1]), 1), dtype=torch.float32, device=dev)
        sc = conv_shape(xv, dxcT, w2.shape[1], True)
        with torch.cuda.device(dev):
            hip.check(lib.cum_causal_conv1d_bwd(ctypes.byref(sc), hip.ptr(xv), hip.ptr(w2), hip.ptr(cb), hip.ptr(dxcT),
                                                hip.ptr(dxT), dxT.stride(0), dxT.stride(1), dxT.stride(2), hip.ptr(dcw),
                                                hip.ptr(dcb), hip.ptr(wsc), hip.stream_ptr()))
        if sink is not None:
            sink[0].wrote(sink[1])
            return dxz, None, None, d_xw, d_dtw, None, None, None, None, None
        return (dxz, dcw.view(conv_w.shape).to(conv_w.dtype), dcb.to(conv_b.dtype), d_xw, d_dtw, dbias.to(dt_bias_p.dtype),
                dA_log.to(A_log_p.dtype), dD.to(D_p.dtype), None, None)


class _SplitXZ(torch.autograd.Function):
    """xz (B, L, 2D) -> x, z as (B, D, L) views.  Plain slicing leaves autograd two zero-filled (B, L, 2D) buffers, two
    slice copies and an add per block; the two gradients are simply concatenated here."""

    @staticmethod
    def forward(ctx, xz, d):
        return xz[..., :d].transpose(1, 2), xz[..., d:].transpose(1, 2)

    @staticmethod
    def backward(ctx, dx, dz):
        return torch.cat([dx.transpose(1, 2), dz.transpose(1, 2)], dim=-1), None


_FUSED_INNER = os.environ.get("CUM_FUSED_INNER", "1") != "0"  # "0": conv / projections / scan as separate autograd nodes
_FUSED_STEP = os.environ.get("CUM_FUSED_STEP", "1") != "0"  # "0": Block + Mamba.step as separate small kernels


def _proj(x, w, bias=None):
    """F.linear(x, w) of a bias-free projection on the library's GEMM kernels (see _ProjFn): training and inference,
    f32 and autocast, any channel counts (the pruned checkpoints' d_model 55, 114, 477 ...: operands and results are padded
    to 16-byte rows).  Only a projection WITH a bias (never built by the reference's configs) stays on F.linear; the
    per-token streaming step has its own kernels (csrc/hop.hip, cum_mamba_step).
    The choice must not depend on the number of rows: two f32 implementations of one GEMM differ in the last bit, which
    flips enough ReLU gates downstream to move end-to-end gradients by 1e-3 -- a 2-rank run and its single-process
    twin would no longer agree (tools/debug_batch_invariance.py)."""
    if bias is None and x.is_cuda:
        cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else x.dtype
        if cd in hip.IO_TYPES and (cd in hip.HALF_TYPES or w.dtype == torch.float32) and x.dtype in hip.IO_TYPES:
            return _ProjFn.apply(x, w, cd)
    return F.linear(x, w, bias)


class Mamba(nn.Module):
    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, conv_bias=True, bias=False,
                 use_fast_path=True, layer_idx=None, device=None, dtype=None):
        factory_kwargs = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model = d_model
        self.d_state = d_state
        self.d_conv = d_conv
        self.expand = expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **factory_kwargs)
        self.conv1d = nn.Conv1d(in_channels=self.d_inner, out_channels=self.d_inner, bias=conv_bias,
                                kernel_size=d_conv, groups=self.d_inner, padding=d_conv - 1, **factory_kwargs)
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **factory_kwargs)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **factory_kwargs)

        # dt_proj init: weight ~ U(+-dt_rank^-0.5 * dt_scale); bias = softplus^-1(dt), dt log-uniform in
        # [dt_min, dt_max]  
        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        self.dt_proj.bias._no_reinit = True

        A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device)[None, :]
        self.A_log = nn.Parameter(torch.log(A.repeat(self.d_inner, 1).contiguous()))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **factory_kwargs)

    def forward(self, hidden_states, inference_params=None):
        """hidden_states: (B, L, d_model) -> (B, L, d_model)."""
        batch, seqlen, _ = hidden_states.shape
        conv_state, ssm_state = None, None
        if inference_params is not None:
            conv_state, ssm_state = self._get_states_from_cache(inference_params, batch)
            if inference_params.seqlen_offset > 0:
                out, _, _ = self.step(hidden_states, conv_state, ssm_state)
                return out

        d_inner = self.in_proj.weight.shape[0] // 2
        dt_rank = self.dt_proj.weight.shape[1]
        d_state = (self.x_proj.weight.shape[0] - dt_rank) // 2
        d_conv = self.conv1d.weight.shape[-1]

        xz = _proj(hidden_states, self.in_proj.weight, self.in_proj.bias)         # (B, L, 2 d_inner)
        if (_FUSED_INNER and xz.is_cuda and conv_state is None and ssm_state is None and causal_conv1d_fn is not None
                and d_conv <= 4 and self.conv1d.bias is not None and self.x_proj.weight.shape[0] % 8 == 0
                and dt_rank % 8 == 0 and d_inner % 8 == 0 and self.dt_proj.weight.dtype == torch.float32):
            cd = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else xz.dtype
            if cd in hip.IO_TYPES:
                params = (self.conv1d.weight, self.conv1d.bias, self.x_proj.weight, self.dt_proj.weight, self.dt_proj.bias,
                          self.A_log, self.D)
                save = torch.is_grad_enabled() and (xz.requires_grad or any(p.requires_grad for p in params))
                y = _MambaInnerFn.apply(xz, *params, cd, save)                             # (B, L, d_inner)
                return _proj(y, self.out_proj.weight, self.out_proj.bias)
        x, z = _SplitXZ.apply(xz, d_inner)                                         # (B, d_inner, L) views, channel stride 1
        A = -torch.exp(self.A_log.float())
        if conv_state is not None:
            conv_state.copy_(F.pad(x, (d_conv - x.shape[-1], 0)))
        if causal_conv1d_fn is None:
            x = self.act(self.conv1d(x)[..., :seqlen])
            x = x.transpose(1, 2).contiguous().transpose(1, 2)
        else:
            x = causal_conv1d_fn(x, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x_dbl = _proj(x.transpose(1, 2), self.x_proj.weight)                      # (B, L, R + 2N)
        dt, Bm, Cm = torch.split(x_dbl, [dt_rank, d_state, d_state], dim=-1)
        dt = _proj(dt, self.dt_proj.weight).transpose(1, 2)                       # (B, d_inner, L); bias goes in the scan
        y = selective_scan_fn(x, dt, A, Bm.transpose(1, 2), Cm.transpose(1, 2), self.D.float(), z=z,
                              delta_bias=self.dt_proj.bias.float(), delta_softplus=True,
                              return_last_state=ssm_state is not None)
        if ssm_state is not None:
            y, last_state = y
            ssm_state.copy_(last_state)
        return _proj(y.transpose(1, 2), self.out_proj.weight, self.out_proj.bias)

    def _neg_exp_A_log(self):
        """A = -exp(A_log); in inference (no grad) it is computed once per value of A_log instead of once per token
        (two tiny kernels per block and hop of the streaming path)."""
        if torch.is_grad_enabled() and self.A_log.requires_grad:
            return -torch.exp(self.A_log.float())
        key = (self.A_log._version, self.A_log.data_ptr(), self.A_log.device)
        hit = self.__dict__.get("_A_cache")
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, -torch.exp(self.A_log.float()))
            self.__dict__["_A_cache"] = hit
        return hit[1]

    def step(self, hidden_states, conv_state, ssm_state):
        """One token for every stream.  hidden_states: (B, 1, d_model); states updated in place."""
        assert hidden_states.shape[1] == 1, "step() decodes one token at a time"
        dt_rank = self.dt_proj.weight.shape[1]
        d_state = (self.x_proj.weight.shape[0] - dt_rank) // 2
        xz = _proj(hidden_states.squeeze(1), self.in_proj.weight, self.in_proj.bias)
        x, z = xz.chunk(2, dim=-1)
        x = causal_conv1d_update(x.float(), conv_state, self.conv1d.weight.squeeze(1).float(),
                                 None if self.conv1d.bias is None else self.conv1d.bias.float(), self.activation)
        x_db = _proj(x, self.x_proj.weight)
        dt, Bv, Cv = torch.split(x_db, [dt_rank, d_state, d_state], dim=-1)
        dt = _proj(dt, self.dt_proj.weight)
        A = self._neg_exp_A_log()
        y = selective_state_update(ssm_state, x, dt.float(), A, Bv, Cv, self.D.float(), z=z.float(),
                                   dt_bias=self.dt_proj.bias.float(), dt_softplus=True)
        out = _proj(y.to(hidden_states.dtype), self.out_proj.weight, self.out_proj.bias)
        return out.unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        device = self.out_proj.weight.device
        d_inner = self.in_proj.weight.shape[0] // 2
        conv_state = torch.zeros(batch_size, d_inner, self.conv1d.weight.shape[-1], device=device,
                                 dtype=torch.float32)
        ssm_state = torch.zeros(batch_size, d_inner, self.A_log.shape[1], device=device, dtype=torch.float32)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        assert self.layer_idx is not None
        if self.layer_idx not in inference_params.key_value_memory_dict:
            inference_params.key_value_memory_dict[self.layer_idx] = self.allocate_inference_cache(batch_size, 1)
        conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
        if initialize_states:
            conv_state.zero_()
            ssm_state.zero_()
        return conv_state, ssm_state


class Block(nn.Module):
    """Pre-norm residual block, non-fused path (the only one the reference executes:
    fused_add_norm=False at src/network/CleanUMamba.py:43,156).  ``mixer`` is registered
    before ``norm`` as in mamba-ssm 1.2.2."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False, residual_in_fp32=False):
        super().__init__()
        if fused_add_norm:
            raise NotImplementedError("fused_add_norm=True (Triton layer norm) is not on the reference's path")
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)

    def _fused_step_ok(self, hidden_states, residual, inference_params):
        m = self.mixer
        if not (_FUSED_STEP and inference_params is not None and inference_params.seqlen_offset > 0
                and hidden_states.is_cuda and hidden_states.shape[1] == 1 and hidden_states.dtype == torch.float32
                and not torch.is_grad_enabled() and isinstance(m, Mamba) and isinstance(self.norm, nn.LayerNorm)
                and self.norm.elementwise_affine and m.in_proj.weight.dtype == torch.float32
                and m.activation in ("silu", "swish") and (residual is None or residual.dtype == torch.float32)):
            return False
        d_inner, dt_rank = m.in_proj.weight.shape[0] // 2, m.dt_proj.weight.shape[1]
        d_state = (m.x_proj.weight.shape[0] - dt_rank) // 2
        return bool(hip.lib().cum_mamba_step_supported(hidden_states.shape[-1], d_inner, d_state, dt_rank,
                                                       m.conv1d.weight.shape[-1]))

    def _fused_step(self, hidden_states, residual, inference_params):
        """Block.forward + Mamba.step for one token of every stream in one launch (csrc/mamba_step.hip)."""
        m = self.mixer
        bsz, _, dm = hidden_states.shape
        conv_state, ssm_state = m._get_states_from_cache(inference_params, bsz)
        d_inner, dt_rank = m.in_proj.weight.shape[0] // 2, m.dt_proj.weight.shape[1]
        d_state = (m.x_proj.weight.shape[0] - dt_rank) // 2
        d_conv = m.conv1d.weight.shape[-1]
        h = hidden_states.reshape(bsz, dm).contiguous()
        r = None if residual is None else residual.reshape(bsz, dm).contiguous()
        out, res_out = torch.empty_like(h), torch.empty_like(h)
        keep = []        # contiguous copies must outlive the launch: ctypes passes bare addresses

        def c(t):
            if t is None:
                return None
            keep.append(t.detach().contiguous())
            return keep[-1]
        with torch.cuda.device(h.device):
            hip.check(hip.lib().cum_mamba_step(
                bsz, dm, d_inner, d_state, dt_rank, d_conv, float(self.norm.eps), hip.ptr(h), hip.ptr(r),
                hip.ptr(c(self.norm.weight)), hip.ptr(c(self.norm.bias)), hip.ptr(c(m.in_proj.weight)),
                hip.ptr(c(m.in_proj.bias)), hip.ptr(conv_state), hip.ptr(c(m.conv1d.weight).view(d_inner, d_conv)),
                hip.ptr(c(m.conv1d.bias)), hip.ptr(c(m.x_proj.weight)), hip.ptr(c(m.dt_proj.weight)),
                hip.ptr(c(m.dt_proj.bias)), hip.ptr(c(m._neg_exp_A_log())), hip.ptr(c(m.D)), hip.ptr(ssm_state),
                hip.ptr(c(m.out_proj.weight)), hip.ptr(c(m.out_proj.bias)), hip.ptr(out), hip.ptr(res_out),
                hip.stream_ptr()))
        del keep
        return out.view(bsz, 1, dm), res_out.view(bsz, 1, dm)

    def forward(self, hidden_states, residual=None, inference_params=None):
        if self._fused_step_ok(hidden_states, residual, inference_params):
            return self._fused_step(hidden_states, residual, inference_params)
        if self.residual_in_fp32 and _ln.supported(hidden_states, self.norm):
            # add + LayerNorm in one kernel each way (csrc/layernorm.hip); same arithmetic as the three lines below
            hidden_states, residual = _ln.add_layer_norm(hidden_states, residual, self.norm)
        else:
            residual = (hidden_states + residual) if residual is not None else hidden_states
            hidden_states = self.norm(residual.to(dtype=self.norm.weight.dtype))
            if self.residual_in_fp32:
                residual = residual.to(torch.float32)
        hidden_states = self.mixer(hidden_states, inference_params=inference_params)
        return hidden_states, residual

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)
