"""Oracle (test infrastructure): pure-torch restatement of the Mamba block the
reference obtains from ``mamba-ssm==1.2.2`` / ``causal-conv1d==1.1.0``.

Those packages are NOT under /root/reference (``environment.yml:29-30`` pins
them); what is restated here follows their published algorithm and is anchored
on the reference's own call sites and in-repo mirror:

* ``create_block`` call site ............ src/network/CleanUMamba.py:172-189
* ``block(hidden, residual, inference_params=...)`` .. CleanUMamba.py:289-290, 451-454
* ``InferenceParams(max_seqlen, max_batch_size, key_value_memory_dict, seqlen_offset)``
  ........................................ CleanUMamba.py:374-381
* constructor / forward structure of the mixer: the near-verbatim copy kept in
  src/network/S4/MambaS4.py:367-473 (in_proj layout, conv padding d_conv-1 and
  crop ``[..., :seqlen]`` at :455, xz.chunk at :447) and ``create_block_mamba_s4``
  at :475-503 (Block wiring).
* attribute names used by the loader ..... CleanUMamba.py:336-349, 540-545

PARITY UNPINNED against the upstream CUDA kernels (no golden vectors exist in the
reference); cross-checked in tests against HF transformers' independent
``mamba_selective_scan`` / ``MambaMixer``.

Everything runs in the dtype it is given (fp32 for fixtures, fp64 for the
high-precision recompute) and is differentiable through autograd, which is how
backward fixtures are produced.
"""
import math
from dataclasses import dataclass, field
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------- ops
def softplus_thr20(x):
    """softplus with the kernel's threshold: x <= 20 ? log1p(exp(x)) : x."""
    return torch.where(x <= 20.0, torch.log1p(torch.exp(torch.clamp(x, max=20.0))), x)


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                       delta_softplus=False, return_last_state=False):
    """Sequential selective scan.

    u, delta, z: (B, D, L); A: (D, N); B, C: (B, N, L); D, delta_bias: (D,).
    x_t = exp(delta_t A) x_{t-1} + delta_t B_t u_t ; y_t = <C_t, x_t> + D u_t ;
    out = y * silu(z).  Math in the input dtype promoted to at least fp32.
    """
    dtype_in = u.dtype
    ct = torch.float64 if u.dtype == torch.float64 else torch.float32
    u_, delta_ = u.to(ct), delta.to(ct)
    if delta_bias is not None:
        delta_ = delta_ + delta_bias.to(ct)[None, :, None]
    if delta_softplus:
        delta_ = softplus_thr20(delta_)
    A_, B_, C_ = A.to(ct), B.to(ct), C.to(ct)
    bsz, dim, L = u_.shape
    N = A_.shape[1]
    x = u_.new_zeros(bsz, dim, N)
    ys = []
    for t in range(L):
        dt = delta_[:, :, t]                                    # (B, D)
        a = torch.exp(dt[:, :, None] * A_[None])                # (B, D, N)
        b = (dt * u_[:, :, t])[:, :, None] * B_[:, None, :, t]  # (B, D, N)
        x = a * x + b
        ys.append((x * C_[:, None, :, t]).sum(-1))
    y = torch.stack(ys, dim=2) if L > 0 else u_.new_zeros(bsz, dim, 0)
    if D is not None:
        y = y + u_ * D.to(ct)[None, :, None]
    if z is not None:
        z_ = z.to(ct)
        y = y * (z_ * torch.sigmoid(z_))
    y = y.to(dtype_in)
    return (y, x) if return_last_state else y


def causal_conv1d_ref(x, weight, bias=None, activation=None):
    """y[b,d,t] = act(bias_d + sum_k w[d,k] x[b,d,t-(W-1)+k]), zero left pad.

    Same as ``act(conv1d(x, padding=W-1)[..., :L])`` -- src/network/S4/MambaS4.py:455.
    x: (B, D, L); weight: (D, W)."""
    if activation not in (None, "silu", "swish"):
        raise NotImplementedError("activation must be None, silu, or swish")
    D, W = weight.shape
    L = x.shape[-1]
    out = F.conv1d(x, weight.unsqueeze(1), bias, padding=W - 1, groups=D)[..., :L]
    return out if activation is None else F.silu(out)


def causal_conv1d_update_ref(x, conv_state, weight, bias=None, activation=None):
    """One step: roll conv_state (B, D, W) left, append x (B, D), dot with weight."""
    conv_state.copy_(torch.roll(conv_state, shifts=-1, dims=-1))
    conv_state[:, :, -1] = x
    out = torch.sum(conv_state * weight[None], dim=-1)
    if bias is not None:
        out = out + bias
    return out if activation is None else F.silu(out)


def selective_state_update_ref(state, x, dt, A, B, C, D=None, z=None, dt_bias=None,
                               dt_softplus=False):
    """One time step of the scan; state (B, D, N) updated in place.
    x, dt, z: (B, D); A: (D, N); B, C: (B, N)."""
    if dt_bias is not None:
        dt = dt + dt_bias
    if dt_softplus:
        dt = softplus_thr20(dt)
    dA = torch.exp(dt[:, :, None] * A[None])
    dB = dt[:, :, None] * B[:, None, :]
    state.copy_(state * dA + dB * x[:, :, None])
    out = (state * C[:, None, :]).sum(-1)
    if D is not None:
        out = out + x * D
    if z is not None:
        out = out * F.silu(z)
    return out


# ------------------------------------------------------------------------ modules
@dataclass
class InferenceParams:
    """Field-compatible with mamba_ssm.utils.generation.InferenceParams as the
    reference constructs it (CleanUMamba.py:376-381)."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: object = None


class Mamba(nn.Module):
    """Mixer; class name must be "Mamba" (CleanUMamba.py:540)."""

    def __init__(self, d_model, d_state=16, d_conv=4, expand=2, dt_rank="auto",
                 dt_min=0.001, dt_max=0.1, dt_init="random", dt_scale=1.0,
                 dt_init_floor=1e-4, conv_bias=True, bias=False, use_fast_path=True,
                 layer_idx=None, device=None, dtype=None):
        fk = {"device": device, "dtype": dtype}
        super().__init__()
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(self.expand * self.d_model)
        self.dt_rank = math.ceil(self.d_model / 16) if dt_rank == "auto" else dt_rank
        self.use_fast_path = use_fast_path
        self.layer_idx = layer_idx

        self.in_proj = nn.Linear(self.d_model, self.d_inner * 2, bias=bias, **fk)
        self.conv1d = nn.Conv1d(self.d_inner, self.d_inner, bias=conv_bias, kernel_size=d_conv,
                                groups=self.d_inner, padding=d_conv - 1, **fk)
        self.activation = "silu"
        self.act = nn.SiLU()
        self.x_proj = nn.Linear(self.d_inner, self.dt_rank + self.d_state * 2, bias=False, **fk)
        self.dt_proj = nn.Linear(self.dt_rank, self.d_inner, bias=True, **fk)

        dt_init_std = self.dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(self.dt_proj.weight, dt_init_std)
        elif dt_init == "random":
            nn.init.uniform_(self.dt_proj.weight, -dt_init_std, dt_init_std)
        else:
            raise NotImplementedError
        dt = torch.exp(torch.rand(self.d_inner, **fk) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        inv_dt = dt + torch.log(-torch.expm1(-dt))     # inverse softplus
        with torch.no_grad():
            self.dt_proj.bias.copy_(inv_dt)
        self.dt_proj.bias._no_reinit = True

        A = torch.arange(1, self.d_state + 1, dtype=torch.float32, device=device)
        A = A[None, :].repeat(self.d_inner, 1).contiguous()
        self.A_log = nn.Parameter(torch.log(A))
        self.A_log._no_weight_decay = True
        self.D = nn.Parameter(torch.ones(self.d_inner, device=device))
        self.D._no_weight_decay = True
        self.out_proj = nn.Linear(self.d_inner, self.d_model, bias=bias, **fk)

    def forward(self, hidden_states, inference_params=None):
        batch, seqlen, _ = hidden_states.shape
        conv_state, ssm_state = None, None
        if inference_params is not None:
            conv_state, ssm_state = self._get_states_from_cache(inference_params, batch)
            if inference_params.seqlen_offset > 0:
                out, _, _ = self.step(hidden_states, conv_state, ssm_state)
                return out
        # "b l d -> d (b l)" matmul then "d (b l) -> b d l"   (MambaS4.py:440-444)
        xz = (self.in_proj.weight @ hidden_states.reshape(batch * seqlen, -1).t())
        xz = xz.reshape(-1, batch, seqlen).permute(1, 0, 2)
        if self.in_proj.bias is not None:
            xz = xz + self.in_proj.bias.to(xz.dtype)[None, :, None]
        A = -torch.exp(self.A_log.float())
        x, z = xz.chunk(2, dim=1)
        if conv_state is not None:
            conv_state.copy_(F.pad(x, (self.d_conv - x.shape[-1], 0)))
        x = causal_conv1d_ref(x, self.conv1d.weight.squeeze(1), self.conv1d.bias, self.activation)
        x_dbl = self.x_proj(x.permute(0, 2, 1).reshape(batch * seqlen, -1))
        dt, B, C = torch.split(x_dbl, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = (self.dt_proj.weight @ dt.t()).reshape(-1, batch, seqlen).permute(1, 0, 2)
        B = B.reshape(batch, seqlen, -1).permute(0, 2, 1).contiguous()
        C = C.reshape(batch, seqlen, -1).permute(0, 2, 1).contiguous()
        y = selective_scan_ref(x, dt, A, B, C, self.D.float(), z=z,
                               delta_bias=self.dt_proj.bias.float(), delta_softplus=True,
                               return_last_state=ssm_state is not None)
        if ssm_state is not None:
            y, last_state = y
            ssm_state.copy_(last_state)
        return self.out_proj(y.permute(0, 2, 1))

    def step(self, hidden_states, conv_state, ssm_state):
        assert hidden_states.shape[1] == 1
        xz = self.in_proj(hidden_states.squeeze(1))
        x, z = xz.chunk(2, dim=-1)
        x = causal_conv1d_update_ref(x, conv_state, self.conv1d.weight.squeeze(1),
                                     self.conv1d.bias, self.activation)
        x_db = self.x_proj(x)
        dt, B, C = torch.split(x_db, [self.dt_rank, self.d_state, self.d_state], dim=-1)
        dt = F.linear(dt, self.dt_proj.weight)
        A = -torch.exp(self.A_log.float())
        y = selective_state_update_ref(ssm_state, x, dt, A, B, C, self.D, z=z,
                                       dt_bias=self.dt_proj.bias, dt_softplus=True)
        out = self.out_proj(y)
        return out.unsqueeze(1), conv_state, ssm_state

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        device = self.out_proj.weight.device
        conv_dtype = self.conv1d.weight.dtype if dtype is None else dtype
        ssm_dtype = self.dt_proj.weight.dtype if dtype is None else dtype
        conv_state = torch.zeros(batch_size, int(self.d_model * self.expand), self.d_conv,
                                 device=device, dtype=conv_dtype)
        ssm_state = torch.zeros(batch_size, int(self.d_model * self.expand), self.d_state,
                                device=device, dtype=ssm_dtype)
        return conv_state, ssm_state

    def _get_states_from_cache(self, inference_params, batch_size, initialize_states=False):
        assert self.layer_idx is not None
        if self.layer_idx not in inference_params.key_value_memory_dict:
            inference_params.key_value_memory_dict[self.layer_idx] = \
                self.allocate_inference_cache(batch_size, 1)
        conv_state, ssm_state = inference_params.key_value_memory_dict[self.layer_idx]
        if initialize_states:
            conv_state.zero_()
            ssm_state.zero_()
        return conv_state, ssm_state


class Block(nn.Module):
    """Pre-norm residual wrapper (non-fused path only); mixer registered before
    norm as in mamba-ssm 1.2.2.  Mirror: src/network/S4/MambaS4.py:494-503."""

    def __init__(self, dim, mixer_cls, norm_cls=nn.LayerNorm, fused_add_norm=False,
                 residual_in_fp32=False):
        super().__init__()
        if fused_add_norm:
            raise NotImplementedError("oracle restates the non-fused path the reference uses")
        self.residual_in_fp32 = residual_in_fp32
        self.fused_add_norm = fused_add_norm
        self.mixer = mixer_cls(dim)
        self.norm = norm_cls(dim)

    def forward(self, hidden_states, residual=None, inference_params=None):
        residual = (hidden_states + residual) if residual is not None else hidden_states
        hidden_states = self.norm(residual.to(dtype=self.norm.weight.dtype))
        if self.residual_in_fp32 and residual.dtype != torch.float64:
            residual = residual.to(torch.float32)
        hidden_states = self.mixer(hidden_states, inference_params=inference_params)
        return hidden_states, residual

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return self.mixer.allocate_inference_cache(batch_size, max_seqlen, dtype=dtype, **kwargs)


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, rms_norm=False,
                 residual_in_fp32=False, fused_add_norm=False, layer_idx=None,
                 device=None, dtype=None):
    """Signature as called at src/network/CleanUMamba.py:174-189."""
    if rms_norm:
        raise NotImplementedError("rms_norm=True is not on the reference's executed path")
    ssm_cfg = {} if ssm_cfg is None else ssm_cfg
    fk = {"device": device, "dtype": dtype}
    mixer_cls = partial(Mamba, layer_idx=layer_idx, **ssm_cfg, **fk)
    norm_cls = partial(nn.LayerNorm, eps=norm_epsilon, **fk)
    block = Block(d_model, mixer_cls, norm_cls=norm_cls, fused_add_norm=fused_add_norm,
                  residual_in_fp32=residual_in_fp32)
    block.layer_idx = layer_idx
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True,
                  n_residuals_per_layer=1):
    """Applied via ``self.apply(partial(_init_weights, n_layer=...))`` -- CleanUMamba.py:201-206."""
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        for name, p in module.named_parameters():
            if name in ["out_proj.weight", "fc2.weight"]:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)
