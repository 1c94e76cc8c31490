"""CleanUMamba module for MI355X -- drop-in for src/network/CleanUMamba.py:30-550.

Same constructor keywords and defaults (:33-54), same sub-module tree and therefore
the same state-dict keys and shapes, same public methods
(forward, feed, flush, load_pruned_state_dict, valid_length, pad_signal,
total_stride, frame_length, time_per_frame, reset_time_per_frame,
allocate_inference_cache).  The arithmetic of the Mamba bottleneck runs in the HIP
kernels of csrc/ (selective scan, causal depthwise conv, single-step update).

Differences from the reference, on purpose:
  * ``forward`` does not mutate its argument (the reference divides the caller's
    tensor by its std in place, :262); the returned value is identical.
  * ``feed``/``flush`` implement the intended streaming semantics (stream output ==
    ``forward`` output with normalize_input=False).  The reference's own feed()
    raises on every shipped model (skip order at :474) and its flush() drops the
    decoder overlap of the tail (:364); see SURVEY.md fact 9.
  * ablation variants (LSTM, Mamba2, MambaS4, residual_projection, rms_norm,
    fused_add_norm) are out of scope and raise NotImplementedError.
"""
import os
import warnings
import time
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..mamba_ssm.models.mixer_seq_simple import _init_weights, create_block
from ..mamba_ssm.ops import layernorm as _ln
from ..mamba_ssm.utils.generation import InferenceParams
from ..util.util import weight_scaling_init
from .. import hip
from . import convstack as cs
from . import hopplan
from .layers import Activation


class CleanUMamba(nn.Module):
    """CleanUNet encoder/decoder with a Mamba bottleneck."""

    def __init__(self, channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8,
                 kernel_size=4, stride=2, encoder_groups=1, bypass_channels=0, glu_activation="Sigmoid",
                 tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048, fused_add_norm=False,
                 use_fast_path=False, rms_norm=False, mamba_s4=False, LSTM=False, mamba_v2=False,
                 residual_projection=False, norm_epsilon: float = 1e-5, normalize_input=True, device=None,
                 dtype=None):
        super().__init__()
        assert glu_activation in ["Sigmoid", "ReLU", "SiLU", "GELU"], f"glu_activation={glu_activation} not supported"
        for flag, name in ((mamba_s4, "mamba_s4"), (LSTM, "LSTM"), (mamba_v2, "mamba_v2"),
                           (residual_projection, "residual_projection"), (rms_norm, "rms_norm"),
                           (fused_add_norm, "fused_add_norm")):
            if flag:
                raise NotImplementedError(f"{name}=True is an ablation variant outside the MI355X hot path")
        factory_kwargs = {"device": device, "dtype": dtype}

        self.channels_input = channels_input
        self.channels_output = channels_output
        self.channels_H = channels_H
        self.max_H = max_H
        self.encoder_n_layers = encoder_n_layers
        self.kernel_size = kernel_size
        self.stride = stride
        self.tsfm_n_layers = tsfm_n_layers
        self.tsfm_n_head = tsfm_n_head
        self.tsfm_d_model = tsfm_d_model
        self.tsfm_d_inner = tsfm_d_inner
        self.residual_projection = residual_projection
        self.normalize_input = normalize_input
        self.dtype = dtype

        self.encoder = nn.ModuleList()
        self.decoder = nn.ModuleList()
        for i in range(encoder_n_layers):
            ec_groups = encoder_groups[i] if isinstance(encoder_groups, list) else encoder_groups
            bp_channels = bypass_channels[i] if isinstance(bypass_channels, list) else bypass_channels
            self.encoder.append(nn.Sequential(
                nn.Conv1d(channels_input, channels_H, kernel_size, stride, groups=ec_groups if i > 0 else 1,
                          **factory_kwargs),
                nn.ReLU(),
                nn.Conv1d(channels_H, bp_channels + (channels_H - bp_channels) * 2, 1, **factory_kwargs),
                Activation(glu_activation, bp_channels)))
            channels_input = channels_H
            decoder_i = nn.Sequential(
                nn.Conv1d(channels_H, bp_channels + (channels_H - bp_channels) * 2, 1, **factory_kwargs),
                Activation(glu_activation, bp_channels),
                nn.ConvTranspose1d(channels_H, channels_output, kernel_size, stride, **factory_kwargs))
            if i > 0:  # ReLU on all but the outermost decoder layer
                decoder_i.append(nn.ReLU())
            self.decoder.insert(0, decoder_i)
            channels_output = channels_H
            channels_H = min(channels_H * 2, max_H)

        self.tsfm_conv1 = nn.Conv1d(channels_output, tsfm_d_model, kernel_size=1, **factory_kwargs)
        ssm_cfg = {"d_state": tsfm_d_model // tsfm_n_head, "d_conv": 4, "expand": tsfm_d_inner // tsfm_d_model,
                   "use_fast_path": use_fast_path}
        self.rms_norm = rms_norm
        self.residual_in_fp32 = True
        self.fused_add_norm = fused_add_norm
        self.LSTM = LSTM
        self.tsfm_Mamba_layers = nn.ModuleList([
            create_block(tsfm_d_model, ssm_cfg=ssm_cfg, norm_epsilon=norm_epsilon, rms_norm=rms_norm,
                         residual_in_fp32=self.residual_in_fp32, fused_add_norm=self.fused_add_norm, layer_idx=i,
                         **factory_kwargs)
            for i in range(tsfm_n_layers)])
        self.norm_f = nn.LayerNorm(tsfm_d_model, eps=norm_epsilon, **factory_kwargs)
        self.tsfm_conv2 = nn.Conv1d(tsfm_d_model, channels_output, kernel_size=1, **factory_kwargs)

        # initialisation order of the reference: weight scaling on every conv (the Mamba depthwise conv
        # included), then mamba's _init_weights on every sub-module (:197-206)
        for layer in self.modules():
            if isinstance(layer, (nn.Conv1d, nn.ConvTranspose1d)):
                weight_scaling_init(layer)
        self.apply(partial(_init_weights, n_layer=tsfm_n_layers))

        # True: encoder/decoder run on the fused HIP GEMM kernels (network/convstack.py).  False: the layers
        # are called as torch modules (needed only when forward hooks on the conv modules must fire, as the
        # reference's pruning tools expect); the Mamba bottleneck uses the HIP kernels either way.
        self.use_fused_convs = True
        # True: the encoder and the decoder are one autograd node each (cs.EncoderStack / cs.DecoderStack) whose
        # backward folds the ReLU gate, the GLU backward and the skip-gradient add into GEMM epilogues.  False: one
        # node per layer with separate elementwise kernels (CUM_STACK_BACKWARD=0 selects it for A/B timing).
        self.use_stack_backward = os.environ.get("CUM_STACK_BACKWARD", "1") != "0"
        # True: after the first hop of a stream the (launch-bound, ~100 tiny kernels) hop is captured once in a
        # hipGraph and replayed; stream state lives in static buffers updated in place.
        self.use_hop_graph = True
        # True: streaming hops run on the fused GEMM kernels (_denoise_frame_fused); False: torch modules with
        # per-layer encoder caches (_denoise_frame, the reference's structure)
        self.use_fused_stream = True
        # True: the fused streaming hop stores activations and runs its GEMMs in bf16 (f32 accumulate; the Mamba
        # steps and all stream state stay f32).  Off by default: the hop then matches ``forward`` to 1e-4.
        self.stream_bf16 = False

        # streaming state
        self.total_time = 0
        self.cat_time = 0
        self.frames = 0                 # frames denoised since reset_time_per_frame(): only for time_per_frame
        self._std_frames = 0            # frames of the CURRENT stream: denominator of the running input std
        self.input_std = 0
        self.pending = torch.zeros(self.channels_input, 0, dtype=self.dtype, device=device)
        self.frame_length = self.valid_length(1)
        self.inference_params = None
        self.encoder_decoder_state = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop("_pack_plans", None)      # device-side caches of packed weights are rebuilt on demand
        state.pop("_plist", None)
        state.pop("_wv_call", None)
        state.pop("_hop_graph", None)       # captured hipGraph of the streaming hop
        state.pop("_hop_plan", None)        # packed weights / plan of the one-launch hop
        state.pop("_hop_kernel_why", None)
        return state

    # ------------------------------------------------------------------ geometry
    def valid_length(self, length):
        """Smallest length >= ``length`` that survives D strided convs and their transposes exactly."""
        D, K, S = self.encoder_n_layers, self.kernel_size, self.stride
        for _ in range(D):
            length = 1 if length < K else 1 + np.ceil((length - K) / S)
        for _ in range(D):
            length = (length - 1) * S + K
        return int(length)

    def pad_signal(self, input):
        return F.pad(input, (0, self.valid_length(input.shape[-1]) - input.shape[-1]))

    @property
    def total_stride(self):
        return self.stride ** self.encoder_n_layers

    # ------------------------------------------------------------------- forward
    @staticmethod
    def _pointwise_linear(conv, x):
        """A 1x1 Conv1d as a matmul over (B*T, C): for the one-column inputs of a streaming hop MIOpen falls back
        to a naive kernel (23 us against 6 us).  Module hooks on ``conv`` do not fire on this route."""
        w = conv.weight.squeeze(-1)
        if x.is_cuda and x.shape[-1] == 1 and x.dtype == torch.float32 and w.dtype == torch.float32 \
                and w.shape[1] <= 1024 and not torch.is_grad_enabled():
            # one-column input: a workgroup per stream with the matrix staged in LDS (csrc/mamba_step.hip)
            xin = x.reshape(x.shape[0], x.shape[1]).contiguous()
            out = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
            # named locals: a contiguous() temporary must outlive the launch (ctypes passes bare addresses)
            wc = w.detach().contiguous()
            bc = None if conv.bias is None else conv.bias.detach().contiguous()
            with torch.cuda.device(x.device):
                hip.check(hip.lib().cum_small_linear(x.shape[0], w.shape[0], w.shape[1], hip.ptr(xin), w.shape[1],
                                                     hip.ptr(wc), hip.ptr(bc), hip.ptr(out), w.shape[0],
                                                     hip.stream_ptr()))
            return out.unsqueeze(-1)
        return F.linear(x.transpose(1, 2), w, conv.bias).transpose(1, 2)

    def _bottleneck(self, x, inference_params=None, pointwise_as_linear=False):
        """tsfm_conv1 -> Mamba blocks -> add + norm_f -> tsfm_conv2.  x: (B, C, T)."""
        if pointwise_as_linear:
            x = self._pointwise_linear(self.tsfm_conv1, x)
        else:
            x = self.tsfm_conv1(x)
        hidden_states = x.permute(0, 2, 1)
        residual = None
        for layer in self.tsfm_Mamba_layers:
            hidden_states, residual = layer(hidden_states, residual, inference_params=inference_params)
        if _ln.supported(hidden_states, self.norm_f):
            hidden_states, _ = _ln.add_layer_norm(hidden_states, residual, self.norm_f)
        else:
            residual = (hidden_states + residual) if residual is not None else hidden_states
            hidden_states = self.norm_f(residual.to(dtype=self.norm_f.weight.dtype))
        tsfm_out = hidden_states.permute(0, 2, 1)
        if pointwise_as_linear:
            return self._pointwise_linear(self.tsfm_conv2, tsfm_out), tsfm_out
        return self.tsfm_conv2(tsfm_out), tsfm_out

    def forward(self, noisy_audio, return_skip_connections=False):
        if noisy_audio.dim() == 2:
            noisy_audio = noisy_audio.unsqueeze(1)
        B, C, L = noisy_audio.shape
        assert C == 1
        if B == 0 and not return_skip_connections:          # an empty batch: the torch modules return an empty result too
            return noisy_audio.new_zeros(0, self.channels_output, L)
        fused = getattr(self, "use_fused_convs", True) and noisy_audio.is_cuda
        if fused and not cs.supported(self):
            raise NotImplementedError("fused conv stack covers kernel 4 / stride 2 / ungrouped / sigmoid-GLU "
                                      "layers; set model.use_fused_convs = False for other variants")
        if fused and noisy_audio.dtype == torch.float32 and not noisy_audio.requires_grad and self.channels_output == 1:
            # waveform ends on the library's own kernels (csrc/loss.hip): per-clip std, `noisy / std` + padding written
            # straight into the first conv's row buffer, `x[:, :, :L] * std` read straight from the last one's
            std = cs.clip_std(noisy_audio, 1e-3) if self.normalize_input else None
            T0 = self.valid_length(L)
            dt = self._fused_dtype()
            buf = cs.frame_input(noisy_audio, std, T0, dt)
            if torch.is_grad_enabled():
                buf, geo, skip_connections, tsfm_out = self._forward_fused(buf, B, T0, dt)
            else:
                with cs.small_m_gemms():     # inference on short inputs: few-tile GEMMs may split K over the waves
                    buf, geo, skip_connections, tsfm_out = self._forward_fused(buf, B, T0, dt)
            # (without input normalisation the reference returns the padded length: src/network/CleanUMamba.py:318-319)
            x = cs.Unframe.apply(buf, std, geo, L if self.normalize_input else T0)
            if return_skip_connections:
                skip_connections.append(tsfm_out)
                return x, skip_connections
            return x
        if self.normalize_input:
            std = noisy_audio.std(dim=2, keepdim=True) + 1e-3
            noisy_audio = noisy_audio / std
        x = self.pad_signal(noisy_audio)

        if fused:
            dt = self._fused_dtype()
            geo = cs.Geo(B, x.shape[-1], 1)
            buf = cs.to_rows(x, geo, dt)
            if torch.is_grad_enabled():
                buf, geo, skip_connections, tsfm_out = self._forward_fused(buf, B, x.shape[-1], dt)
            else:
                with cs.small_m_gemms():
                    buf, geo, skip_connections, tsfm_out = self._forward_fused(buf, B, x.shape[-1], dt)
            x = cs.from_rows(buf, geo).float()
        else:
            skip_connections = []
            for downsampling_block in self.encoder:
                x = downsampling_block(x)
                skip_connections.append(x)
            skip_connections = skip_connections[::-1]
            x, tsfm_out = self._bottleneck(x)
            for i, upsampling_block in enumerate(self.decoder):
                skip_i = skip_connections[i]
                x = x + skip_i[:, :, :x.shape[-1]]
                x = upsampling_block(x)

        if self.normalize_input:
            x = x[:, :, :L] * std
        if return_skip_connections:
            skip_connections.append(tsfm_out)
            return x, skip_connections
        return x

    def _fused_dtype(self):
        """Element type of every activation and GEMM operand of the fused path: the autocast dtype -- float16 (torch's
        default, what the reference trains with: src/training/train.py:278-280) or bfloat16 -- else float32;
        accumulation is f32 in every mode."""
        if torch.is_autocast_enabled("cuda"):
            dt = torch.get_autocast_dtype("cuda")
            if dt not in hip.HALF_TYPES:
                raise RuntimeError(f"autocast dtype {dt} is not supported by the fused conv stack")
            return dt
        return torch.float32

    def _forward_fused(self, buf, B, T0, dt):
        """Encoder, bottleneck and decoder on channels-last row buffers (network/convstack.py).
        buf: row buffer of the (B, 1, T0 = valid_length) input in element type dt.  Returns (row buffer of the output,
        its Geo, skips deepest first as (B, C, T) views, tsfm_out)."""
        E = self.encoder_n_layers
        save = torch.is_grad_enabled()
        self._activate_pack_plan(dt)
        geo = cs.Geo(B, T0, 1)
        enc_geos, enc_params = [], []
        for enc in self.encoder:
            T1 = (geo.T - self.kernel_size) // self.stride + 1
            g_mid = cs.Geo(B, T1, enc[0].weight.shape[0])
            if geo.P != 2 * g_mid.P:
                raise RuntimeError("fused conv stack needs an input of valid_length()")
            g_out = cs.Geo(B, T1, enc[2].weight.shape[0] // 2)
            enc_geos.append((geo, g_mid, g_out))
            enc_params += [enc[0].weight, enc[0].bias, enc[2].weight, enc[2].bias]
            geo = g_out
        if getattr(self, "use_stack_backward", True):
            outs = cs.EncoderStack.apply(buf, enc_geos, save, *enc_params)
        else:                              # per-layer autograd nodes (unfused elementwise backward), kept for A/B runs
            outs = []
            for (gi, gm, go), enc in zip(enc_geos, self.encoder):
                y1 = cs.ConvK4S2ReLU.apply(buf, enc[0].weight, enc[0].bias, gi, gm)
                buf = cs.PointwiseGLU.apply(y1, enc[2].weight, enc[2].bias, gm, go, save)
                outs.append(buf)
        cut = self.__dict__.get("_encoder_cut")
        if cut is not None and torch.is_grad_enabled():
            # training/train_step.py, captured multi-rank step: the backward is cut at the encoder's outputs so that the
            # decoder + bottleneck gradients can be exchanged while the encoder's backward runs (a second graph)
            leaves = tuple(o.detach().requires_grad_(True) for o in outs)
            cut.append((tuple(outs), leaves))
            outs = leaves
        skips = [(b, g[2]) for b, g in zip(outs, enc_geos)][::-1]
        buf = outs[-1]

        g_t = cs.Geo(B, geo.T, self.tsfm_conv1.weight.shape[0])
        hbuf = cs.Pointwise.apply(buf, self.tsfm_conv1.weight, self.tsfm_conv1.bias, None, geo, g_t)
        hidden_states = g_t.rows(hbuf)[:, :g_t.T, :g_t.C]
        residual = None
        for layer in self.tsfm_Mamba_layers:
            hidden_states, residual = layer(hidden_states, residual, inference_params=None)
        if _ln.supported(hidden_states, self.norm_f):
            hidden_states, _ = _ln.add_layer_norm(hidden_states, residual, self.norm_f)
        else:
            residual = hidden_states + residual
            hidden_states = self.norm_f(residual.to(dtype=self.norm_f.weight.dtype))
        tsfm_out = hidden_states.permute(0, 2, 1)
        tbuf = cs.to_rows(tsfm_out, g_t, dt)
        # tsfm_conv2 with the deepest skip added in its epilogue
        buf = cs.Pointwise.apply(tbuf, self.tsfm_conv2.weight, self.tsfm_conv2.bias, skips[0][0], g_t, geo)

        dec_geos, dec_params, dec_skips = [], [], []
        for j, dec in enumerate(self.decoder):
            g_glu = cs.Geo(B, geo.T, dec[0].weight.shape[0] // 2)
            g_out = cs.Geo(B, 2 * geo.T + 2, dec[2].weight.shape[1])
            if j < E - 1:
                skip, g_skip = skips[j + 1]
                assert g_skip.T == g_out.T and g_skip.C == g_out.C
                dec_skips.append(skip)
            dec_geos.append((geo, g_glu, g_out))
            dec_params += [dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias]
            geo = g_out
        if getattr(self, "use_stack_backward", True):
            buf = cs.DecoderStack.apply(buf, dec_geos, save, len(dec_skips), *dec_skips, *dec_params)
        else:
            for j, ((gi, gg, go), dec) in enumerate(zip(dec_geos, self.decoder)):
                gbuf = cs.PointwiseGLU.apply(buf, dec[0].weight, dec[0].bias, gi, gg, save)
                buf = cs.ConvT4S2.apply(gbuf, dec[2].weight, dec[2].bias, dec_skips[j] if j < E - 1 else None, gg, go,
                                        j < E - 1)
        return buf, geo, [cs.from_rows(b, g) for b, g in skips], tsfm_out

    def _activate_pack_plan(self, dt):
        """One batched re-pack of all conv weights for this forward (and its backward); see cs.PackPlan.  Skipped
        while no parameter has been modified since the last pack (inference loops, streaming hops)."""
        plans = self.__dict__.setdefault("_pack_plans", {})
        plan = plans.get(dt)
        conv_params = [p for m in (self.encoder, self.decoder, self.tsfm_conv1, self.tsfm_conv2) for p in m.parameters()]
        for layer in self.tsfm_Mamba_layers:              # the Mamba projections' GEMM operands ride in the same gather
            mixer = getattr(layer, "mixer", None)
            for name in ("in_proj", "x_proj", "dt_proj", "out_proj"):
                lin = getattr(mixer, name, None)
                if lin is not None:
                    conv_params.append(lin.weight)
        if plan is None or [p.data_ptr() for p in plan.params] != [p.data_ptr() for p in conv_params]:
            plan = plans[dt] = cs.PackPlan(conv_params)
        # tensor version counters see optimizer steps, load_state_dict and every other in-place update, but not
        # writes through ``param.data``: call invalidate_packed_weights() after those.  Training always re-packs.
        version = (sum(p._version for p in conv_params), len(plan.reqs))
        if torch.is_grad_enabled() or getattr(plan, "packed_version", None) != version or not plan.current:
            plan.refresh()
            plan.packed_version = (version[0], len(plan.reqs))
        cs.set_active_plan(plan)

    def invalidate_packed_weights(self):
        """Drop the cached GEMM-layout copies of the conv weights (rebuilt on the next forward)."""
        self.__dict__.pop("_pack_plans", None)
        self.__dict__.pop("_hop_graph", None)
        self.__dict__.pop("_hop_plan", None)
        self.__dict__.pop("_hop_kernel_why", None)      # (shapes may have changed: ask again)
        self.__dict__.pop("_plist", None)

    # ----------------------------------------------------------------- streaming
    def reset_time_per_frame(self):
        self.total_time = 0
        self.frames = 0

    @property
    def time_per_frame(self):
        return 0 if self.frames == 0 else self.total_time / self.frames

    def allocate_inference_cache_layer(self, layer, batch_size, dtype=None):
        return layer.allocate_inference_cache(batch_size, 1, dtype=dtype)

    def allocate_inference_cache(self, batch_size, max_seqlen, dtype=None, **kwargs):
        return {i: self.allocate_inference_cache_layer(layer.mixer, batch_size, dtype=dtype)
                for i, layer in enumerate(self.tsfm_Mamba_layers)}

    def reset_stream(self):
        """Forget all streaming state (pending samples, conv tails, Mamba states)."""
        dev = self.tsfm_conv1.weight.device
        self.pending = torch.zeros(self.channels_input, 0, dtype=self.dtype, device=dev)
        self.inference_params = None
        self.encoder_decoder_state = {}
        self.input_std = 0
        self._std_frames = 0
        self.__dict__.pop("_hop_graph", None)    # it captured the addresses of the dropped state buffers
        self.__dict__.pop("_hop_state", None)    # state blocks of the one-launch hop (csrc/hop.hip)

    @torch.no_grad()
    def flush(self):
        """Emit the samples still pending: pad one frame of zeros, run it through the SAME stream
        state (so the decoder overlap of the tail is kept), then reset the stream."""
        return self.flush_batch()

    @torch.no_grad()
    def feed(self, noisy_input):
        """noisy_input: (1, n) samples of one stream -> (1, m) denoised samples, m a multiple of total_stride
        (interface of src/network/CleanUMamba.py:370-418)."""
        if noisy_input.dim() != 2:
            raise ValueError("input should be two dimensional.")
        C, _ = noisy_input.shape
        if C != 1:
            raise ValueError(f"Expected 1 channel, got {C}")
        return self.feed_batch(noisy_input)

    @torch.no_grad()
    def flush_batch(self):
        """End of the streams: emit the samples still pending so that feed + flush reproduce ``forward`` on the whole
        signal, tail included.  ``forward`` zero-pads the signal to ``valid_length`` and its last
        ``frame_length - total_stride`` output samples come from the transposed convs' overhang of the LAST real
        frame -- no later frame exists.  So flush (1) pads the stream with exactly the zeros ``forward`` would add
        and runs the hops of the frames that exist, then (2) drains the decoder (``_drain``): the per-layer overlap
        tails and the not-yet-consumed encoder rows are pushed through the remaining decoder layers with no new
        frame.  (The reference's flush() feeds one whole frame of zeros after clearing the decoder state,
        src/network/CleanUMamba.py:358-368 -- SURVEY fact 9: ~70 % error on the tail.)"""
        S, pending_length = self.pending.shape[0], self.pending.shape[1]
        dev = self.pending.device
        consumed = getattr(self, "_std_frames", 0) * self.total_stride
        if consumed + pending_length == 0:
            return torch.zeros(S, 0, device=dev)
        target = self.valid_length(consumed + pending_length)
        pad = torch.zeros(S, target - consumed - pending_length, device=dev, dtype=self.pending.dtype)
        head = self.feed_batch(pad)      # the frames forward() has: they count as timed frames like any other
        if S == 0:
            self.reset_stream()
            return head.new_zeros(0, pending_length)
        out = torch.cat([head, self._drain().to(head.dtype)], 1)[:, :pending_length]
        self.reset_stream()              # the next clip starts a fresh stream: its running std starts over too
        return out

    def _drain(self):
        """Output samples behind the last hop, (S, frame_length - total_stride): what ``forward`` produces there.
        Decoder layer j still holds 2 overhang rows of its transposed conv (``dec{j}``, bias excluded) and encoder
        layer i holds ``2^(E-i) - 2`` output rows no hop has consumed as skips yet; layer j maps its
        ``2^(j+1) - 2`` trailing input rows to ``2^(j+2) - 2`` trailing output rows."""
        hs = self.__dict__.get("_hop_state")
        if hs is not None:                                      # the one-launch hop owns the state: bring it back
            self.encoder_decoder_state = hs["plan"].export_state(self, hs["state"])
        state, E = self.encoder_decoder_state, self.encoder_n_layers
        S, dev = self.pending.shape[0], self.pending.device

        rows_layout = state["enc0"].dim() == 2                 # fused hop: 2-D row buffers; cached hop: (S, C, T)

        def trailing_skip(i):
            t = state[f"enc{i}"]
            hop = self.total_stride // self.stride ** (i + 1)
            if not rows_layout:                                # cached path: (S, C, rows not yet consumed)
                return t.float()
            C = self.encoder[i][2].weight.shape[0] // 2        # fused path: row buffer of the frame's whole window
            T = self.frame_length
            for _ in range(i + 1):
                T = (T - self.kernel_size) // self.stride + 1
            return cs.from_rows(t, cs.Geo(S, T, C))[..., hop:].float()

        def tail(j):
            t = state[f"dec{j}"]
            if rows_layout:                                    # fused path: (S, 2, Cp) channels-last
                return t.transpose(1, 2)[:, :self.decoder[j][2].weight.shape[1]].float()
            return t.float()                                   # cached path: (S, C, 2)

        from ..mamba_ssm.modules.mamba_simple import _proj       # cum_gemm_nt on padded operands: any channel count
        if dev.type == "cuda":
            # the drain's GEMM operands come out of the model's pack plan: re-pack if a parameter changed since the last
            # per-layer hop (the one-launch hop packs its own blob and never touches the plan)
            self._activate_pack_plan(torch.float32)

        def linear_ct(w2d, x):
            """(O, C) x (S, C, T) -> (S, O, T) on the library's GEMM (a handful of columns per stream: rows = S * T)."""
            return _proj(x.transpose(1, 2).contiguous(), w2d.contiguous()).transpose(1, 2)

        def conv_t(g, conv):
            """ConvTranspose1d on a handful of columns as one GEMM per tap + K strided adds (the few-column shapes of the
            drain are not worth a MIOpen solver search, and small transposed convs abort in MIOpen on some boxes)."""
            w, K, S = conv.weight.float(), self.kernel_size, self.stride           # (Cin, Cout, K)
            T = g.shape[-1]
            out = conv.bias.float().view(1, -1, 1).repeat(g.shape[0], 1, (T - 1) * S + K)
            for k in range(K):
                out[..., k:k + (T - 1) * S + 1:S] += linear_ct(w[:, :, k].t(), g)
            return out

        x = None
        for j, dec in enumerate(self.decoder):
            if j == 0:
                y = tail(0) + dec[2].bias.float().view(1, -1, 1)
            else:
                x = x + trailing_skip(E - 1 - j)[..., :x.shape[-1]]
                pre = linear_ct(dec[0].weight.float().squeeze(-1), x) + dec[0].bias.float().view(1, -1, 1)
                y = conv_t(dec[1](pre), dec[2])
                y[..., :self.stride] += tail(j)
            if j != E - 1:
                y = torch.relu(y)
            x = y
        out = x[:, 0]
        if self.normalize_input:
            out = out * self.input_std
        return out

    @torch.no_grad()
    def feed_batch(self, noisy_input):
        """S concurrent streams in lock-step: (S, n) new samples per stream -> (S, m).  Every stream owns a row of
        the pending buffer, of the per-layer conv tails and of the Mamba conv / SSM states; one hop costs the same
        number of kernel launches whatever S is (the reference streams one clip at a time, batch fixed to 1 at
        src/network/CleanUMamba.py:375-381)."""
        if noisy_input.dim() != 2:
            raise ValueError("input should be two dimensional: (streams, samples)")
        S = noisy_input.shape[0]
        if self.pending.shape[0] != S or self.pending.device != noisy_input.device:
            if self.pending.shape[1] != 0 or self.inference_params is not None:
                if self.pending.shape[0] != S:
                    raise ValueError(f"stream count changed from {self.pending.shape[0]} to {S}; call reset_stream()")
            self.pending = torch.zeros(S, 0, dtype=noisy_input.dtype, device=noisy_input.device)
        if self.inference_params is None:
            self.inference_params = InferenceParams(max_seqlen=1, max_batch_size=S,
                                                    key_value_memory_dict=self.allocate_inference_cache(S, 1),
                                                    seqlen_offset=1)
        begin = time.time()
        total_stride = self.total_stride
        self.pending = torch.cat([self.pending, noisy_input], dim=1)
        denoised_frames = []
        self.__dict__["_wv_call"] = None            # weights cannot change inside one call: checked on its first hop only
        while self.pending.shape[1] >= self.frame_length:
            if S == 0:                                  # no stream: only the bookkeeping of the hops that would have run
                n_hops = (self.pending.shape[1] - self.frame_length) // total_stride + 1
                self.frames += n_hops
                self._std_frames = getattr(self, "_std_frames", 0) + n_hops
                denoised_frames.append(self.pending.new_zeros(0, n_hops * total_stride))
                self.pending = self.pending[:, n_hops * total_stride:]
                break
            hs = self._hop_kernel_state()
            if hs is not None:
                # every remaining hop of this call in ONE launch (csrc/hop.hip): a workgroup per stream walks them
                n_hops = (self.pending.shape[1] - self.frame_length) // total_stride + 1
                out = torch.empty(S, n_hops * total_stride, dtype=torch.float32, device=self.pending.device)
                hs["plan"].run(hs["state"], self.pending, out, n_hops)
                self.frames += n_hops
                self._std_frames += n_hops
                if self.normalize_input:
                    self.input_std = hs["state"][:, 0:1]         # the kernel keeps the running std in the state block
                denoised_frames.append(out)
                self.pending = self.pending[:, n_hops * total_stride:]
                break
            self.frames += 1
            self._std_frames = getattr(self, "_std_frames", 0) + 1
            frame = self.pending[:, :self.frame_length]
            if self.normalize_input:
                # running mean of the per-frame std, per stream (src/network/CleanUMamba.py:399-401).  The mean runs
                # over the frames of THIS stream (the reference shares one counter with time_per_frame and never
                # resets either; a stream here ends at flush(), so the counter of its running mean ends there too)
                n = self._std_frames
                self.input_std = (frame.std(dim=1, keepdim=True) + 1e-3) / n + (1 - 1 / n) * self.input_std
                frame = frame / self.input_std
            out = self._hop(frame)[:, :total_stride]
            if self.normalize_input:
                out = out * self.input_std
            denoised_frames.append(out)
            self.pending = self.pending[:, total_stride:]
        self.total_time += time.time() - begin
        if denoised_frames:
            return torch.cat(denoised_frames, 1)
        return torch.zeros(S, 0, device=noisy_input.device)

    def _hop_kernel_state(self):
        """{"plan", "state"} once the one-launch hop (csrc/hop.hip) can take this stream's hops, else None: it needs a
        model the plan supports (hopplan.unsupported_reason), f32 streams on the GPU, and the state the per-layer path
        leaves after the FIRST frame of the streams (whole windows, no history), which is converted here once."""
        hs = self.__dict__.get("_hop_state")
        if hs is not None:
            wv = self.__dict__.get("_wv_call")
            if wv is None:
                wv = self.__dict__["_wv_call"] = self._weights_version()
            if hs["plan_weights"] != wv:                    # weights changed under a live stream: re-pack them
                hs["plan"], hs["plan_weights"] = hopplan.HopPlan(self), wv
            return hs
        if not getattr(self, "use_hop_kernel", True) or getattr(self, "stream_bf16", False) \
                or not getattr(self, "stream_incremental", True):
            return None
        state = self.encoder_decoder_state
        if not state or "enc0" not in state or state["enc0"].dim() != 2 or not self.pending.is_cuda \
                or self.pending.dtype != torch.float32 or self.pending.stride(1) != 1:
            return None
        why = self.__dict__.get("_hop_kernel_why")
        if why is None:
            why = self.__dict__["_hop_kernel_why"] = hopplan.unsupported_reason(self) or ""
        if why:
            return None
        wv = self._weights_version()
        self.__dict__["_wv_call"] = wv
        cached = self.__dict__.get("_hop_plan")
        if cached is None or cached[0] != wv:
            try:
                cached = (wv, hopplan.HopPlan(self))
            except ValueError as exc:                       # e.g. LDS budget: stay on the per-layer path
                self.__dict__["_hop_kernel_why"] = str(exc)
                return None
            self.__dict__["_hop_plan"] = cached
        plan = cached[1]
        hs = {"plan": plan, "plan_weights": wv, "state": plan.import_state(self, self.pending.shape[0])}
        self.__dict__["_hop_state"] = hs
        self.__dict__.pop("_hop_graph", None)
        return hs

    @property
    def hop_kernel_status(self):
        """"active" while the one-launch hop owns the stream state, else why not ("off", "first frame pending", reason)."""
        if self.__dict__.get("_hop_state") is not None:
            return "active"
        if not getattr(self, "use_hop_kernel", True):
            return "off"
        return self.__dict__.get("_hop_kernel_why") or "first frame pending"

    @property
    def hop_graph_status(self):
        """"off" (disabled), "pending" (no hop captured yet), "captured", or "failed: <error>" (hops run eagerly)."""
        if not getattr(self, "use_hop_graph", False):
            return "off"
        hg = self.__dict__.get("_hop_graph")
        if hg is None:
            return "pending"
        return "failed: " + hg["error"] if hg.get("failed") else "captured"

    def _weights_version(self):
        # in-place updates of any parameter (optimizer steps, load_state_dict) bump these counters.  The walk over the
        # module tree costs ~60 us, a whole streaming hop of one stream ~450: the parameter list is cached (and keyed
        # by the identity of the Parameter objects, which pruning / load_pruned_state_dict replace).
        plist, age = self.__dict__.get("_plist", (None, 0))
        if plist is None or age >= 64:              # re-walk now and then: foreign code may swap Parameter objects
            plist, age = list(self.parameters()), 0
        self.__dict__["_plist"] = (plist, age + 1)
        return sum(p._version for p in plist)

    def _hop(self, frame):
        """Eager first hop (it creates the state buffers), hipGraph replay afterwards."""
        denoise = self._denoise_frame
        if frame.is_cuda and getattr(self, "use_fused_stream", True) and getattr(self, "use_fused_convs", True) \
                and cs.supported(self) \
                and frame.shape[1] == self.valid_length(1) and frame.dtype == torch.float32 \
                and frame.shape[0] >= getattr(self, "fused_min_streams", 1):
            # (single streams too: 0.45 ms per hop against 0.52 ms on the cached path)
            denoise = self._denoise_frame_fused
        if not (getattr(self, "use_hop_graph", False) and frame.is_cuda and self.encoder_decoder_state):
            return denoise(frame)
        hg = self.__dict__.get("_hop_graph")
        wv = self.__dict__.get("_wv_call")
        if wv is None:
            wv = self.__dict__["_wv_call"] = self._weights_version()
        if hg is not None and not hg.get("failed") and hg.get("weights") != wv:
            hg = None          # the graph replays kernels on the weight copies of its capture: capture again
        if hg is None:
            hg = {"failed": False}
            try:
                static_in = frame.clone()
                stream = torch.cuda.Stream(device=frame.device)
                stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(stream):      # warm-up on a side stream, as graph capture requires
                    saved = {k: v.clone() for k, v in self.encoder_decoder_state.items()}
                    cache = {k: tuple(t.clone() for t in v)
                             for k, v in self.inference_params.key_value_memory_dict.items()}
                    denoise(static_in, True)
                    # undo the warm-up's state changes
                    for k, v in saved.items():
                        self.encoder_decoder_state[k].copy_(v)
                    for k, v in cache.items():
                        for dst, src in zip(self.inference_params.key_value_memory_dict[k], v):
                            dst.copy_(src)
                torch.cuda.current_stream().wait_stream(stream)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    static_out = denoise(static_in, True)
                # capture does not execute: state is untouched
                hg.update(graph=graph, static_in=static_in, static_out=static_out, shape=tuple(frame.shape),
                          weights=wv)
            except Exception as exc:                 # noqa: BLE001 - capture is an optimisation; stay eager
                hg = {"failed": True, "error": repr(exc)}
                warnings.warn(f"CleanUMamba: hipGraph capture of the streaming hop failed ({exc!r}); this stream "
                              "runs its hops eagerly (2-4x slower).  See model.hop_graph_status.")
            self.__dict__["_hop_graph"] = hg
        if hg.get("failed") or hg["shape"] != tuple(frame.shape):
            return denoise(frame)
        hg["static_in"].copy_(frame)
        hg["graph"].replay()
        return hg["static_out"].clone()

    def _denoise_frame_fused(self, frame, inplace=True):
        assert inplace, "the fused hop always updates its stream state in place"
        with cs.small_m_gemms():
            return self._denoise_frame_fused_impl(frame)

    def _denoise_frame_fused_impl(self, frame):
        """One hop on the fused GEMM kernels, same arithmetic as _denoise_frame.  Every encoder layer keeps a persistent
        window of its output (the decoder's skips read its oldest rows); the first hop of a stream computes the
        windows whole (S independent clips of valid_length(1) samples), later hops compute only the hop's new rows
        from the newest rows of the window below (cum_stream_tail_rows) and append them (cum_stream_window_update),
        so older activations keep the input scaling of the hop that produced them, as with the reference's per-layer
        caches (``stream_incremental = False`` recomputes the windows every hop); a decoder layer is
        1x1+GLU GEMM, transposed-conv GEMM and one overlap-add kernel (cum_stream_overlap_add) that also applies
        ReLU, adds the skip and keeps the tail for the next hop."""
        S, E, dev = frame.shape[0], self.encoder_n_layers, frame.device
        dt = torch.bfloat16 if getattr(self, "stream_bf16", False) else torch.float32
        self._activate_pack_plan(dt)
        geo = cs.Geo(S, frame.shape[1], self.encoder[0][0].weight.shape[1])
        state, lib = self.encoder_decoder_state, hip.lib()
        # After the first hop of a stream only the hop's new rows of every layer are computed: layer i emits
        # n = total_stride >> (i + 1) rows from the 2 n + 2 newest rows of its input (2 carried + 2 n new ones).
        incremental = state.get("enc0") is not None and getattr(self, "stream_incremental", True)
        buf = None if incremental else cs.to_rows(frame.unsqueeze(1), geo, dt)
        dc = hip.dtype_code(dt)
        enc_geos, outs, n_new = [], [], self.total_stride
        for i, enc in enumerate(self.encoder):
            T1 = (geo.T - self.kernel_size) // self.stride + 1
            g_mid = cs.Geo(S, T1, enc[0].weight.shape[0])
            g_out = cs.Geo(S, T1, enc[2].weight.shape[0] // 2)
            n_new //= self.stride
            window = state.get(f"enc{i}")
            if incremental:
                t_in = self.stride * n_new + self.kernel_size - self.stride
                g_cin = cs.Geo(S, t_in, geo.C)
                xin = state.get(f"encin{i}")
                if xin is None:                   # persistent: its framing rows are zeroed once
                    xin = state[f"encin{i}"] = g_cin.new(dt, dev, zero=True)
                if i == 0:
                    g_cin.rows(xin)[:, :t_in, :1] = frame[:, frame.shape[1] - t_in:].unsqueeze(-1).to(dt)
                # (deeper layers: the window update of the layer below has already written xin)
                g_cm, g_co = cs.Geo(S, n_new, g_mid.C), cs.Geo(S, n_new, g_out.C)
                y1 = cs._conv_relu_fwd(xin, enc[0].weight, enc[0].bias, g_cin, g_cm)
                fresh, _ = cs._glu_fwd(y1, enc[2].weight, enc[2].bias, g_cm, g_co, False)
                nxt = state.get(f"encin{i + 1}")   # the next layer's compact input: 2 carried rows + the new ones
                with torch.cuda.device(dev):        # in place, one launch (windows are far below 8192 kept rows)
                    hip.check(lib.cum_stream_window_update(dc, S, g_out.T, n_new, g_out.Cp, hip.ptr(window[1:]),
                                                           hip.ptr(fresh[1:]), g_out.P, g_co.P, g_out.T - n_new,
                                                           None, None if nxt is None else hip.ptr(nxt[1:]),
                                                           n_new + 4, hip.stream_ptr()))
            else:
                y1 = cs._conv_relu_fwd(buf, enc[0].weight, enc[0].bias, geo, g_mid)
                fresh, _ = cs._glu_fwd(y1, enc[2].weight, enc[2].bias, g_mid, g_out, False)
                # the layer's window keeps older rows as the hop that produced them left them (per-layer caches of
                # the reference, :425-447): only the n_new newest rows of the recomputed window are taken over
                if window is None:
                    window = state[f"enc{i}"] = fresh
                    # compact input of the incremental hops; allocated here so that the captured hop allocates nothing
                    t_in = self.stride * n_new + self.kernel_size - self.stride
                    state[f"encin{i}"] = cs.Geo(S, t_in, geo.C).new(dt, dev, zero=True)
                else:
                    with torch.cuda.device(dev):
                        hip.check(lib.cum_stream_window_update(dc, S, g_out.T, n_new, g_out.Cp, hip.ptr(window[1:]),
                                                               hip.ptr(fresh[1:]), g_out.P, g_out.P, 0, None, None, 0,
                                                               hip.stream_ptr()))
            enc_geos.append((geo, g_mid, g_out))
            outs.append(window)
            buf, geo = window, g_out
        x, _ = self._bottleneck(cs.from_rows(outs[-1], geo).float(), inference_params=self.inference_params,
                                pointwise_as_linear=True)                                                     # (S, C, 1)
        L = x.shape[-1]
        x = x + cs.from_rows(outs[-1], geo)[..., :L].float()
        g_in = cs.Geo(S, L, x.shape[1])
        ubuf = cs.to_rows(x, g_in, dt)
        for j, dec in enumerate(self.decoder):
            last = j == E - 1
            g_glu = cs.Geo(S, L, dec[0].weight.shape[0] // 2)
            g_ct = cs.Geo(S, 2 * L + 2, dec[2].weight.shape[1])
            gbuf, _ = cs._glu_fwd(ubuf, dec[0].weight, dec[0].bias, g_in, g_glu, False)
            ybuf, _ = cs._convt_fwd(gbuf, dec[2].weight, dec[2].bias, None, g_glu, g_ct, False)
            tail = state.get(f"dec{j}")
            if tail is None:                       # first hop of the stream: nothing to overlap with
                tail = state[f"dec{j}"] = torch.zeros(S, 2, g_ct.Cp, dtype=dt, device=dev)
            g_next = cs.Geo(S, 2 * L, g_ct.C)
            # persistent between hops (the kernel rewrites every data row, the framing rows stay zero) -- except the
            # last layer's, which is handed to the caller
            nbuf = None if last else state.get(f"decbuf{j}")
            if nbuf is None or nbuf.dtype != dt:
                nbuf = g_next.new(dt, dev, zero=True)
                if not last:
                    state[f"decbuf{j}"] = nbuf
            skip, skip_pitch = None, 0
            if not last:
                sbuf, g_skip = outs[E - 2 - j], enc_geos[E - 2 - j][2]
                assert g_skip.C == g_ct.C and g_skip.T >= 2 * L
                skip, skip_pitch = sbuf[1:], g_skip.P
            with torch.cuda.device(dev):
                hip.check(lib.cum_stream_overlap_add(
                    hip.dtype_code(dt), S, 2 * L, g_ct.Cp, g_ct.C, hip.ptr(ybuf[1:]), g_ct.P, hip.ptr(tail),
                    hip.ptr(dec[2].bias.float()), hip.ptr(skip), skip_pitch, hip.ptr(nbuf[1:]), g_next.P,
                    int(not last), hip.stream_ptr()))
            ubuf, g_in, L = nbuf, g_next, 2 * L
        return cs.from_rows(ubuf, g_in)[:, 0].float()

    def _denoise_frame(self, frame, inplace=False):
        """One hop: frame (S, frame_length) -> (S, >= total_stride) samples.  Encoder outputs that overlap
        the previous frame are cached per layer; the decoder keeps the last ``stride`` samples of every
        transposed conv for overlap-add with the next frame."""
        x = frame.unsqueeze(1)
        state = self.encoder_decoder_state
        skip_connections = []
        hop = self.total_stride
        for i, encode in enumerate(self.encoder):
            hop //= self.stride                      # new outputs of this layer per frame
            prev = state.get(f"enc{i}")
            if prev is not None:
                length = x.shape[2]
                n_new = (length - self.kernel_size) // self.stride + 1 - prev.shape[-1]
                x = x[..., length - self.kernel_size - self.stride * (n_new - 1):]
            x = encode(x)
            if prev is not None:
                x = torch.cat([prev, x], -1)
            if inplace:
                prev.copy_(x[..., hop:])             # static buffer (hipGraph replay reads it next hop)
            else:
                state[f"enc{i}"] = x[..., hop:].clone()
            skip_connections.append(x)

        x, _ = self._bottleneck(x, inference_params=self.inference_params)

        for i, upsampling_block in enumerate(self.decoder):
            skip_i = skip_connections[-1 - i]        # deepest first, as in forward()
            x = x + skip_i[..., :x.shape[-1]]
            x = upsampling_block[2](upsampling_block[1](upsampling_block[0](x)))
            prev = state.get(f"dec{i}")
            tail = x[..., -self.stride:] - upsampling_block[2].bias.view(-1, 1)
            x = x[..., :-self.stride]
            if prev is not None:
                x = torch.cat([x[..., :self.stride] + prev, x[..., self.stride:]], -1)
            if inplace:
                prev.copy_(tail)
            else:
                state[f"dec{i}"] = tail
            if i != self.encoder_n_layers - 1:
                x = upsampling_block[3](x)
        return x[:, 0]

    # ------------------------------------------------------------ pruned loading
    def load_pruned_state_dict(self, pruned_state_dict):
        """Load a structurally pruned checkpoint (interface of src/network/CleanUMamba.py:492-550, used by
        src/examples/loading_pretrained_models.py:12-13): every parameter takes the checkpoint's shape, the
        modules' size attributes follow their new weights, then the dict is loaded strictly.  Keys absent from the
        checkpoint keep their current tensors and are reported by the strict load."""
        resize = {
            nn.LayerNorm: lambda m, w: setattr(m, "normalized_shape", tuple(w.shape)),
            nn.Linear: lambda m, w: (setattr(m, "out_features", w.shape[0]), setattr(m, "in_features", w.shape[1])),
            nn.ConvTranspose1d: lambda m, w: (setattr(m, "in_channels", w.shape[0]),
                                              setattr(m, "out_channels", w.shape[1])),
            # a depthwise conv (the Mamba conv1d) stays depthwise: groups follows the channel count.  in_channels is
            # set to weight.shape[1] (= 1 there) as the reference does; F.conv1d only looks at the tensors.
            nn.Conv1d: lambda m, w: (setattr(m, "out_channels", w.shape[0]), setattr(m, "in_channels", w.shape[1]),
                                     setattr(m, "groups", w.shape[0] if m.groups > 1 else m.groups)),
        }
        for prefix, module in self.named_modules():
            dot = prefix + "." if prefix else ""
            own = list(module._parameters.items()) + [(k, b) for k, b in module._buffers.items()
                                                       if k not in module._non_persistent_buffers_set]
            for name, tensor in own:
                src = pruned_state_dict.get(dot + name)
                if tensor is not None and src is not None:
                    tensor.data = src.detach().clone().to(device=tensor.device)
            if getattr(module, "weight", None) is not None:
                for kind, fix in resize.items():
                    if isinstance(module, kind):
                        fix(module, module.weight)
                        break
        for module in self.modules():
            if type(module).__name__ == "Mamba":       # by name, as the reference does (:540): pickled models qualify
                module.d_model = module.in_proj.in_features
                module.d_inner = module.x_proj.in_features
                module.dt_rank = module.dt_proj.in_features
                module.d_state = (module.x_proj.out_features - module.dt_rank) // 2
                module.expand = module.d_inner / module.d_model
        self.load_state_dict(pruned_state_dict, strict=True)
        self.invalidate_packed_weights()
