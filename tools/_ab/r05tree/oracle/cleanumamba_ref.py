"""Oracle (test infrastructure): functional pure-torch restatement of the whole
CleanUMamba forward, the streaming semantics and the training loss.

It takes a reference-format ``state_dict`` (keys as listed in SURVEY.md 8a/a1)
and walks it with ``torch.nn.functional`` calls, so it needs no module classes
and can travel to the GPU box, where /root/reference does not exist.  It is
pinned in ``tests/test_oracle_golden.py`` against outputs of the reference's own
``src/network/CleanUMamba.py`` run in the build container (tests/golden/e2e_*).

Follows (file:line under /root/reference):
  forward ................ src/network/CleanUMamba.py:252-324
  valid_length ........... src/network/CleanUMamba.py:225-246
  encoder layer .......... src/network/CleanUMamba.py:108-113, GLU src/network/layers.py:26-33
  decoder layer .......... src/network/CleanUMamba.py:121-130, 313-316
  Mamba block ............ oracle/mamba_ref.py (third-party, see its header)
  loss ................... src/util/util.py:215-327, src/util/stft_loss.py:16-184
"""
import math

import torch
import torch.nn.functional as F

from . import mamba_ref


def valid_length(length, depth, kernel_size=4, stride=2):
    for _ in range(depth):
        length = 1 if length < kernel_size else 1 + math.ceil((length - kernel_size) / stride)
    for _ in range(depth):
        length = (length - 1) * stride + kernel_size
    return int(length)


def glu(x):
    a, b = x.chunk(2, dim=1)
    return a * torch.sigmoid(b)


def count_layers(sd, prefix):
    idx = {int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix)}
    return (max(idx) + 1) if idx else 0


def encoder_layer(sd, i, x, stride=2, store=None):
    """``store``: optional map applied to the H-channel intermediate (tests pass the straight-through rounding to the
    16-bit type a kernel path stores it in); None = the reference arithmetic."""
    p = f"encoder.{i}."
    x = F.relu(F.conv1d(x, sd[p + "0.weight"], sd[p + "0.bias"], stride=stride))
    if store is not None:
        x = store(x)
    return glu(F.conv1d(x, sd[p + "2.weight"], sd[p + "2.bias"]))


def decoder_layer(sd, j, x, last, stride=2, store=None):
    p = f"decoder.{j}."
    x = glu(F.conv1d(x, sd[p + "0.weight"], sd[p + "0.bias"]))
    if store is not None:
        x = store(x)
    x = F.conv_transpose1d(x, sd[p + "2.weight"], sd[p + "2.bias"], stride=stride)
    return x if last else F.relu(x)


def mamba_mixer(sd, p, h, eps_unused=None):
    """Mamba.forward, non-fast path. h: (B, L, d_model)."""
    bsz, L, _ = h.shape
    w_in = sd[p + "in_proj.weight"]
    d_inner = w_in.shape[0] // 2
    xz = (w_in @ h.reshape(bsz * L, -1).t()).reshape(-1, bsz, L).permute(1, 0, 2)
    x, z = xz.chunk(2, dim=1)
    x = mamba_ref.causal_conv1d_ref(x, sd[p + "conv1d.weight"].squeeze(1), sd[p + "conv1d.bias"], "silu")
    w_dt = sd[p + "dt_proj.weight"]
    R = w_dt.shape[1]
    x_dbl = F.linear(x.permute(0, 2, 1).reshape(bsz * L, d_inner), sd[p + "x_proj.weight"])
    N = (x_dbl.shape[1] - R) // 2
    dt, Bm, Cm = torch.split(x_dbl, [R, N, N], dim=-1)
    dt = (w_dt @ dt.t()).reshape(-1, bsz, L).permute(1, 0, 2)
    Bm = Bm.reshape(bsz, L, N).permute(0, 2, 1).contiguous()
    Cm = Cm.reshape(bsz, L, N).permute(0, 2, 1).contiguous()
    A = -torch.exp(sd[p + "A_log"].to(h.dtype if h.dtype == torch.float64 else torch.float32))
    y = mamba_ref.selective_scan_ref(x, dt, A, Bm, Cm, sd[p + "D"], z=z,
                                     delta_bias=sd[p + "dt_proj.bias"], delta_softplus=True)
    return F.linear(y.permute(0, 2, 1), sd[p + "out_proj.weight"])


def bottleneck(sd, x, eps=1e-5):
    """tsfm_conv1 -> N x Block -> add + norm_f -> tsfm_conv2.  x: (B, C, T)."""
    x = F.conv1d(x, sd["tsfm_conv1.weight"], sd["tsfm_conv1.bias"])
    h = x.permute(0, 2, 1)
    residual = None
    for k in range(count_layers(sd, "tsfm_Mamba_layers.")):
        p = f"tsfm_Mamba_layers.{k}."
        residual = h if residual is None else h + residual
        d = residual.shape[-1]
        h = F.layer_norm(residual, (d,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps)
        h = mamba_mixer(sd, p + "mixer.", h)
    residual = h + residual if residual is not None else h
    h = F.layer_norm(residual, (residual.shape[-1],), sd["norm_f.weight"], sd["norm_f.bias"], eps)
    tsfm_out = h.permute(0, 2, 1)
    return F.conv1d(tsfm_out, sd["tsfm_conv2.weight"], sd["tsfm_conv2.bias"]), tsfm_out


def forward_ref(sd, noisy, normalize_input=True, eps=1e-5, stride=2, kernel_size=4,
                return_intermediates=False):
    """Whole-network forward.  Does NOT mutate ``noisy`` (the reference divides
    its argument in place, CleanUMamba.py:262; the returned value is identical)."""
    if noisy.dim() == 2:
        noisy = noisy.unsqueeze(1)
    bsz, c, L = noisy.shape
    assert c == 1
    E = count_layers(sd, "encoder.")
    if normalize_input:
        std = noisy.std(dim=2, keepdim=True) + 1e-3
        x = noisy / std
    else:
        x = noisy
    x = F.pad(x, (0, valid_length(L, E, kernel_size, stride) - L))
    skips = []
    for i in range(E):
        x = encoder_layer(sd, i, x, stride)
        skips.append(x)
    skips = skips[::-1]
    x, tsfm_out = bottleneck(sd, x, eps)
    inter = {"tsfm_in": skips[0], "tsfm_out": tsfm_out}
    for j in range(E):
        x = x + skips[j][:, :, :x.shape[-1]]
        x = decoder_layer(sd, j, x, last=(j == E - 1), stride=stride)
    if normalize_input:
        x = x[:, :, :L] * std
    if return_intermediates:
        return x, skips, inter
    return x


# ----------------------------------------------------------------------- loss
def stft_mag(x, fft_size, hop, win_length, window):
    s = torch.stft(x, fft_size, hop, win_length, window, return_complex=True)
    return torch.sqrt(torch.clamp(s.real ** 2 + s.imag ** 2, min=1e-7)).transpose(2, 1)


def mrstft_loss_ref(x, y, fft_sizes=(512, 1024, 2048), hop_sizes=(50, 120, 240),
                    win_lengths=(240, 600, 1200), sc_lambda=0.5, mag_lambda=0.5, band="full"):
    sc, mag = 0.0, 0.0
    for fs, hs, wl in zip(fft_sizes, hop_sizes, win_lengths):
        w = torch.hann_window(wl, dtype=x.dtype, device=x.device)
        xm, ym = stft_mag(x, fs, hs, wl, w), stft_mag(y, fs, hs, wl, w)
        if band == "high":
            k = xm.shape[1] // 2
            xm, ym = xm[:, k:, :], ym[:, k:, :]
        sc = sc + torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
        mag = mag + F.l1_loss(torch.log(ym), torch.log(xm))
    n = len(fft_sizes)
    return sc * sc_lambda / n, mag * mag_lambda / n


def loss_ref(denoised, clean, ell_p=1, ell_p_lambda=1, stft_lambda=1, stft_config=None):
    """loss_fn arithmetic (src/util/util.py:303-325) given the network output."""
    stft_config = stft_config or {}
    ae = F.l1_loss(denoised, clean) if ell_p == 1 else F.mse_loss(denoised, clean)
    loss = ae * ell_p_lambda
    if stft_lambda > 0:
        sc, mag = mrstft_loss_ref(denoised.squeeze(1), clean.squeeze(1), **stft_config)
        loss = loss + (sc + mag) * stft_lambda
    return loss


# ------------------------------------------------------------------ streaming
def streaming_ref(sd, noisy_1d, eps=1e-5):
    """The INTENDED streaming semantics: feeding a stream frame by
    frame yields the parallel ``forward`` output (normalize_input=False).  The
    reference's own feed() crashes as shipped (CleanUMamba.py:474), so the oracle
    for streaming is the parallel forward itself."""
    return forward_ref(sd, noisy_1d[None, None, :], normalize_input=False, eps=eps)[0]
