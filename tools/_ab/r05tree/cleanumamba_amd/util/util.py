"""Pieces of src/util/util.py that sit on the measured train step.

weight_scaling_init ........ src/util/util.py:174-181 (used by the model constructor)
LinearWarmupCosineDecay .... src/util/util.py:69-161
loss_fn .................... src/util/util.py:215-327 (L1 + multi-resolution STFT; the
                             cross-entropy / distillation branches are out of scope)
"""
from math import cos, pi

import torch
import torch.nn.functional as F

from .stft_loss import MultiResolutionSTFTLoss


@torch.no_grad()
def weight_scaling_init(layer):
    """Divide a conv layer's weight and bias by sqrt(10 * std(weight)) -- the rescaling of arXiv:1911.13254 the
    reference applies to every Conv1d / ConvTranspose1d at construction (src/util/util.py:174-181)."""
    scale = layer.weight.std().mul(10.0).sqrt()
    layer.weight.div_(scale)
    layer.bias.div_(scale)


_RAMPS = {
    # value at fraction f in [0, 1] of a segment that starts at a and ends at b
    "linear": lambda a, b, f: a + (b - a) * f,
    "cosine": lambda a, b, f: b + (a - b) * 0.5 * (1.0 + cos(pi * f)),
}


class LinearWarmupCosineDecay:
    """Learning-rate schedule of the reference's training loop (src/util/util.py:115-161, driven from
    src/training/train.py:236-244, 312) as a closed form of the step count.

    With W = int(n_iter * warmup_proportion) warm-up steps and lo = lr_max / divider, the k-th call of ``step()``
    (k = 1, 2, ..., counted from ``iteration`` when resuming) sets
        k <= W :  ramp[0] from lo     to lr_max     at fraction k / W
        k >  W :  ramp[1] from lr_max to lo / 1e4   at fraction (k - W) / (n_iter - W)
    and the count wraps to zero after n_iter steps (the reference restarts both segments)."""

    def __init__(self, optimizer, lr_max, n_iter, iteration=0, divider=25, warmup_proportion=0.3,
                 phase=("linear", "cosine")):
        self.optimizer = optimizer
        self.n_iter = n_iter
        self.warmup = int(n_iter * warmup_proportion)
        self.lr_max = lr_max
        self.lr_lo = lr_max / divider
        self.ramps = (_RAMPS[phase[0]], _RAMPS[phase[1]])
        self.k = iteration

    def lr_at(self, k):
        """Learning rate in force after k steps of a run (1 <= k <= n_iter)."""
        if k <= self.warmup:
            return self.ramps[0](self.lr_lo, self.lr_max, k / self.warmup)
        return self.ramps[1](self.lr_max, self.lr_lo / 1e4, (k - self.warmup) / (self.n_iter - self.warmup))

    def step(self):
        self.k += 1
        lr = self.lr_at(self.k)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        if self.k >= self.n_iter:
            self.k = 0
        return lr


class _Combine(torch.autograd.Function):
    """out = W @ [v_0 .. v_{n-1}] for 0-dim device scalars v_i and a constant matrix W (k x n): the linear combinations
    of loss terms that loss_fn forms (src/util/util.py:300-327: term * lambda, sums over the three STFT resolutions, / 3,
    the total) as three small launches forward and two backward instead of ~40 one-element kernels with their autograd
    nodes (each a launch boundary inside the replayed step)."""

    _W = {}

    @staticmethod
    def forward(ctx, rows, *vals):
        dev = vals[0].device
        key = (rows, dev)
        W = _Combine._W.get(key)
        if W is None:
            W = _Combine._W[key] = torch.tensor(rows, dtype=torch.float32, device=dev)
        v = torch.stack([t.float() for t in vals])
        ctx.W = W
        return (W * v).sum(1)

    @staticmethod
    def backward(ctx, g):
        gv = (ctx.W * g[:, None].float()).sum(0)
        return (None,) + tuple(gv.unbind())


def loss_fn(net, X, cross_entropy=None, ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=None, kd_p=1,
            min_max=(-1, 1), teacher_net=None, student_teacher_adapter_layers=None, **kwargs):
    """loss = ell_p(denoised, clean) * ell_p_lambda + (sc + mag) * stft_lambda.

    X = (clean_audio, noisy_audio), both (B, 1, L).  Returns (loss, dict of components)."""
    assert type(X) == tuple and len(X) == 2
    if cross_entropy or teacher_net is not None:
        raise NotImplementedError("cross-entropy and distillation branches are outside the hot path")
    clean_audio, noisy_audio = X
    output_dic = {}
    denoised_audio = net(noisy_audio)
    if ell_p not in (1, 2):
        raise NotImplementedError
    plain = (denoised_audio.is_cuda and clean_audio.is_cuda and denoised_audio.shape == clean_audio.shape
             and (torch.is_autocast_enabled() or (denoised_audio.dtype == clean_audio.dtype == torch.float32)))
    if plain:
        # two launches with a fixed summation order; ATen's one-value reduction of 16 x 160 000 samples goes through a
        # staging buffer + semaphore that did not survive the replay of the captured train step (csrc/loss.hip)
        from ..network.convstack import LpLoss
        ae_loss = LpLoss.apply(denoised_audio, clean_audio, ell_p)
    elif ell_p == 2:
        ae_loss = F.mse_loss(denoised_audio, clean_audio)
    else:
        ae_loss = F.l1_loss(denoised_audio, clean_audio)
    if stft_lambda > 0 and mrstftloss is None:
        mrstftloss = MultiResolutionSTFTLoss(sc_lambda=0.5, mag_lambda=0.5, band="high",
                                             hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200],
                                             fft_sizes=[512, 1024, 2048]).to(denoised_audio.device)
    if plain and stft_lambda > 0 and hasattr(mrstftloss, "components"):
        # every term on the GPU kernels: one combination instead of a chain of one-element multiplies and adds
        pairs = mrstftloss.components(denoised_audio.squeeze(1), clean_audio.squeeze(1))
        n = len(pairs)
        cs_, cm_ = stft_lambda * mrstftloss.sc_lambda / n, stft_lambda * mrstftloss.mag_lambda / n
        sc_w, mag_w = [cs_, 0.0] * n, [0.0, cm_] * n
        rows = ((float(ell_p_lambda),) + tuple(a + b for a, b in zip(sc_w, mag_w)),     # the loss
                (float(ell_p_lambda),) + (0.0,) * (2 * n),                              # "reconstruct"
                (0.0,) + tuple(sc_w), (0.0,) + tuple(mag_w))                            # "stft_sc", "stft_mag"
        out = _Combine.apply(rows, ae_loss, *[t for pair in pairs for t in pair])
        output_dic["reconstruct"], output_dic["stft_sc"], output_dic["stft_mag"] = out[1].detach(), out[2].detach(), out[3].detach()
        return out[0], output_dic
    loss = ae_loss * ell_p_lambda
    output_dic["reconstruct"] = ae_loss.data * ell_p_lambda
    if stft_lambda > 0:
        sc_loss, mag_loss = mrstftloss(denoised_audio.squeeze(1), clean_audio.squeeze(1))
        loss = loss + (sc_loss + mag_loss) * stft_lambda
        output_dic["stft_sc"] = sc_loss.data * stft_lambda
        output_dic["stft_mag"] = mag_loss.data * stft_lambda
    return loss, output_dic
