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


def weight_scaling_init(layer):
    """w, b /= sqrt(10 * std(w))  (weight rescaling of arXiv:1911.13254)."""
    w = layer.weight.detach()
    alpha = 10.0 * w.std()
    layer.weight.data /= torch.sqrt(alpha)
    layer.bias.data /= torch.sqrt(alpha)


def anneal_linear(start, end, proportion):
    return start + proportion * (end - start)


def anneal_cosine(start, end, proportion):
    cos_val = cos(pi * proportion) + 1
    return end + (start - end) / 2 * cos_val


class Phase:
    def __init__(self, start, end, n_iter, cur_iter, anneal_fn):
        self.start, self.end = start, end
        self.n_iter = n_iter
        self.anneal_fn = anneal_fn
        self.n = cur_iter

    def step(self):
        self.n += 1
        return self.anneal_fn(self.start, self.end, self.n / self.n_iter)

    def reset(self):
        self.n = 0

    @property
    def is_done(self):
        return self.n >= self.n_iter


class LinearWarmupCosineDecay:
    """Linear warm-up from lr_max/divider to lr_max over ``warmup_proportion`` of the run,
    then cosine decay to lr_max/divider/1e4."""

    def __init__(self, optimizer, lr_max, n_iter, iteration=0, divider=25, warmup_proportion=0.3,
                 phase=("linear", "cosine")):
        self.optimizer = optimizer
        phase1 = int(n_iter * warmup_proportion)
        phase2 = n_iter - phase1
        lr_min = lr_max / divider
        phase_map = {"linear": anneal_linear, "cosine": anneal_cosine}
        self.lr_phase = [
            Phase(lr_min, lr_max, phase1, iteration, phase_map[phase[0]]),
            Phase(lr_max, lr_min / 1e4, phase2, max(0, iteration - phase1), phase_map[phase[1]]),
        ]
        self.phase = 0 if iteration < phase1 else 1

    def step(self):
        lr = self.lr_phase[self.phase].step()
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        if self.lr_phase[self.phase].is_done:
            self.phase += 1
        if self.phase >= len(self.lr_phase):
            for phase in self.lr_phase:
                phase.reset()
            self.phase = 0
        return lr


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
    if ell_p == 2:
        ae_loss = F.mse_loss(denoised_audio, clean_audio)
    elif ell_p == 1:
        ae_loss = F.l1_loss(denoised_audio, clean_audio)
    else:
        raise NotImplementedError
    loss = ae_loss * ell_p_lambda
    output_dic["reconstruct"] = ae_loss.data * ell_p_lambda
    if stft_lambda > 0:
        if mrstftloss is None:
            mrstftloss = MultiResolutionSTFTLoss(sc_lambda=0.5, mag_lambda=0.5, band="high",
                                                 hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200],
                                                 fft_sizes=[512, 1024, 2048]).to(denoised_audio.device)
        sc_loss, mag_loss = mrstftloss(denoised_audio.squeeze(1), clean_audio.squeeze(1))
        loss = loss + (sc_loss + mag_loss) * stft_lambda
        output_dic["stft_sc"] = sc_loss.data * stft_lambda
        output_dic["stft_mag"] = mag_loss.data * stft_lambda
    return loss, output_dic
