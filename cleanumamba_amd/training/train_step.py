"""One optimisation step with the semantics of the reference's hot loop
(src/training/train.py:255-312): zero grads, loss_fn under optional autocast,
(scaled) backward with the gradient exchange inside it, unscale, clip_grad_norm_(10),
Adam step (lr 1e-4, betas .9/.999, eps 1e-8, fused), LR schedule.

Host-side differences: no per-step ``loss.item()`` -- the loss stays
on the device and is synchronised only when the caller reads it.
"""
import torch
import torch.nn as nn

from ..util.stft_loss import MultiResolutionSTFTLoss
from ..util.util import LinearWarmupCosineDecay, loss_fn

DEFAULT_OPTIM = {"n_iters": 1600000, "learning_rate": 1e-4, "betas": (0.9, 0.999), "eps": 1e-8,
                 "clip_grad_norm_max": 10, "weight_decay": 0, "fused_adam": True}
DEFAULT_LOSS = {"ell_p": 1, "ell_p_lambda": 1, "stft_lambda": 1,
                "stft_config": {"sc_lambda": 0.5, "mag_lambda": 0.5, "band": "full",
                                "hop_sizes": [50, 120, 240], "win_lengths": [240, 600, 1200],
                                "fft_sizes": [512, 1024, 2048]}}


class TrainStep:
    def __init__(self, net, optimization=None, loss_config=None, autocast_dtype=None, iteration=0):
        self.net = net
        self.opt_cfg = dict(DEFAULT_OPTIM, **(optimization or {}))
        self.loss_cfg = dict(DEFAULT_LOSS, **(loss_config or {}))
        dev = next(net.parameters()).device
        fused = bool(self.opt_cfg["fused_adam"]) and dev.type == "cuda"
        self.optimizer = torch.optim.Adam(net.parameters(), lr=self.opt_cfg["learning_rate"],
                                          betas=tuple(self.opt_cfg["betas"]), eps=self.opt_cfg["eps"],
                                          fused=fused, weight_decay=self.opt_cfg["weight_decay"])
        self.scheduler = LinearWarmupCosineDecay(self.optimizer, lr_max=self.opt_cfg["learning_rate"],
                                                 n_iter=self.opt_cfg["n_iters"], iteration=iteration, divider=25,
                                                 warmup_proportion=0.05, phase=("linear", "cosine"))
        self.mrstft = None
        if self.loss_cfg["stft_lambda"] > 0:
            self.mrstft = MultiResolutionSTFTLoss(**self.loss_cfg["stft_config"]).to(dev)
        self.autocast_dtype = autocast_dtype
        # fp16 autocast needs loss scaling (reference: GradScaler, train.py:158-160); bf16 does not
        self.scaler = torch.amp.GradScaler("cuda") if autocast_dtype == torch.float16 else None
        self.buckets = getattr(net, "grad_buckets", None)

    def zero_grad(self):
        if self.buckets is not None:
            self.buckets.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def __call__(self, clean_audio, noisy_audio):
        """Returns (loss tensor on device, grad_norm tensor)."""
        self.zero_grad()
        kw = {k: v for k, v in self.loss_cfg.items() if k != "stft_config"}
        if self.autocast_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                loss, _ = loss_fn(self.net, (clean_audio, noisy_audio), mrstftloss=self.mrstft, **kw)
        else:
            loss, _ = loss_fn(self.net, (clean_audio, noisy_audio), mrstftloss=self.mrstft, **kw)
        if self.scaler is not None:
            self.scaler.scale(loss).backward()
            self.scaler.unscale_(self.optimizer)
            grad_norm = nn.utils.clip_grad_norm_(self.net.parameters(), self.opt_cfg["clip_grad_norm_max"])
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            loss.backward()
            grad_norm = nn.utils.clip_grad_norm_(self.net.parameters(), self.opt_cfg["clip_grad_norm_max"])
            self.optimizer.step()
        self.scheduler.step()
        return loss.detach(), grad_norm
