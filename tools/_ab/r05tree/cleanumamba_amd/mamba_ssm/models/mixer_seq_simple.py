"""create_block / _init_weights with the signatures the reference calls
(src/network/CleanUMamba.py:12, 174-189, 201-206)."""
import math
from functools import partial

import torch
import torch.nn as nn

from ..modules.mamba_simple import Block, Mamba


def create_block(d_model, ssm_cfg=None, norm_epsilon=1e-5, rms_norm=False, residual_in_fp32=False,
                 fused_add_norm=False, layer_idx=None, device=None, dtype=None):
    if rms_norm:
        raise NotImplementedError("rms_norm=True needs the Triton RMSNorm; the reference runs with LayerNorm "
                                  "(src/network/CleanUMamba.py:45)")
    if ssm_cfg is None:
        ssm_cfg = {}
    factory_kwargs = {"device": device, "dtype": dtype}
    mixer_cls = partial(Mamba, layer_idx=layer_idx, **ssm_cfg, **factory_kwargs)
    norm_cls = partial(nn.LayerNorm, eps=norm_epsilon, **factory_kwargs)
    block = Block(d_model, mixer_cls, norm_cls=norm_cls, fused_add_norm=fused_add_norm,
                  residual_in_fp32=residual_in_fp32)
    block.layer_idx = layer_idx
    return block


def _init_weights(module, n_layer, initializer_range=0.02, rescale_prenorm_residual=True,
                  n_residuals_per_layer=1):
    if isinstance(module, nn.Linear):
        if module.bias is not None and not getattr(module.bias, "_no_reinit", False):
            nn.init.zeros_(module.bias)
    elif isinstance(module, nn.Embedding):
        nn.init.normal_(module.weight, std=initializer_range)
    if rescale_prenorm_residual:
        # residual-branch output projections: kaiming-uniform then / sqrt(n_layer)
        for name, p in module.named_parameters():
            if name in ["out_proj.weight", "fc2.weight"]:
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                with torch.no_grad():
                    p /= math.sqrt(n_residuals_per_layer * n_layer)
