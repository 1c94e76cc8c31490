"""Oracle (test infrastructure): deterministic synthetic weights and inputs.

Full-size E6/E8 checkpoints are missing from the reference
(``checkpoints/.MISSING_LARGE_BLOBS``) and 41 M parameters cannot be committed as
a fixture, so tests fill a state dict from a seeded generator, key by key in
sorted order.  The same filler is applied to the reference class when the golden
vectors are made (oracle/make_golden.py) and to the product module in tests.
"""
import math

import torch


def fill_state_dict(shapes, seed=0, dtype=torch.float32):
    """shapes: {key: tuple}.  Returns {key: tensor} with values that keep every
    activation O(1) through the network (fan-in scaled weights, positive dt)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key in sorted(shapes):
        shape = tuple(shapes[key])
        if key.endswith("A_log"):
            n = shape[1]
            t = torch.log(torch.arange(1, n + 1, dtype=torch.float32))[None, :].repeat(shape[0], 1)
            t = t + 0.05 * torch.randn(shape, generator=g)
        elif key.endswith(".D"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif key.endswith("dt_proj.bias"):
            dt = torch.exp(torch.rand(shape, generator=g) * (math.log(0.1) - math.log(1e-3)) + math.log(1e-3))
            t = dt + torch.log(-torch.expm1(-dt))
        elif "norm" in key and key.endswith("weight"):
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        elif key.endswith("bias"):
            t = 0.05 * torch.randn(shape, generator=g)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            if ".2.weight" in key and key.startswith("decoder") and len(shape) == 3:
                fan_in = shape[0] * shape[2] // 2        # ConvTranspose1d (Cin, Cout, K), stride 2
            t = torch.randn(shape, generator=g) * (1.4 / math.sqrt(max(fan_in, 1)))
        sd[key] = t.to(dtype)
    return sd


def waveform(batch, length, seed=1234, scale=0.05):
    """SURVEY.md 8d synthetic input: clean = scale*randn, noisy = clean + scale*randn."""
    g = torch.Generator().manual_seed(seed)
    clean = scale * torch.randn(batch, 1, length, generator=g)
    noise = scale * torch.randn(batch, 1, length, generator=g)
    return clean, clean + noise
