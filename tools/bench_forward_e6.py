"""BASELINE config 2: CleanUMamba-E6 (27.2M) forward on 1 x MI355X, batch 32, 10 s @ 16 kHz, and its selective scan
in isolation (B=32, D=2048, N=64, L=2499).  GPU box only."""
import json
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest_shim import golden_meta  # noqa: E402  (tests/golden metadata of the seeded E6 weights)
from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn  # noqa: E402
from cleanumamba_amd.network import CleanUMamba  # noqa: E402
from oracle import synth  # noqa: E402

dev = torch.device("cuda")
meta = golden_meta("e2e_e6_synth")
net = CleanUMamba(**meta["cfg"])
net.load_state_dict(synth.fill_state_dict(dict(zip(meta["keys"], meta["shapes"])), seed=meta["seed"]), strict=True)
net = net.to(dev).eval()
_, noisy = synth.waveform(32, 160000, seed=3)
noisy = noisy.to(dev)


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


out = {}
with torch.no_grad():
    out["forward_f32_ms"] = round(timeit(lambda: net(noisy)), 2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out["forward_bf16_ms"] = round(timeit(lambda: net(noisy)), 2)
    bsz, dim, N, L = 32, 2048, 64, 2499
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        xz = rn(bsz, L, 2 * dim).to(dt)
        dl = (0.3 * rn(bsz, L, dim)).to(dt)
        A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float())[None].repeat(dim, 1)).contiguous()
        xd = rn(bsz, L, 32 + 2 * N)
        D, bias = rn(dim), 0.3 * rn(dim)
        f = lambda: selective_scan_fn(xz[..., :dim].transpose(1, 2), dl.transpose(1, 2), A, xd[..., 32:32 + N].transpose(1, 2),
                                      xd[..., 32 + N:].transpose(1, 2), D, z=xz[..., dim:].transpose(1, 2), delta_bias=bias,
                                      delta_softplus=True)
        ms = timeit(f)
        out[f"scan_{name}_ms"] = round(ms, 3)
        out[f"scan_{name}_Tupdates_per_s"] = round(bsz * L * dim * N / ms / 1e9, 3)
        esz = 4 if dt == torch.float32 else 2
        out[f"scan_{name}_GBs"] = round(bsz * L * (esz * 4 * dim + 4 * 2 * N) / ms / 1e6, 1)
out["samples_per_s_bf16"] = round(32 * 160000 / out["forward_bf16_ms"] * 1e3)
print(json.dumps(out))
