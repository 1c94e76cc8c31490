"""One-launch hop vs per-layer hop with the running input std over 30 s streams (GPU box): where do they part?"""
import json, sys
import numpy as np, torch
sys.path.insert(0, ".")
from cleanumamba_amd.network import CleanUMamba
dev = torch.device("cuda")
with np.load("tests/golden/ckpt_pruned500k.npz") as f:
    cfg = json.loads(bytes(f["__network_config__"]).decode())
    sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}


def make():
    net = CleanUMamba(**cfg)
    net.load_pruned_state_dict(sd)
    return net.to(dev).eval()


net = make()
hop, F = net.total_stride, net.frame_length
L = int(sys.argv[1]) if len(sys.argv) > 1 else 480000
x = 0.1 * torch.randn(4, L, device=dev, generator=torch.Generator(device=dev).manual_seed(2024))
x[0] *= 0.01
x[1] *= 5.0


def run(net, kernel, per_call, first_frame_alone):
    net.reset_stream()
    net.use_hop_kernel = kernel
    chunks, i = [], 0
    if first_frame_alone:
        chunks.append(net.feed_batch(x[:, :F]))
        i = F
    while i < L:
        chunks.append(net.feed_batch(x[:, i:i + per_call * hop]))
        i += per_call * hop
    stds = net.input_std.flatten().tolist() if torch.is_tensor(net.input_std) else net.input_std
    chunks.append(net.flush_batch())
    return torch.cat(chunks, 1), stds


def rel(a, b):
    return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


with torch.no_grad():
    for norm in (False, True):
        net.normalize_input = norm
        a, sa = run(net, True, 16, True)
        b, sb = run(net, False, 64, False)
        c, sc = run(net, False, 16, True)
        print("normalize", norm, "stds kernel", sa, "per-layer", sb)
        for s in range(4):
            print(" stream", s, "kernel vs per-layer", rel(a[s], b[s]), "per-layer(16, F first) vs per-layer(64)", rel(c[s], b[s]))
            w = [rel(a[s, i:i + 16000], b[s, i:i + 16000]) for i in range(0, L, 48000)]
            print("   per 1 s window every 3 s:", " ".join(f"{v:.1e}" for v in w))
        # single stream, one at a time, on a fresh net (the reference's own usage: feed one 2-D stream)
        one = make()
        one.normalize_input = norm
        one.use_hop_kernel = False
        outs = []
        for s in range(2):
            one.reset_stream()
            ch = [one.feed(x[s:s + 1, i:i + 64 * hop]) for i in range(0, L, 64 * hop)]
            ch.append(one.flush())
            outs.append(torch.cat(ch, 1))
        for s in range(2):
            print(" stream", s, "single-stream per-layer vs batched per-layer", rel(outs[s][0], b[s]), "vs kernel", rel(outs[s][0], a[s]))
