"""Soak of the one-launch hop (GPU box): many streams x many hops, two cuts of the audio into calls and a repeat -- all
three outputs must be bit-identical (the kernel orders a hop's ops with LDS-only barriers; a missing ordering of the
global stream state would show here)."""
import json, sys
import numpy as np, torch
sys.path.insert(0, ".")
from cleanumamba_amd.network import CleanUMamba
name = sys.argv[1] if len(sys.argv) > 1 else "pruned500k"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
HOPS = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
dev = torch.device("cuda")
with np.load(f"tests/golden/ckpt_{name}.npz") as f:
    cfg = json.loads(bytes(f["__network_config__"]).decode())
    sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}
net = CleanUMamba(**cfg)
net.load_state_dict(sd) if name == "442k" else net.load_pruned_state_dict(sd)
net = net.to(dev).eval()
hop, F = net.total_stride, net.frame_length
L = F + HOPS * hop
x = 0.1 * torch.randn(S, L, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
outs = []
with torch.no_grad():
    for per_call in (16, 61, 16):
        net.reset_stream()
        chunks = [net.feed_batch(x[:, :F])]
        for i in range(F, L, per_call * hop):
            chunks.append(net.feed_batch(x[:, i:i + per_call * hop]))
        assert net.hop_kernel_status == "active"
        chunks.append(net.flush_batch())
        outs.append(torch.cat(chunks, 1))
torch.cuda.synchronize()
print(name, S, "streams x", HOPS, "hops:", "identical" if torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]) else "DIFFERENT",
      "finite" if bool(torch.isfinite(outs[0]).all()) else "NON-FINITE", float(outs[0].abs().mean()))
