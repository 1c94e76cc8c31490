"""BASELINE config 5: pruned CleanUMamba-E8 (492K) streaming inference, 256 concurrent 30 s @ 16 kHz streams.
Real-time factor = audio seconds produced / wall seconds (aggregate over streams).  GPU box only."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from cleanumamba_amd.network import CleanUMamba

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
SECONDS = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
dev = torch.device("cuda")
CKPT = sys.argv[4] if len(sys.argv) > 4 else "pruned500k"      # any tests/golden/ckpt_<name>.npz
with np.load(f"tests/golden/ckpt_{CKPT}.npz") as f:
    cfg = json.loads(bytes(f["__network_config__"]).decode())
    sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}
net = CleanUMamba(**cfg)
net.load_state_dict(sd) if CKPT == "442k" else net.load_pruned_state_dict(sd)
net = net.to(dev).eval()
if len(sys.argv) > 3:
    net.use_fused_stream = sys.argv[3] != "cached"      # third argument "cached": torch-module hop with encoder caches
    net.stream_bf16 = sys.argv[3] == "bf16"              # "bf16": fused hop with bf16 activations / GEMMs
    net.fused_min_streams = 1                            # an explicit variant also applies to a single stream
    net.use_hop_kernel = sys.argv[3] == "kernel"          # "kernel" (default without the argument): one launch per hop
n = int(SECONDS * 16000)
x = 0.05 * torch.randn(S, n, device=dev)
hop = net.total_stride
with torch.no_grad():
    net.feed_batch(x[:, :4 * hop + net.frame_length])       # warm-up (with a flush: the first drain of a process is slow)
    net.flush_batch()
    net.reset_stream()
    torch.cuda.synchronize()
    t0 = time.time()
    chunk = 16 * hop                                      # 256 ms of audio per call
    for i in range(0, n, chunk):
        net.feed_batch(x[:, i:i + chunk])
    status = net.hop_kernel_status
    net.flush_batch()
    torch.cuda.synchronize()
    dt = time.time() - t0
frames = n // hop
print(json.dumps({"checkpoint": CKPT, "variant": sys.argv[3] if len(sys.argv) > 3 else "kernel", "streams": S, "seconds_per_stream": SECONDS, "wall_s": round(dt, 3),
                  "rtf_aggregate": round(S * SECONDS / dt, 1), "rtf_per_stream": round(SECONDS / dt, 2),
                  "ms_per_hop": round(1e3 * dt / frames, 3), "hop_ms_audio": 1e3 * hop / 16000, "hop_kernel": status}))
