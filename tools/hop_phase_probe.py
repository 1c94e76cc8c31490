"""Cycles per phase of the one-launch streaming hop (csrc/hop.hip) from s_memtime stamps: needs a library built with
-DCUM_HOP_PROBE (tools/hop_probe_build.sh -> tools/_ab/lib_hop_probe.so; run with CUM_LIB pointing at it).  GPU box only."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import CleanUMamba

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
name = sys.argv[2] if len(sys.argv) > 2 else "pruned500k"
dev = torch.device("cuda")
with np.load(f"tests/golden/ckpt_{name}.npz") as f:
    cfg = json.loads(bytes(f["__network_config__"]).decode())
    sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}
net = CleanUMamba(**cfg)
net.load_pruned_state_dict(sd) if name != "442k" else net.load_state_dict(sd)
net = net.to(dev).eval()
hop = net.total_stride
x = 0.05 * torch.randn(S, 40 * hop + net.frame_length, device=dev)
with torch.no_grad():
    for _ in range(3):
        net.reset_stream()
        net.feed_batch(x)
    torch.cuda.synchronize()
assert net.hop_kernel_status == "active"
lib = hip.lib()
lib.cum_stream_hop_probe_read.restype = ctypes.c_int
plan = net.__dict__["_hop_plan"][1]
n = len(plan.ops)
buf = (ctypes.c_ulonglong * (n + 1))()
assert lib.cum_stream_hop_probe_read(buf, n + 1) == 0
t = np.array(buf[:], dtype=np.int64)
KIND = ["end", "std", "enc0 conv (VALU)", "gemm", "ring", "add+layernorm", "conv step", "ssm step", "overlap-add"]
print(f"{name}, {S} streams: shader cycles per op of workgroup 0's last hop (s_memtime)")
tot = t[n] - t[0]
by_kind = {}
for k, op in enumerate(plan.ops):
    label = KIND[op[0]]
    if op[0] == 3:
        label += f" nacc{op[19]} M={op[9]} tiles={op[4]} kc={op[5]}"
    dt = int(t[k + 1] - t[k])
    by_kind[KIND[op[0]]] = by_kind.get(KIND[op[0]], 0) + dt
    print(f"{k:3d} {label:40s} {dt:9d}  {100.0 * dt / tot:5.1f} %")
print(f"hop total {tot}")
fine = (ctypes.c_ulonglong * 161)()
lib.cum_stream_hop_probe_read(fine, 161)
ff = np.array(fine[120:161], dtype=np.int64)
t0 = ff[34] if ff[34] > 0 else (ff[ff > 0].min() if (ff > 0).any() else 0)
print("fine stamps of the probed op (-DCUM_HOP_PROBE_PC, thread 0), cycles after the op's top: 34 top, 35 op decoded and the "
      "next op's stage list requested, [product: 0 entry, 1 first stage's loads issued, 2.. (loads issued, stage computed) "
      "..., 30 slices combined], 36 body done, 37 behind the barrier:")
print("  ", {int(i): int(v - t0) for i, v in enumerate(ff) if v > 0})
for k, v in sorted(by_kind.items(), key=lambda kv: -kv[1]):
    print(f"  {k:24s} {v:9d}  {100.0 * v / tot:5.1f} %")
