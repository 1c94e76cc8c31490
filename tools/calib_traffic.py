"""Calibration for the gfx950 FETCH_SIZE / WRITE_SIZE counters on this code's 4-byte-per-lane row accesses:
runs the causal depthwise conv forward (known traffic: reads x once + 3/16 halo, writes y once)."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd.causal_conv1d import causal_conv1d_fn
dev = torch.device("cuda")
x = torch.randn(16, 624, 2048, device=dev).transpose(1, 2)
w, b = torch.randn(2048, 4, device=dev), torch.randn(2048, device=dev)
with torch.no_grad():
    for _ in range(10):
        y = causal_conv1d_fn(x, w, b, "silu")
torch.cuda.synchronize()
print("bytes in/out per launch:", x.numel() * 4, y.numel() * 4)
