"""Functional check of the whole fused training path (GPU box): E8, bf16 autocast, synthetic 'speech' = a few
amplitude-modulated sines + white noise; the loss must fall and stay finite over a few dozen steps."""
import math
import sys

import torch

sys.path.insert(0, ".")
from bench import E8
from cleanumamba_amd.network import Net
from cleanumamba_amd.training.train_step import TrainStep

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", E8).to(dev).train()
step = TrainStep(net, optimization={"n_iters": 200, "learning_rate": 3e-4}, autocast_dtype=torch.bfloat16)   # 10 warm-up steps
L, B = 64000, 8
t = torch.arange(L, device=dev) / 16000.0
g = torch.Generator(device=dev).manual_seed(3)
losses = []
for i in range(steps):
    f0 = 100 + 300 * torch.rand(B, 1, generator=g, device=dev)
    clean = sum(0.1 / k * torch.sin(2 * math.pi * k * f0 * t + k) for k in range(1, 5)) * (0.6 + 0.4 * torch.sin(2 * math.pi * 3 * t))
    clean = clean.unsqueeze(1)
    noisy = clean + 0.05 * torch.randn(B, 1, L, generator=g, device=dev)
    loss, _ = step(clean, noisy)
    losses.append(float(loss))
    if i % 10 == 0 or i == steps - 1:
        print(i, round(losses[-1], 4), flush=True)
assert all(math.isfinite(v) for v in losses), "non-finite loss"
first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
print("mean of first 5:", round(first, 4), " mean of last 5:", round(last, 4))
assert last < 0.9 * first, "loss did not fall"
print("OK")
