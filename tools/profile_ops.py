"""Op-level attribution of one E8 train step (GPU box): torch.profiler table grouped by operator and input shape.

usage: python tools/profile_ops.py [steps] > gpurun_out/ops.txt
"""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
from bench import CLIP, E8  # noqa: E402
from cleanumamba_amd.network import Net  # noqa: E402
from cleanumamba_amd.training.train_step import TrainStep  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", E8).to(dev).train()
step = TrainStep(net, autocast_dtype=torch.float16, use_graph=False)
g = torch.Generator(device=dev).manual_seed(1234)
clean = 0.05 * torch.randn(16, 1, CLIP, generator=g, device=dev)
noisy = clean + 0.05 * torch.randn(16, 1, CLIP, generator=g, device=dev)
for _ in range(3):
    step(clean, noisy)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(steps):
        step(clean, noisy)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = sorted((e for e in ka if e.self_cpu_time_total > 0), key=lambda e: -e.self_device_time_total)
tot = sum(e.self_device_time_total for e in rows)
print(f"total self device time per step: {tot / steps / 1e3:.2f} ms")
for e in rows[:160]:
    shapes = str(e.input_shapes)[:110]
    print(f"{e.self_device_time_total / steps / 1e3:8.3f} ms {e.count / steps:6.1f}x  {e.key[:40]:40s} {shapes}")
print("---- by CPU time")
for e in sorted(ka, key=lambda e: -e.self_cpu_time_total)[:25]:
    print(f"{e.self_cpu_time_total / steps / 1e3:8.3f} ms cpu {e.count / steps:6.1f}x  {e.key[:50]:50s} {str(e.input_shapes)[:60]}")
