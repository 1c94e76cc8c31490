"""Sizes and timing of the weight re-pack (PackPlan.refresh) of the E8 model (GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from bench import CLIP, E8
from cleanumamba_amd.network import Net
from cleanumamba_amd.network import convstack as cs
from cleanumamba_amd.training.train_step import TrainStep
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", E8).to(dev).train()
step = TrainStep(net, autocast_dtype=torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(1)
clean = 0.05 * torch.randn(2, 1, CLIP, generator=g, device=dev)
for _ in range(3):
    step(clean, clean)
calls = []
orig = cs.gather
def spy(src, idx, dt):
    calls.append((str(src.dtype), str(dt), idx.numel()))
    return orig(src, idx, dt)
cs.gather = spy
step(clean, clean)
cs.gather = orig
big = [c for c in calls if c[2] > 1_000_000]
print("gather calls per step:", len(calls), " >1M elements:", len(big), big[:12])
for plan in net._pack_plans.values():
    for dt, (gi, metas) in plan.gidx.items():
        print("plan", dt, gi.numel(), len(metas))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        plan.refresh()
    e.record(); torch.cuda.synchronize()
    print("refresh ms", s.elapsed_time(e) / 10)
