"""cProfile of the host side of the train step (GPU box): where the per-step enqueue time goes."""
import cProfile, pstats, sys, io, os
sys.path.insert(0, ".")
import torch
from cleanumamba_amd.network import Net
from cleanumamba_amd.training.train_step import TrainStep
import bench
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", bench.E8).to(dev).train()
step = TrainStep(net, autocast_dtype=torch.bfloat16)
g = torch.Generator(device=dev).manual_seed(1234)
clean = 0.05 * torch.randn(16, 1, bench.CLIP, generator=g, device=dev)
noisy = clean + 0.05 * torch.randn(16, 1, bench.CLIP, generator=g, device=dev)
for _ in range(5): step(clean, noisy)
torch.cuda.synchronize()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pr = cProfile.Profile()
pr.enable()
for _ in range(n): step(clean, noisy)
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
