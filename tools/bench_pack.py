"""Where the weight re-pack gather spends its time: the model's real index against an identity index and a plain cast."""
import sys, torch
sys.path.insert(0, ".")
import bench
from cleanumamba_amd.network import Net, convstack as cs
from cleanumamba_amd.training.train_step import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = Net("CleanUMamba", bench.E8).to(dev).train()
step = TrainStep(net, autocast_dtype=torch.float16, use_graph=False)
g = torch.Generator(device=dev).manual_seed(1)
clean = 0.05 * torch.randn(2, 1, 16000, generator=g, device=dev)
for _ in range(2):
    step(clean, clean)
plan = net._pack_plans[torch.float16]
for gk, (gi, metas) in plan.gidx.items():
    n = gi.numel()
    src = plan.source
    out = torch.empty(n, dtype=gk[0], device=dev)
    t_real = bench._time(lambda: cs.gather(src, gi, gk[0], out=out))
    ident = (torch.arange(n, device=dev, dtype=torch.int64) % src.numel()).to(torch.int32)
    t_id = bench._time(lambda: cs.gather(src, ident, gk[0], out=out))
    t_cast = bench._time(lambda: out.copy_(src[:n] if n <= src.numel() else src.repeat(3)[:n]))
    print(gk, "elements", n, "real idx %.3f ms" % t_real, "identity idx %.3f ms" % t_id, "plain cast/copy %.3f ms" % t_cast, flush=True)
