"""The per-step weight re-pack of the benched model: index-free cum_pack2d (+ index gather for what does not separate)
against the all-index gather (CUM_PACK2D=0), and each operand checked bit for bit against the index gather."""
import os, sys, torch
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
plan.refresh()                     # from the weights as the last optimizer step left them
bad = 0
for rk, (gidx, shape) in plan.reqs.items():
    want = cs.gather(plan.source, gidx.to(dev), rk[2]).view(shape)
    got = plan.current[rk]
    if not torch.equal(want, got):
        bad += 1
        print("MISMATCH", rk[0], shape, float((want.float() - got.float()).abs().max()))
print("operands", len(plan.reqs), "mismatching", bad)
for gk, (gi, metas, pack2d, rest_start, total) in plan.gidx.items():
    print(gk, "elements", total, "through pack2d", rest_start, "tiles", pack2d[3] if pack2d else 0,
          "transposed jobs", int(sum(1 for j in pack2d[0].cpu().numpy().view("<i4").reshape(-1, 8)[:, 6] if j)) if pack2d else 0)
print("refresh (PACK2D=%s): %.3f ms" % (os.environ.get("CUM_PACK2D", "1"), bench._time(plan.refresh)))
