import sys, torch
sys.path.insert(0, ".")
from cleanumamba_amd.network import Net
from cleanumamba_amd.training.train_step import TrainStep
from cleanumamba_amd.training import flat_optim as fo
import bench
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", bench.E8).to(dev).train()
step = TrainStep(net, autocast_dtype=torch.float16, use_graph=False)
orig = fo.FlatParams.zero_grad
def zg(self):
    print("zero_grad: sunk_last", None if self.sunk_last is None else len(self.sunk_last), "of", len(self.params), "on_write", self.on_write, flush=True)
    return orig(self)
fo.FlatParams.zero_grad = zg
origs = fo.FlatParams.settle
def st(self):
    print("settle: sunk_now", len(self.sunk_now), "stale", len(self.stale), flush=True)
    return origs(self)
fo.FlatParams.settle = st
g = torch.Generator(device=dev).manual_seed(1234)
clean = 0.05 * torch.randn(2, 1, 16000, generator=g, device=dev)
noisy = clean + 0.05 * torch.randn(2, 1, 16000, generator=g, device=dev)
for _ in range(3):
    step(clean, noisy)
torch.cuda.synchronize()
