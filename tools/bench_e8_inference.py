import sys, time, torch
sys.path.insert(0, ".")
import bench
from cleanumamba_amd.network import Net
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", bench.E8).to(dev).eval()
for B, L in ((1, 160000), (1, 16000), (4, 160000)):
    x = 0.05 * torch.randn(B, 1, L, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        for _ in range(5): net(x)
        torch.cuda.synchronize(); t0 = time.time()
        for _ in range(30): net(x)
        torch.cuda.synchronize()
    print("E8 forward f16 B=%d L=%d: %.3f ms" % (B, L, (time.time() - t0) / 30 * 1e3))
