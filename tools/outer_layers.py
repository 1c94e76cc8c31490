"""The two fused outer layers at the E8 / B=16 training shapes, forward + backward, a few times (for rocprofv3 passes):
first encoder layer (csrc/enc0.hip) through EncoderStack, last decoder layer (csrc/dec7.hip) through a two-layer
DecoderStack (dec6 supplies the ReLU sign bits)."""
import sys, torch
sys.path.insert(0, ".")
import bench
from cleanumamba_amd.network import Net, convstack as cs
dev = torch.device("cuda:0")
dt = torch.float16
torch.manual_seed(0)
net = Net("CleanUMamba", bench.E8).to(dev).train()
B, T = bench.B16, bench.ENC_T
enc = net.encoder[0]
g_in, g_mid, g_out = cs.Geo(B, T[0], 1), cs.Geo(B, T[1], 64), cs.Geo(B, T[1], 64)
x = cs.to_rows(0.5 * torch.randn(B, 1, T[0], device=dev), g_in, dt)
d6, d7 = net.decoder[-2], net.decoder[-1]
g0 = (cs.Geo(B, T[2], 128), cs.Geo(B, T[2], 128), cs.Geo(B, T[1], 64))
g1 = (cs.Geo(B, T[1], 64), cs.Geo(B, T[1], 64), cs.Geo(B, 2 * T[1] + 2, 1))
u = cs.to_rows(0.5 * torch.randn(B, 128, T[2], device=dev), g0[0], dt).requires_grad_(True)
skip = cs.to_rows(0.5 * torch.randn(B, 64, T[1], device=dev), g0[2], dt).requires_grad_(True)
ep = [enc[0].weight, enc[0].bias, enc[2].weight, enc[2].bias]
dp = [d6[0].weight, d6[0].bias, d6[2].weight, d6[2].bias, d7[0].weight, d7[0].bias, d7[2].weight, d7[2].bias]
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    (y,) = cs.EncoderStack.apply(x, [(g_in, g_mid, g_out)], True, *ep)
    y.float().sum().backward()
    o = cs.DecoderStack.apply(u, [g0, g1], True, 1, skip, *dp)
    o.float().sum().backward()
    for p in ep + dp:
        p.grad = None
    u.grad = skip.grad = None
torch.cuda.synchronize()
print("ok")
