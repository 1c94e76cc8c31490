"""Gradient of the mean L1 loss on a 2-clip batch vs the average of the two 1-clip gradients (f32, real E8): what the
2-rank test compares, without the ranks.  Prints the per-parameter relative error, worst first."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth
from cleanumamba_amd.network import Net
from cleanumamba_amd.util.util import loss_fn
E8 = dict(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8, kernel_size=4,
          stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)
dev = torch.device("cuda:0")
torch.manual_seed(1000)
net = Net("CleanUMamba", E8).to(dev).train()
parts = [synth.waveform(1, 16000, seed=500 + r) for r in range(2)]
clean = torch.cat([p[0] for p in parts]).to(dev); noisy = torch.cat([p[1] for p in parts]).to(dev)
def grads(c, n):
    net.zero_grad(set_to_none=True)
    loss, _ = loss_fn(net, (c, n), stft_lambda=0)
    loss.backward()
    return {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}, float(loss)
with torch.no_grad():
    y2 = net(noisy); y0 = net(noisy[:1]); y1 = net(noisy[1:])
print("forward batch-invariant:", torch.equal(y2[:1], y0), torch.equal(y2[1:], y1), float((y2[:1] - y0).abs().max()))
g2, l2 = grads(clean, noisy)
g0, l0 = grads(clean[:1], noisy[:1]); g1, l1 = grads(clean[1:], noisy[1:])
print("loss", l2, (l0 + l1) / 2)
errs = sorted(((((g0[k] + g1[k]) / 2 - g2[k]).norm() / g2[k].norm().clamp_min(1e-30)).item(), k) for k in g2)
for e, k in errs[::-1][:12]:
    print(f"{e:.3e} {k}")
g2b, _ = grads(clean, noisy)
print("run-to-run (same batch):", max(((g2b[k] - g2[k]).norm() / g2[k].norm().clamp_min(1e-30)).item() for k in g2))
