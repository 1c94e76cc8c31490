"""GPU fused STFT loss vs f32 / f64 CPU oracle: which differences are f32 noise."""
import sys
import torch
sys.path.insert(0, ".")
from oracle import cleanumamba_ref as R
from cleanumamba_amd.util.stft_loss import MultiResolutionSTFTLoss
rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
for B, L in [(2, 16000), (2, 48000), (4, 160000)]:
    g = torch.Generator().manual_seed(L)
    clean = 0.05 * torch.randn(B, L, generator=g); den = clean + 0.05 * torch.randn(B, L, generator=g)
    mr = MultiResolutionSTFTLoss(sc_lambda=0.5, mag_lambda=0.5, hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200], fft_sizes=[512, 1024, 2048]).cuda()
    res = {}
    for name, wsc, wmag in (("sc", 1.0, 0.0), ("mag", 0.0, 1.0)):
        xg = den.cuda().requires_grad_(True); sc, mag = mr(xg, clean.cuda()); (wsc * sc + wmag * mag).backward()
        x32 = den.clone().requires_grad_(True); a, b = R.mrstft_loss_ref(x32, clean); (wsc * a + wmag * b).backward()
        x64 = den.double().requires_grad_(True); a64, b64 = R.mrstft_loss_ref(x64, clean.double()); (wsc * a64 + wmag * b64).backward()
        print(B, L, name, "gpu-vs-f64 %.2e  cpu32-vs-f64 %.2e  gpu-vs-cpu32 %.2e" % (rel(xg.grad, x64.grad), rel(x32.grad, x64.grad), rel(xg.grad, x32.grad)),
              "values", sc.item(), a64.item(), mag.item(), b64.item())
