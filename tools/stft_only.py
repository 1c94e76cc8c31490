"""The multi-resolution STFT loss alone at the training shape (16 clips x 160 000 samples, the reference's three
resolutions), forward + backward, a few times -- for rocprofv3 passes (tools/pmc_stft.sh).  CUM_STFT_FUSED=0 selects the
rocFFT route."""
import sys

import torch

sys.path.insert(0, ".")
from cleanumamba_amd.util.stft_loss import MultiResolutionSTFTLoss  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
clean = 0.05 * torch.randn(16, 160000, generator=g, device=dev)
den = (clean + 0.05 * torch.randn(16, 160000, generator=g, device=dev)).requires_grad_(True)
mr = MultiResolutionSTFTLoss(sc_lambda=0.5, mag_lambda=0.5, band="full", hop_sizes=[50, 120, 240],
                             win_lengths=[240, 600, 1200], fft_sizes=[512, 1024, 2048]).to(dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    sc, mag = mr(den, clean)
    (sc + mag).backward()
    den.grad = None
torch.cuda.synchronize()
print("ok")
