"""Multi-resolution STFT loss on PyTorch-ROCm (rocFFT), as north_star leaves it.

Interface and arithmetic of src/util/stft_loss.py:16-184 (itself adapted from
ParallelWaveGAN): per resolution a spectral-convergence term
||Y - X||_F / ||Y||_F and a log-magnitude L1 term on sqrt(clamp(re^2 + im^2, 1e-7)),
averaged over resolutions and weighted by sc_lambda / mag_lambda.
"""
import torch
import torch.nn.functional as F


def stft(x, fft_size, hop_size, win_length, window):
    """(B, T) -> magnitude spectrogram (B, frames, fft_size // 2 + 1)."""
    spec = torch.stft(x, fft_size, hop_size, win_length, window, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2
    return torch.sqrt(torch.clamp(power, min=1e-7)).transpose(2, 1)


class SpectralConvergenceLoss(torch.nn.Module):
    def forward(self, x_mag, y_mag):
        return torch.norm(y_mag - x_mag, p="fro") / torch.norm(y_mag, p="fro")


class LogSTFTMagnitudeLoss(torch.nn.Module):
    def forward(self, x_mag, y_mag):
        return F.l1_loss(torch.log(y_mag), torch.log(x_mag))


class STFTLoss(torch.nn.Module):
    def __init__(self, fft_size=1024, shift_size=120, win_length=600, window="hann_window", band="full"):
        super().__init__()
        self.fft_size, self.shift_size, self.win_length, self.band = fft_size, shift_size, win_length, band
        self.spectral_convergence_loss = SpectralConvergenceLoss()
        self.log_stft_magnitude_loss = LogSTFTMagnitudeLoss()
        self.register_buffer("window", getattr(torch, window)(win_length))

    def forward(self, x, y):
        x_mag = stft(x, self.fft_size, self.shift_size, self.win_length, self.window)
        y_mag = stft(y, self.fft_size, self.shift_size, self.win_length, self.window)
        if self.band == "high":
            k = x_mag.shape[1] // 2
            x_mag, y_mag = x_mag[:, k:, :], y_mag[:, k:, :]
        elif self.band != "full":
            raise NotImplementedError
        return self.spectral_convergence_loss(x_mag, y_mag), self.log_stft_magnitude_loss(x_mag, y_mag)


class MultiResolutionSTFTLoss(torch.nn.Module):
    def __init__(self, fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240],
                 window="hann_window", sc_lambda=0.1, mag_lambda=0.1, band="full"):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.sc_lambda, self.mag_lambda = sc_lambda, mag_lambda
        self.stft_losses = torch.nn.ModuleList(
            [STFTLoss(fs, ss, wl, window, band) for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths)])

    def forward(self, x, y):
        if x.dim() == 3:
            x, y = x.reshape(-1, x.size(2)), y.reshape(-1, y.size(2))
        sc_loss, mag_loss = 0.0, 0.0
        for f in self.stft_losses:
            sc_l, mag_l = f(x, y)
            sc_loss = sc_loss + sc_l
            mag_loss = mag_loss + mag_l
        n = len(self.stft_losses)
        return sc_loss * self.sc_lambda / n, mag_loss * self.mag_lambda / n
