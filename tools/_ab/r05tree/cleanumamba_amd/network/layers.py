"""GLU activation module -- interface of src/network/layers.py:6-41."""
import torch
import torch.nn as nn


class Activation(nn.Module):
    r"""Gated linear unit over channel halves: ``A * act(B)`` with A the first half of the
    channels and B the second; ``bypass_channels`` leading channels pass through untouched."""

    def __init__(self, activation="Sigmoid", bypass_channels=0) -> None:
        super().__init__()
        assert activation in ["Sigmoid", "ReLU", "SiLU", "GELU"], f"activation={activation}"
        self.bypass_channels = bypass_channels
        self.activation = {"SiLU": nn.SiLU, "ReLU": nn.ReLU, "GELU": nn.GELU, "Sigmoid": nn.Sigmoid}[activation]()

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        nX = self.bypass_channels
        nAB = (input.shape[1] - nX) // 2
        if nX == 0:
            A, B = torch.split(input, [nAB, nAB], 1)
            return A * self.activation(B)
        X, A, B = torch.split(input, [nX, nAB, nAB], 1)
        assert A.shape == B.shape, f"A.shape={A.shape}, B.shape={B.shape}"
        return torch.cat([X, A * self.activation(B)], 1)
