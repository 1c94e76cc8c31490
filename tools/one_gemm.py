"""One deep-layer data-gradient GEMM with the plain (EPI_BIAS) and the GLU-backward epilogue, a few launches each:
the subject of counter passes (tools/pmc_epilogue.sh).  GPU box only."""
import sys

import torch

sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs

dev = torch.device("cuda")
dt = torch.bfloat16
M, N, K = 80128, 768, 2048
g = torch.Generator(device=dev).manual_seed(0)
A = torch.randn(M * 1024 + K, device=dev, generator=g).to(dt)
W = (torch.randn(N, K, device=dev, generator=g) / K ** 0.5).to(dt)
ext = torch.randn(M, N, device=dev, generator=g).to(dt)
b = torch.randn(M, N, device=dev, generator=g).to(dt)
y = torch.randn(M, N, device=dev, generator=g).to(dt)
out1 = torch.empty(M, N, device=dev, dtype=dt)
out4 = torch.empty(M, 2 * N, device=dev, dtype=dt)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for _ in range(n):
    cs.gemm(A, 0, 1024, W, None, out1, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N)
for _ in range(n):
    cs.gemm(A, 0, 1024, W, None, out4, 0, 2 * N, M, 1 << 30, 1 << 30, hip.EPI_GLU_BWD, N, res=ext, r_off=0, ldr=N,
            aux=b, x_off=0, ldz=N, aux2=y, y_off=0, ldy=N, gate_only=True)
torch.cuda.synchronize()
