"""cum_gemm_nt against torch.matmul on a few shapes (GPU box); CUM_NT_TILE pins the tile variant."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda")
for dt in (torch.bfloat16, torch.float32):
    for M, N, K in [(300, 64, 64), (300, 128, 256), (300, 256, 256), (300, 512, 128), (1000, 192, 512), (5000, 768, 3072), (70000, 1024, 512)]:
        torch.manual_seed(0)
        A = torch.randn(M, K, device=dev).to(dt)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
        bias = torch.randn(N, device=dev)
        out = torch.full((M, N), float("nan"), device=dev, dtype=dt)
        cs.gemm(A, 0, K, W, bias, out, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N)
        ref = A.float() @ W.float().t() + bias
        err = (out.float() - ref).abs().max().item()
        bad = (~torch.isfinite(out.float())).sum().item()
        worst = ((out.float() - ref).abs() > 0.05).nonzero()
        print(dt, M, N, K, "max err %.3e nonfinite %d" % (err, bad), "first bad", worst[:3].tolist() if len(worst) else "-", "nbad", len(worst))
