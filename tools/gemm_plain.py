"""One plain 8192^3 (or argv[1]^3) 16-bit GEMM on cum_gemm_nt, a few launches: the target of tools/pmc_gemm_plain.sh."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda")
A = torch.randn(n * n // 8 + n // 8 + 64, 8, device=dev).to(torch.bfloat16)
W = (torch.randn(n, n, device=dev) / n ** 0.5).to(torch.bfloat16)
out = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
bias = torch.zeros(n, device=dev)
for _ in range(6):
    cs.gemm(A, 0, n, W, bias, out, 0, n, n, 1 << 30, 1 << 30, hip.EPI_BIAS, n)
torch.cuda.synchronize()
if "time" in sys.argv:
    import os
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20):
        cs.gemm(A, 0, n, W, bias, out, 0, n, n, 1 << 30, 1 << 30, hip.EPI_BIAS, n)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 20 * 1e3
    print({"n": n, "us": round(us, 1), "PFLOPs": round(2 * n ** 3 / us * 1e-9, 3), "group_m": os.environ.get("CUM_NT_GROUPM", "default")})
