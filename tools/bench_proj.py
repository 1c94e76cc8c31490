"""Mamba projection GEMM shapes (E8, B=16): torch (hipBLASLt) against cum_gemm_nt on the same operands.  GPU box."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda")
dt = torch.float16 if "f16" in sys.argv else torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
M = 9984
for name, N, K in (("in_proj fwd", 4096, 512), ("in_proj dgrad", 512, 4096), ("out_proj fwd", 512, 2048), ("out_proj dgrad", 2048, 512),
                   ("x_proj fwd", 160, 2048), ("x_proj dgrad", 2048, 192), ("dt_proj dgrad", 64, 2048), ("tsfm_conv1", 512, 768),
                   ("tsfm_conv2", 768, 512), ("enc7 conv", 768, 3072), ("enc7 1x1", 1536, 768)):
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t_lib = timeit(lambda: torch.matmul(A, W.t(), out=out))
    Wp = torch.zeros(cs.rup(N, 32), K, device=dev, dtype=dt); Wp[:N] = W
    bias = torch.zeros(Wp.shape[0], device=dev)
    t_own = timeit(lambda: cs.gemm(A, 0, K, Wp, bias, out, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N))
    t_sk = timeit(lambda: cs.gemm(A, 0, K, Wp, bias, out, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N, split_k=True)) if K >= 256 else float("nan")
    ref = A.float() @ W.float().t()
    err = float((out.float() - ref).abs().max())
    print(f"{name:16s} M={M} N={N:5d} K={K:5d}  hipBLASLt {t_lib:6.1f} us   cum_gemm_nt {t_own:6.1f} us ({2e-6 * M * N * K / t_own:5.0f} TF/s)   64x64 split-K {t_sk:6.1f} us   max err {err:.3f}")
