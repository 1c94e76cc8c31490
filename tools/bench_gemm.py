"""Micro-benchmark of cum_gemm_nt / cum_gemm_tn on the E8 (B=16) layer shapes, beside torch.matmul
(hipBLASLt) on the same plain shapes as a known-good reference on the same device.  GPU box only."""
import sys
import time

import torch

sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs

dev = torch.device("cuda")
dt = torch.bfloat16 if "f32" not in sys.argv else torch.float32
B = 16
Ts = [160254, 80126, 40062, 20030, 10014, 5006, 2502, 1250, 624]
Cs = [1, 64, 128, 256, 512, 768, 768, 768, 768]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def run(name, M, N, K, lda, epi):
    esz = 2 if dt == torch.bfloat16 else 4
    A = torch.randn(M * lda // 8 + K // 8 + 64, 8, device=dev).to(dt)
    Kp = cs.rup(K, cs.bk_of(dt))
    W = (torch.randn(cs.rup(N, 32), Kp, device=dev) / K ** 0.5).to(dt)
    nout = N // 2 if epi == hip.EPI_GLU else N
    out = torch.empty(M, nout, device=dev, dtype=dt)
    bias = torch.zeros(W.shape[0], device=dev)
    f = lambda: cs.gemm(A, 0, lda, W, bias, out, 0, nout, M, 1 << 30, 1 << 30, epi, nout)
    ms = timeit(f)
    fl = 2.0 * M * N * K
    byt = (M * lda + M * nout) * esz
    line = f"{name:14s} M={M:8d} N={N:5d} K={K:5d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s  {byt / ms / 1e6:7.0f} GB/s"
    if lda >= K:
        A2 = A.view(-1)[: M * lda].view(M, lda)[:, :K]
        W2 = W[:N, :K]
        ms2 = timeit(lambda: torch.matmul(A2, W2.t()))
        line += f"   | hipBLASLt {ms2 * 1e3:8.1f} us {fl / ms2 / 1e9:7.1f} TF/s"
    print(line, flush=True)


if "tn" in sys.argv:
    for i in range(8):
        M, Cin, H = B * (Ts[i + 1] + 2), cs.rup(Cs[i], 8), Cs[i + 1]
        for name, N, K, ldx in ((f"enc{i}.conv.w", H, 4 * Cin, 2 * Cin), (f"enc{i}.1x1.w", 2 * H, H, H)):
            dz = torch.randn(M, N, device=dev).to(dt)
            X = torch.randn(M * ldx // 8 + K // 8 + 64, 8, device=dev).to(dt)
            f = lambda: cs.wgrad(dz, 0, N, N, X, 0, ldx, K, M)
            ms = timeit(f)
            fl = 2.0 * M * N * K
            print(f"{name:14s} M={M:8d} N={N:5d} K={K:5d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)
else:
    for i in range(8):
        M, Cin, H = B * (Ts[i + 1] + 2), cs.rup(Cs[i], 8), Cs[i + 1]
        run(f"enc{i}.conv", M, H, 4 * Cin, 2 * Cin, hip.EPI_RELU)
        run(f"enc{i}.1x1glu", M, 2 * H, H, H, hip.EPI_GLU)
    run("plain 8k", 8192, 8192, 8192, 8192, hip.EPI_BIAS)
    run("plain 4k", 4096, 4096, 4096, 4096, hip.EPI_BIAS)
