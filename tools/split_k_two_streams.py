"""in_proj data gradient (M=9984, N=512, K=4096, f16): ONE cum_gemm_nt launch (312 workgroups of 128 x 128: 1.2 per CU)
against its two K halves as two launches on two streams (624 workgroups co-resident) + the sum, all replayed from a
hipGraph.  GPU box."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda")
dt = torch.float16
M = 9984


def graph_time(fn, reps=20):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / 5 / reps * 1e3


for name, N, K in (("in_proj dgrad", 512, 4096), ("out_proj fwd", 512, 2048), ("enc7 conv", 768, 3072)):
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    oa, ob = torch.empty_like(out), torch.empty_like(out)
    side = torch.cuda.Stream()
    h = K // 2

    def one():
        cs.gemm(A, 0, K, W, bias, out, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N)

    def two():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        cs.gemm(A, 0, K, W[:, :h], bias, oa, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N)
        with torch.cuda.stream(side):
            cs.gemm(A, h, K, W[:, h:], None, ob, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N)
        cur.wait_stream(side)
        torch.add(oa, ob, out=out)

    def two_res():      # second half adds the first in its epilogue (sequential: no concurrency, for reference)
        cs.gemm(A, 0, K, W[:, :h], bias, oa, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N)
        cs.gemm(A, h, K, W[:, h:], None, out, 0, N, M, 1 << 30, 1 << 30, hip.EPI_BIAS, N, res=oa, r_off=0, ldr=N)

    one()
    ref = out.float().clone()
    two()
    err = float((out.float() - ref).abs().max())
    t1, t2 = graph_time(one), graph_time(two)
    try:
        t3 = graph_time(two_res)
    except Exception as ex:
        t3 = float("nan")
    print(f"{name:14s} N={N} K={K}: one launch {t1:6.1f} us   two K halves on two streams + add {t2:6.1f} us   sequential halves {t3:6.1f} us   max diff {err:.4f}")
