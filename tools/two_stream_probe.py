"""Do two full-chip GEMM launches of the step overlap usefully on two streams (tail filling)?  For enc3..enc6: the forward
conv GEMM (gemm_nt9) and the weight-gradient GEMM (gemm_tn9 + reduce) of the same layer, back to back on one stream
against side by side on two streams.  GPU box only."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs

dev = torch.device("cuda")
dt = torch.float16
B = 16
Ts = [160254, 80126, 40062, 20030, 10014, 5006, 2502, 1250, 624]
Cs = [1, 64, 128, 256, 512, 768, 768, 768, 768]
s2 = torch.cuda.Stream()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return 1e3 * s.elapsed_time(e) / iters


for i in range(3, 8):
    M, Cin, H = B * (Ts[i + 1] + 2), Cs[i], Cs[i + 1]
    K, lda = 4 * Cin, 2 * Cin
    A = torch.randn(M * lda // 8 + K // 8 + 64, 8, device=dev).to(dt)
    W = (torch.randn(H, K, device=dev) / K ** 0.5).to(dt)
    out = torch.empty(M, H, device=dev, dtype=dt)
    bias = torch.zeros(H, device=dev)
    dz = torch.randn(M, H, device=dev).to(dt)
    nt = lambda: cs.gemm(A, 0, lda, W, bias, out, 0, H, M, 1 << 30, 1 << 30, hip.EPI_RELU, H)
    ws = torch.empty(max(hip.lib().cum_gemm_tn_workspace_elems(hip.dtype_code(dt), M, H, K), 1), dtype=torch.float32, device=dev)
    dW, db = torch.empty(H, K, device=dev), torch.empty(H, device=dev)
    import ctypes

    def tn():
        hip.check(hip.lib().cum_gemm_tn(hip.dtype_code(dt), M, H, K, hip.ptr(dz), H, hip.ptr(A), lda, hip.ptr(dW), K,
                                        hip.ptr(db), hip.ptr(ws), hip.stream_ptr()))

    def both_seq():
        nt()
        tn()

    def both_par():
        s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s2):
            tn()
        nt()
        torch.cuda.current_stream().wait_stream(s2)
    a, b, c, d = timeit(nt), timeit(tn), timeit(both_seq), timeit(both_par)
    print(f"enc{i}: nt {a:7.1f} us, tn {b:7.1f} us, one stream {c:7.1f} us, two streams {d:7.1f} us ({100 * (c - d) / c:+.1f} %)", flush=True)
