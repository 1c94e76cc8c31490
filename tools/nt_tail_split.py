"""Tail quantisation of the 256 x 256 GEMM (one workgroup per CU: a launch takes ceil(tiles / 256) rounds): one launch against
[whole rounds on 256 x 256 tiles] + [the remaining clips on 128 x 128 tiles, picked by the library for few-tile launches],
rows split at a clip boundary.  Replayed hipGraph, f16, EPI_BIAS.  GPU box."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd import hip
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda")
dt = torch.float16


def graph_time(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
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


B = 16
for name, T, N, K in (("enc4 conv", 5006, 768, 2048), ("enc4 1x1", 5006, 1536, 768), ("enc5 conv", 2502, 768, 3072),
                      ("enc5 1x1", 2502, 1536, 768), ("enc6 1x1", 1250, 1536, 768), ("enc3 conv", 10014, 512, 1024),
                      ("enc3 1x1", 10014, 1024, 512), ("dec3 dgrad", 5006, 768, 1536)):
    pitch = T + 2
    M = B * pitch
    A = torch.randn(M, K, device=dev).to(dt)
    W = (torch.randn(N, K, device=dev) / K ** 0.5).to(dt)
    bias = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev, dtype=dt)
    out2 = torch.empty_like(out)
    NB = (N + 255) // 256

    def run(o, m0, m):
        cs.gemm(A, m0 * K, K, W, bias, o, m0 * N, N, m, pitch, T, hip.EPI_BIAS, N)

    t_one = graph_time(lambda: run(out, 0, M))
    tiles = ((M + 255) // 256) * NB
    best = None
    res = []
    for c in range(1, B):
        m1 = c * pitch
        t1 = ((m1 + 255) // 256) * NB
        if t1 < 224:
            continue
        rounds = t1 / 256
        if rounds - int(rounds) > 0.15 and int(rounds + 0.999) - rounds > 0.12:
            continue                                  # first part would have a tail of its own
        t = graph_time(lambda: (run(out2, 0, m1), run(out2, m1, M - m1)))
        res.append((c, t1 / 256, t))
        if best is None or t < best[1]:
            best = (c, t)
    run(out, 0, M)
    if best:
        run(out2, 0, best[0] * pitch), run(out2, best[0] * pitch, M - best[0] * pitch)
        same = torch.equal(out, out2)
    else:
        same = None
    print(f"{name:11s} M={M} N={N} K={K}: tiles {tiles} = {tiles / 256:.2f} rounds, one launch {t_one:6.1f} us; split "
          + " ".join(f"{c}cl({r:.2f}r):{t:.1f}" for c, r, t in res) + f"  bitwise equal: {same}")
