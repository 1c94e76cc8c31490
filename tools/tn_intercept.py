"""Fixed cost of one weight-gradient launch (gemm_tn9_kernel + tn_reduce_kernel): time against the number of 64-row
reduction steps per split at a fixed (N, K) and split count -- the intercept is prologue + slab store + slab reduce."""
import sys, torch
sys.path.insert(0, ".")
import bench
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda:0")
dt = torch.float16
for name, N, K, ldx in (("enc5.conv", 768, 3072, 1536), ("enc4.conv", 768, 2048, 1024), ("enc6.1x1", 1536, 768, 768)):
    tiles = (N // 256) * (K // 256)
    S = max(1, 256 // tiles)
    pts = []
    for steps in (4, 8, 16, 32, 48, 64, 96):
        M = S * 64 * steps
        dz = torch.randn(M, N, device=dev).to(dt)
        X = torch.randn(M * ldx // 8 + K // 8 + 64, 8, device=dev).to(dt)
        ms = bench._time(lambda: cs.wgrad(dz, 0, N, N, X, 0, ldx, K, M), iters=30)
        pts.append((steps, ms))
        del dz, X
    (s0, t0), (s1, t1) = pts[2], pts[-1]
    slope = (t1 - t0) / (s1 - s0)
    icpt = t0 - slope * s0
    fl_step = 2.0 * 64 * S * N * K
    print(name, "N", N, "K", K, "S", S, "slabs MB %.1f" % (S * N * K * 4 / 1e6),
          " ".join("%d:%.1fus" % (s, 1e3 * t) for s, t in pts),
          "| slope %.2f us/step = %.0f TF/s in the K loop, intercept %.1f us" % (1e3 * slope, fl_step / slope / 1e9, 1e3 * icpt), flush=True)
