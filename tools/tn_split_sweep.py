"""Split-count sweep of the 256x256 weight-gradient GEMM over the MFMA-bound shapes (CUM_LIB=tools/_ab/lib_ab.so; the
knob is read per call, so one process sweeps)."""
import os, sys
import torch
sys.path.insert(0, ".")
import bench
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda")
dt = torch.float16
shapes = [("enc3.conv", 160256, 512, 1024, 512), ("enc3.1x1", 160256, 1024, 512, 512), ("enc4.conv", 80128, 768, 2048, 1024),
          ("enc4.1x1", 80128, 1536, 768, 768), ("enc5.conv", 40064, 768, 3072, 1536), ("enc5.1x1", 40064, 1536, 768, 768),
          ("enc6.conv", 20032, 768, 3072, 1536), ("enc6.1x1", 20032, 1536, 768, 768),
          ("enc7.conv", 10016, 768, 3072, 1536), ("enc7.1x1", 10016, 1536, 768, 768)]
for name, M, N, K, ldx in shapes:
    dz = torch.randn(M, N, device=dev).to(dt)
    X = torch.randn(M * ldx // 8 + K // 8 + 64, 8, device=dev).to(dt)
    tiles = (N // 256) * (K // 256)
    d = 256 // tiles
    res = {}
    os.environ.pop("CUM_TN_SPLITS", None)
    res["def"] = bench._time(lambda: cs.wgrad(dz, 0, N, N, X, 0, ldx, K, M))
    for s in sorted(set(max(1, int(d * f)) for f in (0.25, 0.33, 0.4, 0.5, 0.6, 0.67, 0.75, 0.8, 0.86, 0.9, 1.0, 1.15, 1.3, 1.5, 2.0))):
        os.environ["CUM_TN_SPLITS"] = str(s)
        res[s] = bench._time(lambda: cs.wgrad(dz, 0, N, N, X, 0, ldx, K, M))
    os.environ.pop("CUM_TN_SPLITS", None)
    res["def2"] = bench._time(lambda: cs.wgrad(dz, 0, N, N, X, 0, ldx, K, M))
    best = min((v, k) for k, v in res.items())
    print(f"{name} M={M} tiles={tiles} default S={d}: " + " ".join(f"{k}:{1e3 * v:.1f}" for k, v in res.items()) + f"  best {best[1]}")
    del dz, X
