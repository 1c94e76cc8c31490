"""Per-shape timing of the weight-gradient GEMM on the small-M shapes under AB knobs (CUM_LIB=tools/_ab/lib_ab.so)."""
import os, sys
import torch
sys.path.insert(0, ".")
import bench
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda")
dt = torch.float16
shapes = [("enc5.1x1", 40064, 1536, 768, 768), ("enc6.conv", 20032, 768, 3072, 1536), ("enc6.1x1", 20032, 1536, 768, 768),
          ("enc7.conv", 10016, 768, 3072, 1536), ("enc7.1x1", 10016, 1536, 768, 768), ("enc4.1x1", 80128, 1536, 768, 768)]
out = []
for name, M, N, K, ldx in shapes:
    dz = torch.randn(M, N, device=dev).to(dt)
    X = torch.randn(M * ldx // 8 + K // 8 + 64, 8, device=dev).to(dt)
    ms = bench._time(lambda: cs.wgrad(dz, 0, N, N, X, 0, ldx, K, M))
    out.append(f"{name} {1e3 * ms:6.1f}us {2.0 * M * N * K / ms / 1e9 / 2500:.3f}")
print(os.environ.get("CUM_TN8", "-"), os.environ.get("CUM_TN_SPLITS", "-"), " | ".join(out))
