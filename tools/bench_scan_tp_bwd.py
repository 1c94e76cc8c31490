"""Time-parallel against sequential selective-scan backward on the small-grid shapes (GPU box; graph-replay timing as in
bench.py's scan rows): one JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cleanumamba_amd.mamba_ssm.ops import selective_scan_interface as ssi  # noqa: E402

dev = torch.device("cuda:0")
shapes = [("442K model B=16 D=128 N=16 L=624", 16, 128, 16, 624), ("442K model B=2", 2, 128, 16, 624),
          ("pruned-E8 block B=256 D=48 N=8 L=1875 (30 s)", 256, 48, 8, 1875), ("pruned-E8 block B=16, 10 s", 16, 48, 8, 624),
          ("pruned block d_inner 136 N=14 B=16", 16, 136, 14, 2499),
          ("pruned-E8 block B=32", 32, 48, 8, 1875), ("pruned-E8 block B=64", 64, 48, 8, 1875),
          ("pruned-E8 block B=128", 128, 48, 8, 1875), ("442K B=64 (128 groups)", 64, 128, 16, 624),
          ("442K B=128 (256 groups)", 128, 128, 16, 624)]
for name, bsz, dim, N, L in shapes:
    t_f, t_b = bench._scan_case(dev, bsz, dim, N, L, torch.float32, True)      # (times both forms itself)
    print(json.dumps({"shape": name, "fwd_ms": round(t_f, 4), "fwd_sequential_ms": bench._scan_case.sequential_ms,
                      "bwd_ms": round(t_b, 4), "bwd_sequential_ms": bench._scan_case.sequential_bwd_ms,
                      "bwd_speedup": None if not bench._scan_case.sequential_bwd_ms
                      else round(bench._scan_case.sequential_bwd_ms / t_b, 2)}), flush=True)
