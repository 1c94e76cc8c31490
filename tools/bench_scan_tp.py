"""Time-parallel (csrc/scan_seg.hip) against sequential forward scan on the small-grid shapes, graph-replay timing.
usage: python tools/bench_scan_tp.py            (rocprofv3 --kernel-trace --stats -- python3 tools/bench_scan_tp.py for kernel times)"""
import sys

import torch

sys.path.insert(0, ".")
from bench import _scan_case  # noqa: E402

dev = torch.device("cuda")
for name, bsz, dim, N, L, io in [("E8 B=1", 1, 2048, 64, 624, torch.float16), ("E8 B=2", 2, 2048, 64, 624, torch.float16),
                                 ("E6 B=1", 1, 2048, 64, 2499, torch.float16),
                                 ("442K B=16", 16, 128, 16, 624, torch.float32), ("442K B=1", 1, 128, 16, 624, torch.float32),
                                 ("pruned B=256 D=48", 256, 48, 8, 1875, torch.float32),
                                 ("pruned B=1 D=48 30s", 1, 48, 8, 1875, torch.float32),
                                 ("D=2048 N=8 B=16", 16, 2048, 8, 2499, torch.float16)]:
    t, _ = _scan_case(dev, bsz, dim, N, L, io, False)
    seq = _scan_case.sequential_ms
    print(f"{name:24s} B={bsz} D={dim} N={N} L={L}: {t * 1e3:8.1f} us" +
          (f"   sequential {seq * 1e3:8.1f} us   x{seq / t:.2f}" if seq else "   (sequential path)"), flush=True)
