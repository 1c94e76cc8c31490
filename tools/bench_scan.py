"""Selective-scan micro-benchmark (GPU box): forward (no checkpoints), forward (training) and backward per shape.
  python tools/bench_scan.py                      # default shapes
  CUM_LIB=tools/_ab/lib_x.so python tools/bench_scan.py      # another build of the library (same-box A/B)
Prints one JSON line per shape."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    shapes = [(16, 2048, 64, 624, torch.bfloat16), (16, 2048, 16, 2499, torch.float32), (16, 2048, 8, 2499, torch.float32),
              (16, 2048, 8, 2499, torch.bfloat16), (128, 2048, 8, 2499, torch.float32), (128, 2048, 16, 2499, torch.float32),
              (16, 128, 16, 624, torch.float32), (256, 48, 8, 1875, torch.float32),
              (32, 2048, 8, 2499, torch.float32), (64, 2048, 8, 2499, torch.float32), (32, 2048, 16, 2499, torch.float32),
              (64, 2048, 16, 2499, torch.float32)]
    if len(sys.argv) > 1:
        shapes = shapes[:int(sys.argv[1])]
    for bsz, dim, N, L, io in shapes:
        t_i, t_b = bench._scan_case(dev, bsz, dim, N, L, io, os.environ.get("CUM_BENCH_BWD", "1") != "0")
        t_b = t_b if t_b is not None else -1.0
        sz = torch.empty((), dtype=io).element_size()
        byt = bsz * L * (sz * 4 * dim + 8 * N)
        print(json.dumps({"shape": [bsz, dim, N, L], "io": str(io), "fwd_ms": round(t_i, 4), "bwd_ms": round(t_b, 4),
                          "fwd_GBps": round(byt / t_i / 1e6, 1), "fwd_hbm_frac": round(byt / t_i / 1e6 / 8000, 4),
                          "lib": os.environ.get("CUM_LIB", "default")}), flush=True)


if __name__ == "__main__":
    main()
