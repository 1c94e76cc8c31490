"""Selective scan forward / backward on shapes given on the command line (GPU box): B,D,N,L[,f32|f16|bf16] ...
  python tools/bench_scan_shapes.py 16,2048,40,624 16,2048,24,624,f32
Run from another checkout's root (PYTHONPATH) to compare library generations on the same box."""
import json
import os
import sys

import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402

dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    parts = spec.split(",")
    bsz, dim, N, L = map(int, parts[:4])
    io = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[parts[4] if len(parts) > 4 else "f16"]
    t_i, t_b = bench._scan_case(dev, bsz, dim, N, L, io, True)
    print(json.dumps({"shape": [bsz, dim, N, L], "io": str(io), "fwd_ms": round(t_i, 4), "bwd_ms": round(t_b, 4), "cwd": os.getcwd()}), flush=True)
