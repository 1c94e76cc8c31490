"""The weight gradients of the two widest layers (enc1 / dec6 at E8, B = 16: a 128 x 256 / 256 x 128 result over 641 024 rows),
timed per call through network/convstack.py::wgrad (GEMM + slab reduce).  With the AB library (CUM_LIB=tools/_ab/lib_ab.so),
CUM_TN_STREAM=0 runs the 128 x 128 kernel instead of gemm_tn_stream_kernel.  GPU box only.

usage: python tools/bench_tn_stream.py [bf16|f16]"""
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from cleanumamba_amd.network import convstack as cs  # noqa: E402

dt = torch.bfloat16 if "bf16" in sys.argv else torch.float16
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K, ldz, ldx) in [(641024, 128, 256, 128, 128), (641024, 256, 128, 256, 128), (641024, 128, 256, 128, 256),
                            (320512, 128, 256, 128, 256)]:
    dz = torch.randn(M * ldz + 64, generator=g, device=dev).to(dt)
    x = torch.randn((M - 1) * ldx + K + 64, generator=g, device=dev).to(dt)
    for _ in range(5):
        cs.wgrad(dz, 0, ldz, N, x, 0, ldx, K, M, want_bias=True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(50):
        cs.wgrad(dz, 0, ldz, N, x, 0, ldx, K, M, want_bias=True)
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) / 50 * 1e3
    by = 2 * (M * N + (M * ldx if ldx < K else M * K))
    print(json.dumps({"M": M, "N": N, "K": K, "ldx": ldx, "us": round(us, 1), "GBps": round(by / us * 1e-3, 0),
                      "hbm_frac": round(by / us * 1e-3 / 8000, 3), "TFLOPs": round(2 * M * N * K / us * 1e-6, 1),
                      "stream": os.environ.get("CUM_TN_STREAM", "1"), "lib": os.environ.get("CUM_LIB", "default")}))
