import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleanumamba_amd.network import convstack as cs
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
for dt in (torch.float32, torch.float16):
    for M in (62, 124, 9984):
        for name, N, K in (("in", 4096, 512), ("x", 160, 2048), ("dt", 2048, 32), ("out", 512, 2048)):
            x = torch.randn(M, K, generator=g).to(dev).to(dt)
            w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
            dy = torch.randn(M, N, generator=g).to(dev).to(dt)
            y = cs.proj_fwd(x, w, dt)
            dx = cs.proj_dgrad(dy, w, dt)
            wr = w.to(dt).double()
            yr, dxr = x.double() @ wr.t(), dy.double() @ wr
            e1 = ((y.double() - yr).norm() / yr.norm()).item(); e2 = ((dx.double() - dxr).norm() / dxr.norm()).item()
            print(f"{str(dt):14s} M={M:5d} {name:4s} fwd {e1:.2e} dgrad {e2:.2e}")
