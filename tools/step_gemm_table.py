"""Every cum_gemm_nt / cum_gemm_tn call of ONE eager E8 train step, timed on its own (events around the call,
launch order): shape, epilogue, time, TFLOP/s and the GB/s of the operands the call must move.  GPU box only.

usage: python tools/step_gemm_table.py [bf16|f16] > gpurun_out/gemm_table.txt
"""
import sys

import torch

sys.path.insert(0, ".")
from bench import CLIP, E8  # noqa: E402
from cleanumamba_amd.network import Net  # noqa: E402
from cleanumamba_amd.network import convstack as cs  # noqa: E402
from cleanumamba_amd.training.train_step import TrainStep  # noqa: E402

dt = torch.float16 if "f16" in sys.argv else torch.bfloat16
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", E8).to(dev).train()
step = TrainStep(net, autocast_dtype=dt, use_graph=False)
g = torch.Generator(device=dev).manual_seed(1234)
clean = 0.05 * torch.randn(16, 1, CLIP, generator=g, device=dev)
noisy = clean + 0.05 * torch.randn(16, 1, CLIP, generator=g, device=dev)
for _ in range(3):
    step(clean, noisy)
torch.cuda.synchronize()

rows = []
_gemm, _wgrad = cs.gemm, cs.wgrad


def timed(fn):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    r = fn()
    e.record()
    torch.cuda.synchronize()
    return r, s.elapsed_time(e)


def gemm(A, a_off, lda, Wp, bias, out, o_off, ldc, M, pitch, valid, epilogue, n_store, res=None, r_off=0, ldr=0,
         aux=None, x_off=0, ldz=0, **kw):
    _, ms = timed(lambda: _gemm(A, a_off, lda, Wp, bias, out, o_off, ldc, M, pitch, valid, epilogue, n_store, res=res,
                                r_off=r_off, ldr=ldr, aux=aux, x_off=x_off, ldz=ldz, **kw))
    N, K = Wp.shape
    esz = A.element_size()
    byt = M * (lda + n_store) * esz
    if res is not None:
        byt += M * ldr * (esz if res.dtype == A.dtype else 0.25)
    if aux is not None:
        byt += M * ldz * (esz if aux.dtype == A.dtype else 0.25)
    if kw.get("aux2") is not None:
        byt += M * kw["ldy"] * esz
    rows.append(("nt", epilogue, M, N, K, ms, 2.0 * M * N * K, byt))


def wgrad(dZ, z_off, ldz, N, X, x_off, ldx, K, M, **kw):
    r, ms = timed(lambda: _wgrad(dZ, z_off, ldz, N, X, x_off, ldx, K, M, **kw))
    rows.append(("tn", -1, M, N, K, ms, 2.0 * M * N * K, M * (ldz + ldx) * dZ.element_size()))
    return r


cs.gemm, cs.wgrad = gemm, wgrad
import cleanumamba_amd.mamba_ssm.modules.mamba_simple as ms_mod  # noqa: E402
for mod in (ms_mod,):
    for name in ("gemm", "wgrad"):
        if hasattr(mod, name):
            setattr(mod, name, locals()[name])
step(clean, noisy)
torch.cuda.synchronize()
tot = {}
print(f"{'kind':4s} {'epi':>3s} {'M':>8s} {'N':>5s} {'K':>5s} {'us':>8s} {'TF/s':>7s} {'GB/s':>6s}")
for kind, epi, M, N, K, ms, fl, byt in rows:
    print(f"{kind:4s} {epi:3d} {M:8d} {N:5d} {K:5d} {ms * 1e3:8.1f} {fl / ms / 1e9:7.1f} {byt / ms / 1e6:6.0f}")
    tot[kind] = tot.get(kind, 0.0) + ms
print({k: round(v, 2) for k, v in tot.items()}, "ms per step; calls:", len(rows))
