"""Times the scan forward / backward kernels at the E8 train shape (GPU box)."""
import sys
import torch
sys.path.insert(0, ".")
from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
dev = torch.device("cuda")
bsz, dim, N, L = 16, 2048, 64, 624
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)
xz = rn(bsz, L, 2 * dim).requires_grad_(True)
delta_ = (0.3 * rn(bsz, L, dim)).requires_grad_(True)
A = (-torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float())[None].repeat(dim, 1))).requires_grad_(True)
xd = rn(bsz, L, 32 + 2 * N).requires_grad_(True)
D, bias = rn(dim).requires_grad_(True), (0.3 * rn(dim)).requires_grad_(True)
def fwd():
    u, z = xz[..., :dim].transpose(1, 2), xz[..., dim:].transpose(1, 2)
    return selective_scan_fn(u, delta_.transpose(1, 2), A, xd[..., 32:32 + N].transpose(1, 2), xd[..., 32 + N:].transpose(1, 2), D, z=z, delta_bias=bias, delta_softplus=True)
dout = rn(bsz, dim, L)
def fwd_nograd():
    with torch.no_grad():
        return fwd()
cases = (("fwd(no ckpt)", fwd_nograd),) if "fwdonly" in sys.argv else \
    (("fwd(no ckpt)", fwd_nograd), ("fwd(+ckpt)", lambda: fwd()), ("fwd+bwd", lambda: fwd().backward(dout)))
for name, fn in cases:
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name}: {s.elapsed_time(e) / 10:.3f} ms")
