"""Where the d_state-64 backward scan spends its cycles (GPU box).  Needs a library built with -DCUM_SCAN_PROBE
(csrc/scan_bwd.hip PROBE): wave 0 of every workgroup sums s_memtime deltas per phase, and the totals leave through the
delta-bias slab, so the op's ddelta_bias output holds, per channel group, the cycles of phase i in channel i.
  CUM_LIB=tools/_ab/lib_probe.so python tools/scan_phase_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

NAMES = ["phase A + next rows issue", "barrier", "checkpoint wait + fwd recompute (2nd half)", "reverse walk (2nd half)",
         "barrier", "phase C (2nd half)", "barrier", "fwd recompute (1st half)", "reverse walk (1st half)", "barrier",
         "phase C (1st half)", "-"]


def main():
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    dev = torch.device("cuda:0")
    bsz, dim, Ns, L, io = 16, 2048, 64, 624, torch.float16
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    R = dim // 64
    xz = rn(bsz, L, 2 * dim).to(io)
    u = xz[..., :dim].transpose(1, 2).requires_grad_(True)
    z = xz[..., dim:].transpose(1, 2).requires_grad_(True)
    dl = (0.3 * rn(bsz, L, dim)).to(io).transpose(1, 2).requires_grad_(True)
    Am = (-torch.exp(torch.log(torch.arange(1, Ns + 1, device=dev).float())[None].repeat(dim, 1))).requires_grad_(True)
    xd = rn(bsz, L, R + 2 * Ns)
    Bm = xd[..., R:R + Ns].transpose(1, 2).requires_grad_(True)
    Cm = xd[..., R + Ns:].transpose(1, 2).requires_grad_(True)
    Dv, bv = rn(dim).requires_grad_(True), (0.3 * rn(dim)).requires_grad_(True)
    dout = rn(bsz, L, dim).to(io).transpose(1, 2)
    leaves = (u, z, dl, Am, Bm, Cm, Dv, bv)
    out = selective_scan_fn(u, dl, Am, Bm, Cm, Dv, z=z, delta_bias=bv, delta_softplus=True)
    for _ in range(3):
        grads = torch.autograd.grad(out, leaves, dout, retain_graph=True)
    torch.cuda.synchronize()
    t = bench._time(lambda: torch.autograd.grad(out, leaves, dout, retain_graph=True))
    dbias = grads[-1].float().view(dim // 64, 64)[:, :12].mean(0) / bsz     # per workgroup, whole sequence
    nchunks = (L + 15) // 16
    tot = dbias.sum().item()
    print(f"backward {t:.4f} ms per launch; cycles per chunk and workgroup (wave 0): total {tot / nchunks:.0f}")
    for n, v in zip(NAMES, dbias.tolist()):
        print(f"  {n:40s} {v / nchunks:9.0f}  {100 * v / tot:5.1f} %")


if __name__ == "__main__":
    main()
