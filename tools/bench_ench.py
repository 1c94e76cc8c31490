"""The width-128 encoder layer (E8 / E6 second layer) at the training shape: the fused forward (csrc/ench.hip) beside the
two GEMM launches it replaces, training form (hidden activation, sign nibbles and gate stored) and inference form.  GPU box.
  python tools/bench_ench.py [batch]          (CUM_LIB=other.so for a same-box A/B of two builds)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cleanumamba_amd.network import convstack as cs  # noqa: E402


def main():
    dev, dt = torch.device("cuda:0"), torch.float16
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    tin = bench.ENC_T[1]
    tout = (tin - 4) // 2 + 1
    g = torch.Generator(device=dev).manual_seed(0)
    w1 = (torch.randn(128, 64, 4, generator=g, device=dev) / 16).requires_grad_(False)
    b1 = 0.1 * torch.randn(128, generator=g, device=dev)
    w2 = torch.randn(256, 128, 1, generator=g, device=dev) / 11
    b2 = 0.1 * torch.randn(256, generator=g, device=dev)
    gi, gm, go = cs.Geo(B, tin, 64), cs.Geo(B, tout, 128), cs.Geo(B, tout, 128)
    x = (0.5 * torch.randn(gi.R, gi.Cp, generator=g, device=dev)).to(dt)

    def two(save):
        y1, _ = cs._conv_relu_fwd(x, w1, b1, gi, gm, want_bits=True) if save else (cs._conv_relu_fwd(x, w1, b1, gi, gm), None)
        return cs._glu_fwd(y1, w2, b2, gm, go, save)
    rows = go.M
    for save in (True, False):
        t_f = bench._time(lambda: cs._ench_fwd(x, w1, b1, w2, b2, gi, gm, go, save), iters=20, warm=5)
        t_t = bench._time(lambda: two(save), iters=20, warm=5)
        byt = rows * (128 + 256 + (256 + 32 + 256 if save else 0))
        print(f"{'training' if save else 'inference'} form, {rows} rows: fused {t_f * 1e3:7.1f} us = {byt / t_f / 1e9:5.2f} TB/s of its own "
              f"{byt / 1e6:.0f} MB; two launches {t_t * 1e3:7.1f} us; lib {os.environ.get('CUM_LIB', 'default')}")


def main_dec():
    """The width-128 decoder layer (second-to-last of E8 / E6): csrc/dech.hip beside its two launches."""
    dev, dt = torch.device("cuda:0"), torch.float16
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    t = (bench.ENC_T[1] - 4) // 2 + 1                 # the layer's input length = the second encoder layer's output
    g = torch.Generator(device=dev).manual_seed(1)
    w1 = torch.randn(256, 128, 1, generator=g, device=dev) / 11
    b1 = 0.1 * torch.randn(256, generator=g, device=dev)
    wt = torch.randn(128, 64, 4, generator=g, device=dev) / 16
    bt = 0.1 * torch.randn(64, generator=g, device=dev)
    gi, gg, go = cs.Geo(B, t, 128), cs.Geo(B, t, 128), cs.Geo(B, 2 * t + 2, 64)
    u = (0.5 * torch.randn(gi.R, gi.Cp, generator=g, device=dev)).to(dt)
    skip = (0.5 * torch.randn(go.R, go.Cp, generator=g, device=dev)).to(dt)

    def two(save):
        gb, _ = cs._glu_fwd(u, w1, b1, gi, gg, save)
        return cs._convt_fwd(gb, wt, bt, skip, gg, go, True)
    for save in (True, False):
        t_f = bench._time(lambda: cs._dech_fwd(u, w1, b1, wt, bt, skip, gi, gg, go, save), iters=20, warm=5)
        t_t = bench._time(lambda: two(save), iters=20, warm=5)
        byt = gg.M * (256 + 256 + 256 + (256 + 256 + 32 if save else 0))
        print(f"decoder, {'training' if save else 'inference'} form, {gg.M} rows: fused {t_f * 1e3:7.1f} us = {byt / t_f / 1e9:5.2f} TB/s of "
              f"its own {byt / 1e6:.0f} MB; two launches {t_t * 1e3:7.1f} us; lib {os.environ.get('CUM_LIB', 'default')}")


if __name__ == "__main__":
    main()
    main_dec()
