"""Oracle (test infrastructure, BUILD CONTAINER ONLY): generate tests/golden/*.npz.

Run from the repo root:  python -m oracle.make_golden
Needs /root/reference (imports the reference class through oracle/reference_shim
and reads its shipped checkpoints).  Writes only data: seeded inputs, expected
outputs, and checkpoint weights (fp16, as shipped).  No reference source text is
stored.

What is pinned by what:
  e2e_*  : produced by the REFERENCE class (src/network/CleanUMamba.py) with the
           Mamba block supplied by oracle/mamba_ref.py; an fp64 recompute by
           oracle/cleanumamba_ref.py rides along (out64).
  scan_*, dwconv_*, step : produced by oracle/mamba_ref.py (fp32 and fp64); these
           ops are third-party in the reference (mamba-ssm 1.2.2) - parity unpinned
           against the CUDA kernels, cross-checked against HF transformers in
           tests/test_oracle_golden.py.
  loss   : produced by the reference's src/util/stft_loss.py + F.l1_loss
           (src/util/util.py:313-322).
  ckpt_* / e2e_* of the other seven pruned checkpoints (`python -m oracle.make_golden pruned`): same recipe as
           pruned500k -- every loadable checkpoint the reference ships (src/examples/loading_pretrained_models.py:7-19).
  lr_schedule : values returned by the reference's LinearWarmupCosineDecay.step() (src/util/util.py:115-161), fresh
           and resumed runs, past the wrap-around (`python -m oracle.make_golden lr`).
"""
import json
import os

import numpy as np
import torch

from . import cleanumamba_ref as R
from . import mamba_ref as M
from . import reference_shim, synth

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REF = reference_shim.REFERENCE_ROOT


def npf(t):
    return t.detach().cpu().numpy()


def make_scan(idx, bsz, dim, N, L, with_z=True, with_bias=True, with_D=True, seed=0):
    g = torch.Generator().manual_seed(1000 + seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    u, delta = rn(bsz, dim, L), 0.5 * rn(bsz, dim, L)
    A = -torch.exp(torch.log(torch.arange(1, N + 1).float())[None].repeat(dim, 1) + 0.1 * rn(dim, N))
    Bm, Cm = rn(bsz, N, L), rn(bsz, N, L)
    D = rn(dim) if with_D else None
    z = rn(bsz, dim, L) if with_z else None
    bias = 0.5 * rn(dim) if with_bias else None
    dout = rn(bsz, dim, L)
    dlast = 0.1 * rn(bsz, dim, N)
    rec = dict(u=u, delta=delta, A=A, B=Bm, C=Cm, dout=dout)
    if with_D:
        rec["D"] = D
    if with_z:
        rec["z"] = z
    if with_bias:
        rec["delta_bias"] = bias
    out = {k: npf(v) for k, v in rec.items()}
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        ins = {k: v.detach().clone().to(dt).requires_grad_(True) for k, v in rec.items() if k != "dout"}
        y, last = M.selective_scan_ref(ins["u"], ins["delta"], ins["A"], ins["B"], ins["C"],
                                       ins.get("D"), z=ins.get("z"), delta_bias=ins.get("delta_bias"),
                                       delta_softplus=True, return_last_state=True)
        (y * dout.to(dt)).sum().backward()
        out["out" + tag] = npf(y)
        out["last" + tag] = npf(last)
        for k, v in ins.items():
            out[f"d{k}{tag}"] = npf(v.grad)
    np.savez_compressed(os.path.join(OUT, f"scan_{idx}.npz"), **out)


def make_dwconv(idx, bsz, dim, L, W=4, seed=0):
    g = torch.Generator().manual_seed(2000 + seed)
    x, w, b = torch.randn(bsz, dim, L, generator=g), 0.5 * torch.randn(dim, W, generator=g), \
        0.2 * torch.randn(dim, generator=g)
    dout = torch.randn(bsz, dim, L, generator=g)
    out = dict(x=npf(x), w=npf(w), b=npf(b), dout=npf(dout))
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        xi, wi, bi = (t.detach().clone().to(dt).requires_grad_(True) for t in (x, w, b))
        y = M.causal_conv1d_ref(xi, wi, bi, "silu")
        (y * dout.to(dt)).sum().backward()
        out["y" + tag], out["dx" + tag], out["dw" + tag], out["db" + tag] = \
            npf(y), npf(xi.grad), npf(wi.grad), npf(bi.grad)
    np.savez_compressed(os.path.join(OUT, f"dwconv_{idx}.npz"), **out)


def make_step(seed=0):
    g = torch.Generator().manual_seed(3000 + seed)
    bsz, dim, N, W, steps = 3, 48, 13, 4, 5
    rn = lambda *s: torch.randn(*s, generator=g)
    A = -torch.exp(0.3 * rn(dim, N))
    D, bias, w, cb = rn(dim), 0.5 * rn(dim), 0.5 * rn(dim, W), 0.2 * rn(dim)
    xs, dts, zs = rn(steps, bsz, dim), 0.5 * rn(steps, bsz, dim), rn(steps, bsz, dim)
    Bs, Cs = rn(steps, bsz, N), rn(steps, bsz, N)
    out = dict(A=npf(A), D=npf(D), dt_bias=npf(bias), w=npf(w), conv_bias=npf(cb), xs=npf(xs),
               dts=npf(dts), zs=npf(zs), Bs=npf(Bs), Cs=npf(Cs))
    for tag, dt_ in (("32", torch.float32), ("64", torch.float64)):
        c = lambda t: t.to(dt_)
        conv_state = torch.zeros(bsz, dim, W, dtype=dt_)
        ssm_state = torch.zeros(bsz, dim, N, dtype=dt_)
        ys, cs = [], []
        for s in range(steps):
            xc = M.causal_conv1d_update_ref(c(xs[s]), conv_state, c(w), c(cb), "silu")
            y = M.selective_state_update_ref(ssm_state, xc, c(dts[s]), c(A), c(Bs[s]), c(Cs[s]),
                                             c(D), z=c(zs[s]), dt_bias=c(bias), dt_softplus=True)
            ys.append(y.clone())
            cs.append(xc.clone())
        out["y" + tag] = npf(torch.stack(ys))
        out["xconv" + tag] = npf(torch.stack(cs))
        out["ssm_state" + tag] = npf(ssm_state)
        out["conv_state" + tag] = npf(conv_state)
    np.savez_compressed(os.path.join(OUT, "step.npz"), **out)


def save_ckpt(name, ck):
    sd = ck["model_state_dict"]
    arrs = {k: v.cpu().numpy() for k, v in sd.items()}            # fp16 as shipped
    arrs["__network_config__"] = np.frombuffer(json.dumps(ck["network_config"]).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, f"ckpt_{name}.npz"), **arrs)


def e2e_from_ckpt(ref, name, path, L, pruned):
    ck = torch.load(path, map_location="cpu", weights_only=False)
    save_ckpt(name, ck)
    net = ref.CleanUMamba(**ck["network_config"])
    if pruned:
        net.load_pruned_state_dict(ck["model_state_dict"])
    else:
        net.load_state_dict(ck["model_state_dict"], strict=True)
    net = net.float().eval()
    torch.manual_seed(1234)
    x = 0.1 * torch.randn(2, 1, L)
    x[1] *= 0.3
    out = {"input": npf(x)}
    sd32 = {k: v.float() for k, v in net.state_dict().items()}
    sd64 = {k: v.double() for k, v in net.state_dict().items()}
    with torch.no_grad():
        for norm in (True, False):
            net.normalize_input = norm
            y, skips = net(x.clone(), return_skip_connections=True)
            tag = "norm" if norm else "raw"
            out["out_" + tag] = npf(y)
            out["out64_" + tag] = npf(R.forward_ref(sd64, x.double(), normalize_input=norm))
            if not norm:
                out["tsfm_in_raw"] = npf(skips[0])        # deepest encoder output (B, C, T)
                out["tsfm_out_raw"] = npf(skips[-1])      # norm_f output (B, d_model, T)
                out["skip_first_raw_head"] = npf(skips[-2][:, :, :64])   # enc0 output head
        assert torch.equal(R.forward_ref(sd32, x, normalize_input=False), torch.from_numpy(out["out_raw"]))
    out["frame_length"] = np.int64(net.frame_length)
    out["total_stride"] = np.int64(net.total_stride)
    out["valid_length"] = np.int64(net.valid_length(L))
    np.savez_compressed(os.path.join(OUT, f"e2e_{name}.npz"), **out)
    print(name, "done", net.frame_length, net.total_stride)


def e2e_synth(ref, name, cfg, L, seed):
    torch.manual_seed(0)
    net = ref.CleanUMamba(**cfg).float().eval()
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = synth.fill_state_dict(shapes, seed=seed)
    net.load_state_dict(sd, strict=True)
    clean, noisy = synth.waveform(2, L, seed=77)
    with torch.no_grad():
        y = net(noisy.clone())
        y64 = R.forward_ref({k: v.double() for k, v in sd.items()}, noisy.double())
    # backward fixture: d(loss)/d(input-independent params) sampled, via the reference class
    net.train()
    yy = net(noisy.clone())
    loss = (yy * clean).sum()
    loss.backward()
    gsel = {}
    named = dict(net.named_parameters())
    for k in ("encoder.0.0.weight", "encoder.1.2.bias", "decoder.0.0.weight", f"decoder.{cfg['encoder_n_layers']-1}.2.weight",
              "tsfm_conv1.weight", "tsfm_Mamba_layers.0.mixer.A_log", "tsfm_Mamba_layers.1.mixer.dt_proj.bias",
              "tsfm_Mamba_layers.2.mixer.D", "tsfm_Mamba_layers.0.mixer.conv1d.weight",
              "tsfm_Mamba_layers.1.mixer.x_proj.weight", "tsfm_Mamba_layers.2.norm.weight", "norm_f.bias"):
        gk = named[k].grad
        gsel["grad:" + k] = npf(gk.flatten()[:4096])          # head slice keeps the fixture small
        gsel["gradnorm:" + k] = np.float64(gk.double().norm().item())
    gsel["grad_sq_total"] = np.float64(sum((p.grad.double() ** 2).sum().item() for p in net.parameters()))
    meta = dict(cfg=cfg, seed=seed, L=L, wave_seed=77, keys=sorted(shapes), shapes=[list(shapes[k]) for k in sorted(shapes)])
    np.savez_compressed(os.path.join(OUT, f"e2e_{name}.npz"), out=npf(y), out64=npf(y64), **gsel,
                        meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
    print(name, "done", float(y.abs().mean()))


def make_loss(ref):
    import importlib
    util = importlib.import_module("src.util.util")
    stft = importlib.import_module("src.util.stft_loss")
    cfg = json.load(open(os.path.join(REF, "configs/config.json")))["train_config"]["loss_config"]
    g = torch.Generator().manual_seed(4000)
    clean = 0.05 * torch.randn(2, 1, 16000, generator=g)
    den = (clean + 0.02 * torch.randn(2, 1, 16000, generator=g)).requires_grad_(True)
    mr = stft.MultiResolutionSTFTLoss(**cfg["stft_config"])
    kw = {k: v for k, v in cfg.items() if k != "stft_config"}
    loss, dic = util.loss_fn(lambda x: den, (clean, clean.clone()), mrstftloss=mr, **kw)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), clean=npf(clean), denoised=npf(den), loss=npf(loss),
                        grad=npf(den.grad), reconstruct=npf(dic["reconstruct"]), stft_sc=npf(dic["stft_sc"]),
                        stft_mag=npf(dic["stft_mag"]),
                        cfg=np.frombuffer(json.dumps(cfg).encode(), dtype=np.uint8))


PRUNED = {  # fixture name -> shipped file (checkpoints/pruned/)
    "e8_pruned200k": "CleanUMamba-3N-E8_pruned-200K.pkl", "e8_pruned1m": "CleanUMamba-3N-E8_pruned-1M.pkl",
    "e8_pruned2m": "CleanUMamba-3N-E8_pruned-2M.pkl", "e6_pruned200k": "CleanUMamba-3N-E6_pruned-200k.pkl",
    "e6_pruned500k": "CleanUMamba-3N-E6_pruned-500k.pkl", "e6_pruned1m": "CleanUMamba-3N-E6_pruned-1M.pkl",
    "e6_pruned2m": "CleanUMamba-3N-E6_pruned-2M.pkl",
}


def make_lr(ref):
    import importlib
    util = importlib.import_module("src.util.util")
    out = {}
    for tag, (n_iter, warm, it0) in {"fresh": (1000, 0.05, 0), "resume30": (1000, 0.05, 30), "resume50": (1000, 0.05, 50),
                                     "resume700": (1000, 0.05, 700), "warm30pct": (400, 0.3, 0),
                                     "nowarm": (300, 0.0, 0)}.items():
        opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
        sch = util.LinearWarmupCosineDecay(opt, lr_max=1e-4, n_iter=n_iter, iteration=it0, divider=25,
                                           warmup_proportion=warm, phase=("linear", "cosine"))
        # (with no warm-up the reference divides by zero on the first step after the wrap: stop at the wrap)
        out[tag] = np.array([sch.step() for _ in range(n_iter - it0 + (100 if warm > 0 else 0))], dtype=np.float64)
        out[tag + "_args"] = np.array([n_iter, warm, it0], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "lr_schedule.npz"), **out)


def main():
    import sys
    os.makedirs(OUT, exist_ok=True)
    ref = reference_shim.load_reference()
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("pruned", "all"):
        for name, fn in PRUNED.items():
            e2e_from_ckpt(ref, name, os.path.join(REF, "checkpoints/pruned", fn), 16000, True)
    if what in ("lr", "all"):
        make_lr(ref)
    if what != "all":
        return
    for i, (b, d, n, l) in enumerate([(2, 8, 8, 33), (2, 48, 13, 257), (1, 128, 16, 61), (2, 64, 64, 96)]):
        make_scan(i, b, d, n, l, seed=i)
    make_scan(4, 1, 16, 8, 40, with_z=False, with_bias=False, with_D=False, seed=4)
    make_scan(5, 2, 24, 14, 1, seed=5)
    for i, (b, d, l) in enumerate([(2, 8, 33), (2, 48, 257), (1, 136, 3), (2, 64, 624)]):
        make_dwconv(i, b, d, l, seed=i)
    make_step()
    e2e_from_ckpt(ref, "442k", os.path.join(REF, "checkpoints/experiments/Experiment_CleanU_Mamba.pkl"), 16000, False)
    e2e_from_ckpt(ref, "pruned500k", os.path.join(REF, "checkpoints/pruned/CleanUMamba-3N-E8_pruned-500K.pkl"), 16000, True)
    e8 = json.load(open(os.path.join(REF, "configs/exp/models/DNS-CleanUMamba-3N-E8.json")))["network_config"]
    e6 = json.load(open(os.path.join(REF, "configs/exp/models/DNS-CleanUMamba-3N-E6.json")))["network_config"]
    e2e_synth(ref, "e8_synth", e8, 6000, seed=8)
    e2e_synth(ref, "e6_synth", e6, 4000, seed=6)
    make_loss(ref)


if __name__ == "__main__":
    main()
