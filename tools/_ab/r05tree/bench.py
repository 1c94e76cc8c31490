"""bench.py -- CleanUMamba-E8 train-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (any N: for N > 1 the process spawns its own N ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1, ranks from the env)

A step = one full optimisation step of the reference's hot loop (src/training/train.py:255-312):
forward, L1 + multi-resolution STFT loss, backward (gradient all-reduce over RCCL inside it for
N > 1), clip_grad_norm_(10), fused Adam, LR schedule -- on synthetic 10 s @ 16 kHz clips, 16 per GPU
(BASELINE configs[2]/[3]), random-init E8 weights.  Rank 0 prints ONE JSON line.

value = global_batch * 160000 * K / (max-over-ranks time of K steps).
roofline: the kernel VERDICT names (rocprofv3 --stats: gemm_tn9_kernel + gemm_tn_kernel, the conv-stack weight
gradients), timed live with HIP events over the 16 encoder launch shapes of the step; each shape is priced against the
roof that binds it, the headline is the MFMA-bound group (enc3-enc7) against the dense 16-bit MFMA peak.  `kernels` lists the other heavy kernels the same way
(forward GEMMs; selective scan forward / backward with the SURVEY.md 8d byte counts and state updates / s);
`layers` is the per-layer table of SURVEY.md 8(d) rows a5 / a12: the two forward launches of every encoder / decoder
layer summed, against the FUSED layer's algorithmic bytes (HBM roof) and flops (MFMA roof); `scan` lists the
selective scan at d_state 64 / 16 / 8 (where HBM, not v_exp_f32 issue, is the binding roof).
cpu_baseline: the CPU oracle (oracle/cleanumamba_ref.py, kind "port") doing forward + loss + backward
on a bounded sample (2 clips of 10 s), rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

E8 = dict(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8, kernel_size=4,
          stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)
CLIP = 160000
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: 8 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch-per-gpu", type=int, default=16)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "f32"],
                    help="autocast dtype: element type of every activation and GEMM operand.  f16 (default) is the "
                         "reference's own training mode -- torch.autocast('cuda') + GradScaler, configs/config.json:14, "
                         "src/training/train.py:158-160, 278-280; accumulation, the scan recurrence, parameters and "
                         "optimizer state are f32 in every mode")
    ap.add_argument("--no-graph", action="store_true", help="run the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--graph", action="store_true",
                    help="N > 1 only: time ONLY the captured three-graph step (default for N > 1: both forms are timed, "
                         "the eager backward with per-bucket all-reduce overlap and the three-graph form, and the faster "
                         "one is the headline; --no-graph: only the eager form; one GPU always replays one graph unless "
                         "--no-graph)")
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="seconds the launcher waits for its ranks")
    ap.add_argument("--no-pin", action="store_true", help="N > 1: do not pin the ranks to their GPUs' NUMA nodes")
    ap.add_argument("--no-baseline-mode", action="store_true",
                    help="N > 1: skip the third timed loop (the same GPUs without gradient exchange)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-clip", type=int, default=CLIP, help="samples in the CPU-baseline clip")
    return ap.parse_args()


MFMA_PEAK_TFS = 2500.0      # MI355X_MICROARCH.md: bf16 / f16 dense
# HBM bytes per gemm_tn launch (kernel + slab reduce), mean over the 10 MFMA-bound encoder weight-gradient shapes the
# roofline object is quoted on.  NOT measured inside this run (counters need their own rocprofv3 passes): the figure of this
# round's `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over tools/bench_gemm.py tn, corrected as
# MI355X_MICROARCH.md prescribes (profiles/r05_gemm_tn_pmc.md: 338.5 MB fetched + 71.1 MB written against 218.1 MB algorithmic).
TN_TRAFFIC_BYTES = 409.6e6
# selective-scan issue roof, MEASURED: the bare inner-loop instruction mix of the forward kernel (per state pair v_pk_mul,
# 2 x v_exp_f32, v_pk_mul, 2 x v_pk_fma) on registers only, every SIMD of the chip busy (tools/clock_probe.hip)
SCAN_ISSUE_ROOF = 8.6e12     # state updates / s: tools/clock_probe.hip on MI355X (profiles/r02_clock_probe.txt: 8.56-8.89 T/s
                             # at 6-8 waves per SIMD, clock 2.30-2.36 GHz = 3.6-3.7 updates per clock and SIMD)

B16 = 16
ENC_T = [160254, 80126, 40062, 20030, 10014, 5006, 2502, 1250, 624]
ENC_C = [1, 64, 128, 256, 512, 768, 768, 768, 768]


def _time(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()                     # on torch's current stream = the stream every kernel here is launched on
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters


def _time_graph(fn, reps=10, iters=5, warm=2):
    """GPU time of one call of ``fn``: `reps` calls captured into ONE hipGraph, replayed `iters` times between two events.
    For launches of a few tens of microseconds the eager loop of _time measures the host (Python + ctypes + allocator:
    30-50 us per call), not the kernels; inside a replayed graph nothing runs on the host between the launches."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warm):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        g.replay()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / (iters * reps)


def _rup(x, m):
    return (x + m - 1) // m * m


def _name(dt):
    return {torch.bfloat16: "bf16", torch.float16: "f16", torch.float32: "f32"}[dt]


def tn_roofline(dev, dt=torch.bfloat16):
    """The weight-gradient GEMM (rocprofv3: gemm_tn9_kernel + gemm_tn_kernel, 46 launches per step): weight
    gradients dW = dZ^T X of the conv stack.  Timed live with HIP events on the 16 launch shapes the ENCODER
    contributes to one E8 / B=16 step (conv and 1x1 of each layer; the decoder's 16 launches mirror them with the
    same M*N*K).  Every shape is priced against the roof that binds it: algorithmic bytes s*M*(N + ldx) + 4*N*K over
    8 TB/s vs algorithmic flops 2*M*N*K over the dense 16-bit MFMA peak.  enc0-enc2 are HBM-bound by that test
    (intensity N*K/(N+ldx) < 312 flop/B), enc3-enc7 MFMA-bound.  The headline `achieved` / `frac` is the MFMA-bound
    group (10 shapes, 86 % of the kernel's flops): sum of flops / sum of mean launch durations (kernel + its
    deterministic slab reduce); `hbm_bound_shapes` carries the other group against the HBM roof."""
    from cleanumamba_amd.network import convstack as cs
    sz = torch.empty((), dtype=dt).element_size()
    rows = []
    grp = {"mfma": [0.0, 0.0, 0.0], "hbm": [0.0, 0.0, 0.0]}          # flops, bytes, ms
    for i in range(8):
        M, Cin, H = B16 * (ENC_T[i + 1] + 2), _rup(ENC_C[i], 8), ENC_C[i + 1]
        for name, N, K, ldx in ((f"enc{i}.conv.w", H, 4 * Cin, 2 * Cin), (f"enc{i}.1x1.w", 2 * H, H, H)):
            dz = torch.randn(M, N, device=dev).to(dt)
            X = torch.randn(M * ldx // 8 + K // 8 + 64, 8, device=dev).to(dt)
            ms = _time(lambda: cs.wgrad(dz, 0, N, N, X, 0, ldx, K, M))
            fl = 2.0 * M * N * K
            byt = float(sz * M * (N + ldx) + 4 * N * K)
            bound = "hbm" if byt / (HBM_PEAK_GBS * 1e9) > fl / (MFMA_PEAK_TFS * 1e12) else "mfma"
            rows.append({"shape": f"{name} M={M} N={N} K={K}", "launch_ms": round(ms, 4), "bound": bound,
                         "tflops": round(fl / ms / 1e9, 1), "mfma_frac": round(fl / ms / 1e9 / MFMA_PEAK_TFS, 4),
                         "GBps": round(byt / ms / 1e6, 1), "hbm_frac": round(byt / ms / 1e6 / HBM_PEAK_GBS, 4)})
            g = grp[bound]
            g[0] += fl
            g[1] += byt
            g[2] += ms
            del dz, X
    fl, _, ms = grp["mfma"]
    n_mf = sum(r["bound"] == "mfma" for r in rows)
    tf = fl / ms / 1e9
    hb = grp["hbm"]
    return {"bound": "mfma", "kernel": f"gemm_tn9_kernel<{_name(dt)}> + tn_reduce_kernel: the MFMA-bound weight-gradient "
                                       f"launches of one E8 B=16 step ({n_mf} encoder shapes, enc3-enc7; mirrored by the decoder)",
            "achieved": round(tf, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFS, 4),
            "traffic": TN_TRAFFIC_BYTES, "traffic_source": "constant: mean HBM bytes per launch over these 10 shapes from separate "
                                                             "PMC passes (profiles/r05_gemm_tn_pmc.md: 1.88 x algorithmic), not "
                                                             "measured in this run",
            "launch_ms": round(ms / n_mf, 4), "algorithmic_flops": fl / n_mf, "launches": n_mf,
            "hbm_bound_shapes": {"shapes": len(rows) - n_mf, "achieved_GBps": round(hb[1] / hb[2] / 1e6, 1),
                                 "hbm_frac": round(hb[1] / hb[2] / 1e6 / HBM_PEAK_GBS, 4)},
            "all_16_shapes_tflops": round((grp["mfma"][0] + hb[0]) / (grp["mfma"][2] + hb[2]) / 1e9, 1),
            "per_shape": rows}


def _scan_case(dev, bsz, dim, Ns, L, io, backward):
    """One selective-scan shape: forward without checkpoints (inference), and optionally forward + backward."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    R = max(4, dim // 64)
    # u and z are the two halves of the in_proj output (row stride 2 * dim), B and C slices of the x_proj output, as in the
    # model; every operand is its own autograd leaf and the backward is timed as torch.autograd.grad on a retained graph:
    # the op's backward (kernel + deterministic finalize) and nothing of autograd's slice-backward / AccumulateGrad
    # copies (zeros + copy + add of (B, L, 2 dim) tensors: ~1 ms at L = 2499 in f32, as much as the kernel)
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

    def fwd():
        return selective_scan_fn(u, dl, Am, Bm, Cm, Dv, z=z, delta_bias=bv, delta_softplus=True)

    def fwd_nograd():
        with torch.no_grad():
            return fwd()
    from cleanumamba_amd import hip
    from cleanumamba_amd.mamba_ssm.ops import selective_scan_interface as ssi
    # short launches (small grids): graph-replay timing, see _time_graph
    small = bsz * ((dim + 63) // 64) * ((Ns + 7) // 8) <= 1024
    timer = _time_graph if small else _time
    t_i = timer(fwd_nograd)
    _scan_case.sequential_ms = None
    if hip.lib().cum_scan_fwd_workspace_elems(bsz, dim, Ns, L) > 0:
        # this shape takes the time-parallel forward (csrc/scan_seg.hip): the sequential kernels on the same box beside it
        ssi.TIME_PARALLEL = False
        try:
            _scan_case.sequential_ms = timer(fwd_nograd)
        finally:
            ssi.TIME_PARALLEL = True
    t_b = None
    _scan_case.sequential_bwd_ms = None
    if backward:
        out = fwd()                      # checkpoints saved once; the op's backward node alone is replayed
        bwd = lambda: torch.autograd.grad(out, leaves, dout, retain_graph=True)
        if small:
            # graph-replay timing cannot capture the autograd engine (its worker thread is outside the capture): call the
            # op's backward node itself on this thread -- the same kernels, none of the engine
            node = out.grad_fn
            while node is not None and "SelectiveScan" not in type(node).__name__:
                node = node.next_functions[0][0]
            if node is not None:
                bwd = lambda: type(node)._forward_cls.backward(node, dout)      # (a Function's grad_fn IS its ctx)
        t_b = timer(bwd)
        if hip.lib().cum_scan_bwd_tp_workspace_elems(bsz, dim, Ns, L) > 0:
            # this shape takes the time-parallel backward (csrc/scan_bwd_small.hip): the sequential kernels beside it
            ssi.TIME_PARALLEL = False
            try:
                _scan_case.sequential_bwd_ms = timer(bwd)
            finally:
                ssi.TIME_PARALLEL = True
    return t_i, t_b


def scan_rows(dev, dt):
    """The north_star kernel against both of its roofs.  Algorithmic bytes: SURVEY.md 8(d), B*T*(s*4*D + 4*2*N) forward
    (u, delta, z, out in the I/O type of s bytes; B, C in f32), B*T*(s*7*D + 4*4*N) backward.  State updates: B*T*D*N.
    HBM roof 8 TB/s; issue roof = SCAN_ISSUE_ROOF, the measured rate of the bare update loop (one v_exp_f32 per update).
    At d_state 64 the issue roof binds; at d_state <= 16 the HBM roof does."""
    issue_roof = SCAN_ISSUE_ROOF
    cases = [("E8 bottleneck B=16 D=2048 N=64 L=624", 16, 2048, 64, 624, dt, True),
             ("E8 bottleneck, f32 I/O", 16, 2048, 64, 624, torch.float32, False),
             ("E6 bottleneck B=32 D=2048 N=64 L=2499", 32, 2048, 64, 2499, dt, False),
             ("D=2048 N=16 L=2499 B=16", 16, 2048, 16, 2499, dt, True),
             ("D=2048 N=16 L=2499 B=16, f32 I/O", 16, 2048, 16, 2499, torch.float32, False),
             ("D=2048 N=8 L=2499 B=16", 16, 2048, 8, 2499, dt, True),
             ("D=2048 N=8 L=2499 B=16, f32 I/O", 16, 2048, 8, 2499, torch.float32, False),
             ("D=2048 N=16 L=2499 B=128, f32 I/O", 128, 2048, 16, 2499, torch.float32, False),
             ("D=2048 N=8 L=2499 B=128, f32 I/O", 128, 2048, 8, 2499, torch.float32, False),
             ("D=2048 N=8 L=2499 B=128", 128, 2048, 8, 2499, dt, False),
             ("E8 bottleneck B=1 (file denoising) D=2048 N=64 L=624", 1, 2048, 64, 624, dt, False),
             ("442K model B=16 D=128 N=16 L=624", 16, 128, 16, 624, torch.float32, True),
             ("pruned-E8 block B=256 D=48 N=8 L=1875 (30 s)", 256, 48, 8, 1875, torch.float32, True)]
    rows = []
    for name, bsz, dim, Ns, L, io, bwd in cases:
        t_i, t_b = _scan_case(dev, bsz, dim, Ns, L, io, bwd)
        seq_ms = _scan_case.sequential_ms
        sz = torch.empty((), dtype=io).element_size()
        upd = bsz * L * dim * Ns
        for kind, ms, byt in (("fwd", t_i, bsz * L * (sz * 4 * dim + 4 * 2 * Ns)),
                              ("bwd", t_b, bsz * L * (sz * 7 * dim + 4 * 4 * Ns))):
            if ms is None:
                continue
            gbs, ups = byt / (ms * 1e-3) / 1e9, upd / (ms * 1e-3)
            rows.append({"kernel": f"selective scan {kind}, {name}, {_name(io)} I/O", "launch_ms": round(ms, 4),
                         "algorithmic_bytes": byt, "achieved_GBps": round(gbs, 1),
                         "hbm_frac": round(gbs / HBM_PEAK_GBS, 4), "state_updates_T_per_s": round(ups / 1e12, 3),
                         "issue_roof_frac": round(ups / issue_roof, 4) if kind == "fwd" else None,
                         "binding_roof": "hbm" if byt / (HBM_PEAK_GBS * 1e9) > upd / issue_roof else "v_exp_f32 issue"})
            if kind == "bwd" and _scan_case.sequential_bwd_ms is not None:
                sb = _scan_case.sequential_bwd_ms
                rows[-1].update(path="time-parallel (segments + carry, csrc/scan_bwd_small.hip PASS 1 / 0)",
                                sequential_kernel_ms=round(sb, 4), speedup_vs_sequential=round(sb / ms, 2))
            if kind == "fwd" and seq_ms is not None:
                rows[-1].update(path="time-parallel (segments + carry, csrc/scan_seg.hip)",
                                sequential_kernel_ms=round(seq_ms, 4), speedup_vs_sequential=round(seq_ms / ms, 2))
    return rows


def other_kernels(dev, dt=torch.bfloat16):
    """Live HIP-event timings of the other heavy kernels of the step, each against the roof that bounds it
    (16-bit MFMA 2.5 PFLOP/s dense; HBM 8 TB/s).  Informational; `roofline` is the weight-gradient GEMM."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.network import convstack as cs
    out = []
    # forward GEMMs of the encoder (conv k4 s2 + ReLU, 1x1 + GLU), all 16 launch shapes of one step
    fl_sum = ms_sum = 0.0
    for i in range(8):
        M, Cin, H = B16 * (ENC_T[i + 1] + 2), _rup(ENC_C[i], 8), ENC_C[i + 1]
        for N, K, lda, epi in ((H, 4 * Cin, 2 * Cin, hip.EPI_RELU), (2 * H, H, H, hip.EPI_GLU)):
            A = torch.randn(M * lda // 8 + K // 8 + 64, 8, device=dev).to(dt)
            W = (torch.randn(_rup(N, 32), _rup(K, 64), device=dev) / K ** 0.5).to(dt)
            nout = N // 2 if epi == hip.EPI_GLU else N
            y = torch.empty(M, nout, device=dev, dtype=dt)
            bias = torch.zeros(W.shape[0], device=dev)
            ms_sum += _time(lambda: cs.gemm(A, 0, lda, W, bias, y, 0, nout, M, 1 << 30, 1 << 30, epi, nout))
            fl_sum += 2.0 * M * N * K
            del A, W, y
    tf = fl_sum / ms_sum / 1e9
    out.append({"kernel": f"gemm_nt9_kernel / gemm_nt_kernel<{_name(dt)}> (ReLU / GLU epilogues), the 16 encoder forward launches of one step",
                "bound": "mfma", "achieved": round(tf, 1), "peak": MFMA_PEAK_TFS, "unit": "TFLOP/s",
                "frac": round(tf / MFMA_PEAK_TFS, 4), "launch_ms": round(ms_sum / 16, 4)})
    return out


def layer_table(net, dev, dt):
    """SURVEY.md 8(d) rows a5 / a12 on the E8 B=16 training shapes: per encoder / decoder layer the forward launches the
    model runs -- ONE for the first two encoder layers / the last two decoder layers (csrc/enc0.hip, ench.hip, dech.hip, dec7.hip), else TWO (conv + ReLU,
    1x1 + GLU | 1x1 + GLU, transposed conv + ReLU + skip) -- summed, against the FUSED layer's algorithmic
    traffic s*B*(Cin*Tin + H*Tout) (+ weights; decoder: s*B*(2*H*T + Cout*(2T+2))) and flops 2*B*Tout*(4*Cin*H + 2*H^2).
    The H-channel intermediate and the saved gate pre-activation are real traffic of the two-launch form that the
    algorithmic figure does not count: hbm_frac is what a fused layer kernel could recover on the outer layers."""
    from cleanumamba_amd.network import convstack as cs
    B, s = B16, torch.empty((), dtype=dt).element_size()
    with torch.no_grad():
        net._activate_pack_plan(dt)                      # record + pack once; take() then returns cached views
    rows = []
    geo = cs.Geo(B, ENC_T[0], 1)
    enc_geos = []
    for i, enc in enumerate(net.encoder):
        T1 = ENC_T[i + 1]
        gm, go = cs.Geo(B, T1, enc[0].weight.shape[0]), cs.Geo(B, T1, enc[2].weight.shape[0] // 2)
        enc_geos.append((geo, gm, go))
        geo = go
    with torch.no_grad():
        for i, ((gi, gm, go), enc) in enumerate(zip(enc_geos, net.encoder)):
            x = (0.5 * torch.randn(gi.R, gi.Cp, device=dev)).to(dt)

            fused = i == 0 and cs._enc0_ok(enc[0].weight, enc[2].weight, gi, gm, go, dt)      # what the model runs
            fused_h = not fused and cs._ench_ok(enc[0].weight, enc[2].weight, gi, gm, go, dt)  # csrc/ench.hip (width 128)

            def run():
                if fused:
                    return cs._enc0_fwd(x, enc[0].weight, enc[0].bias, enc[2].weight, enc[2].bias, gm, go, True)
                if fused_h:         # training form: hidden activation, sign nibbles and gate stored for the backward
                    return cs._ench_fwd(x, enc[0].weight, enc[0].bias, enc[2].weight, enc[2].bias, gi, gm, go, True)
                y1 = cs._conv_relu_fwd(x, enc[0].weight, enc[0].bias, gi, gm)
                return cs._glu_fwd(y1, enc[2].weight, enc[2].bias, gm, go, True)
            run()
            net._activate_pack_plan(dt)
            ms = _time(run, iters=5, warm=2)
            Cin, H = gi.C, gm.C
            byt = s * B * (Cin * gi.T + H * go.T) + s * (4 * Cin * H + 2 * H * H)
            fl = 2.0 * B * go.T * (4 * Cin * H + 2 * H * H)
            rows.append(_layer_row(f"enc{i} {Cin}->{H} T {gi.T}->{go.T}", ms, byt, fl, 1 if (fused or fused_h) else 2))
            del x
        E = len(net.decoder)
        gi = enc_geos[-1][2]
        for j, dec in enumerate(net.decoder):
            gg = cs.Geo(B, gi.T, dec[0].weight.shape[0] // 2)
            go = cs.Geo(B, 2 * gi.T + 2, dec[2].weight.shape[1])
            u = (0.5 * torch.randn(gi.R, gi.Cp, device=dev)).to(dt)
            skip = (0.5 * torch.randn(go.R, go.Cp, device=dev)).to(dt) if j < E - 1 else None

            fused = j == E - 1 and cs._dec7_ok(dec[0].weight, dec[2].weight, gi, gg, go, dt)
            fused_h = not fused and cs._dech_ok(dec[0].weight, dec[2].weight, skip, j < E - 1, gi, gg, go, dt)   # csrc/dech.hip

            def run():
                if fused:
                    return cs._dec7_fwd(u, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, gi, go)
                if fused_h:         # training form: GLU output, gate and sign nibbles stored for the backward
                    return cs._dech_fwd(u, dec[0].weight, dec[0].bias, dec[2].weight, dec[2].bias, skip, gi, gg, go, True)
                g, _ = cs._glu_fwd(u, dec[0].weight, dec[0].bias, gi, gg, True)
                return cs._convt_fwd(g, dec[2].weight, dec[2].bias, skip, gg, go, j < E - 1)
            run()
            net._activate_pack_plan(dt)
            ms = _time(run, iters=5, warm=2)
            H, Cout = gg.C, go.C
            byt = s * B * (H * gi.T + Cout * go.T * (2 if skip is not None else 1)) + s * (2 * H * H + 4 * H * Cout)
            fl = 2.0 * B * gi.T * (2 * H * H + 4 * H * Cout)
            rows.append(_layer_row(f"dec{j} {H}->{Cout} T {gi.T}->{go.T}", ms, byt, fl, 1 if (fused or fused_h) else 2))
            gi = go
            del u, skip
    return rows


def _layer_row(name, ms, byt, fl, launches=2):
    gbs, tf = byt / (ms * 1e-3) / 1e9, fl / (ms * 1e-3) / 1e12
    hb, mf = gbs / HBM_PEAK_GBS, tf / MFMA_PEAK_TFS
    return {"layer": name, "launches": launches, "ms": round(ms, 4), "algorithmic_bytes": int(byt), "algorithmic_flops": int(fl),
            "achieved_GBps": round(gbs, 1), "hbm_frac": round(hb, 4), "achieved_TFLOPs": round(tf, 1),
            "mfma_frac": round(mf, 4), "bound": "hbm" if byt / (HBM_PEAK_GBS * 1e9) > fl / (MFMA_PEAK_TFS * 1e12) else "mfma"}


E6 = dict(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=6, kernel_size=4,
          stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)


def c2_e6_forward(dev, dt):
    """BASELINE.json configs[1]: CleanUMamba-E6 (27.2 M) forward on one GPU, batch 32, 10 s @ 16 kHz (no_grad, random init,
    reference init seed 0), under autocast `dt` and in f32; plus its selective scan in isolation (the `scan` row of the
    same shape: B = 32, D = 2048, N = 64, L = 2499).  The output-vs-reference tolerance check of this config is
    tests/test_model_gpu.py / test_train_gpu.py on the golden E6 fixture."""
    from cleanumamba_amd.network import Net
    torch.manual_seed(0)
    net = Net("CleanUMamba", E6).to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(4321)
    noisy = 0.05 * torch.randn(32, 1, CLIP, generator=g, device=dev)
    out = {"workload": "CleanUMamba-E6 (27.2M) forward, batch 32, 10 s @ 16 kHz, no_grad", "batch": 32, "clip_samples": CLIP}
    with torch.no_grad():
        with torch.autocast("cuda", dtype=dt):
            ms = _time(lambda: net(noisy), iters=10, warm=3)
        out[f"forward_{_name(dt)}_ms"] = round(ms, 3)
        out[f"samples_per_s_{_name(dt)}"] = round(32 * CLIP / ms * 1e3, 1)
        ms32 = _time(lambda: net(noisy), iters=5, warm=3)     # (first f32 calls build the f32 pack plan: two warm-ups are setup)
        out["forward_f32_ms"] = round(ms32, 3)
        out["samples_per_s_f32"] = round(32 * CLIP / ms32 * 1e3, 1)
    del net, noisy
    torch.cuda.empty_cache()
    return out


def c5_streaming(dev, streams=256, seconds=30.0):
    """BASELINE.json configs[4]: the pruned CleanUMamba-E8 (492 K parameters, shipped checkpoint: tests/golden/
    ckpt_pruned500k.npz is its state dict as data) streaming 256 concurrent 30 s @ 16 kHz streams through feed_batch /
    flush_batch (16 hops = 256 ms of audio per call).  Real-time factor = audio seconds produced / wall seconds, aggregate
    over the streams (the reference prints ms/frame and x real time for ONE stream, src/examples/streaming_demo.py:183-186).
    Rows: "f32" = the default path, every hop of a call in ONE launch (csrc/hop.hip); "f32_per_layer" = the per-layer
    hop it replaced (fused GEMM launches, hipGraph); "bf16_conv_activations" = that path with 16-bit activations.
    `roofline`: the model's multiply-adds of a hop x 2 x streams / time against the f32 matrix peak (the kernel computes
    in exact f32 on v_mfma_f32_16x16x4_f32).  The warm-up includes one flush: the first drain of a process loads the
    BLAS library behind its einsums (0.26 s once, not per stream)."""
    import numpy as np
    from cleanumamba_amd.network import CleanUMamba
    with np.load(os.path.join(ROOT, "tests", "golden", "ckpt_pruned500k.npz")) as f:
        cfg = json.loads(bytes(f["__network_config__"]).decode())
        sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}
    out = {"workload": f"pruned CleanUMamba-E8 (492K) streaming, {streams} concurrent streams x {seconds:.0f} s @ 16 kHz",
           "streams": streams, "seconds_per_stream": seconds}
    n = int(seconds * 16000)
    g = torch.Generator(device=dev).manual_seed(99)
    x = 0.05 * torch.randn(streams, n, generator=g, device=dev)
    for tag, kernel, bf16 in (("f32", True, False), ("f32_per_layer", False, False),
                              ("bf16_conv_activations", False, True)):
        net = CleanUMamba(**cfg)
        net.load_pruned_state_dict(sd)
        net = net.to(dev).eval()
        net.use_hop_kernel, net.stream_bf16 = kernel, bf16
        hop = net.total_stride
        with torch.no_grad():
            net.feed_batch(x[:, :4 * hop + net.frame_length])       # warm-up: state buffers, hop plan / graph, drain
            net.flush_batch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(0, n, 16 * hop):                         # 256 ms of audio per call
                net.feed_batch(x[:, i:i + 16 * hop])
            status = net.hop_kernel_status if kernel else net.hop_graph_status
            plan = net.__dict__.get("_hop_plan")
            net.flush_batch()                                       # (flush ends the streams; their state goes with them)
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
        row = {"wall_s": round(wall, 3), "ms_per_hop": round(1e3 * wall / (n // hop), 4),
               "hop_ms_audio": 1e3 * hop / 16000, "rtf_aggregate": round(streams * seconds / wall, 1),
               "rtf_per_stream": round(seconds / wall, 2), "hop_path": ("one launch: " if kernel else "graph: ") + status}
        if kernel and plan is not None:
            flops = plan[1].flops_per_hop * streams * (n // hop)
            row["launches_per_hop"] = round(1.0 / 16, 4)
            row["roofline"] = {"bound": "mfma", "achieved": round(flops / wall / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s",
                               "frac": round(flops / wall / 157.3e12, 4), "flops_per_hop_and_stream": plan[1].flops_per_hop}
        out[tag] = row
        del net
    return out


def cpu_baseline(clip):
    """Oracle forward + loss + backward on the host cores, one clip (bounded sample)."""
    from oracle import cleanumamba_ref as R
    from oracle import synth
    from cleanumamba_amd.network import CleanUMamba
    threads = min(os.cpu_count() or 1, 32)     # the 624-step scan loop of small ops does not scale past ~32 threads
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    net = CleanUMamba(**E8)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    clean, noisy = synth.waveform(2, clip, seed=1234)
    t0 = time.time()
    y = R.forward_ref(sd, noisy)
    loss = R.loss_ref(y, clean, stft_config={"sc_lambda": 0.5, "mag_lambda": 0.5, "band": "full",
                                             "hop_sizes": [50, 120, 240], "win_lengths": [240, 600, 1200],
                                             "fft_sizes": [512, 1024, 2048]})
    loss.backward()
    dt = time.time() - t0
    return {"value": round(2 * clip / dt, 1), "unit": "audio samples/s", "cores": threads, "kind": "port",
            "sample": f"oracle/cleanumamba_ref.py forward+loss+backward, E8, batch 2, {clip} samples per clip, "
                      f"{dt:.1f} s wall (no optimizer step)"}


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _cpus_of_rank(local_rank, local_world, sys_root="/sys", allowed=None):
    """CPU set for one rank of a node: the cores of the NUMA node its GPU hangs off (AMD display-class PCI devices in bus
    order = HIP's device order), shared evenly with the other ranks of that node; an even contiguous split of the
    allowed cores when sysfs does not tell.  Pure host logic (no GPU call): runs in the rank before anything else."""
    allowed = sorted(allowed if allowed is not None else os.sched_getaffinity(0))

    def cpulist(text):
        out = []
        for part in text.strip().split(","):
            if part:
                a, _, b = part.partition("-")
                out += range(int(a), int(b or a) + 1)
        return out
    try:
        gpus = []
        pci = os.path.join(sys_root, "bus", "pci", "devices")
        for bdf in sorted(os.listdir(pci)):
            d = os.path.join(pci, bdf)
            with open(os.path.join(d, "vendor")) as f:
                vendor = f.read().strip()
            with open(os.path.join(d, "class")) as f:
                cls = f.read().strip()
            if vendor == "0x1002" and cls.startswith(("0x0302", "0x0380", "0x0300", "0x1200")):
                with open(os.path.join(d, "numa_node")) as f:
                    gpus.append(int(f.read().strip()))
        if len(gpus) >= local_world and gpus[local_rank] >= 0:
            node = gpus[local_rank]
            with open(os.path.join(sys_root, "devices", "system", "node", f"node{node}", "cpulist")) as f:
                cores = [c for c in cpulist(f.read()) if c in set(allowed)]
            peers = [r for r in range(local_world) if gpus[r] == node]
            share = len(cores) // len(peers)
            if share >= 1:
                i = peers.index(local_rank)
                return cores[i * share:(i + 1) * share]
    except (OSError, ValueError, IndexError):
        pass
    share = max(1, len(allowed) // max(local_world, 1))
    return allowed[local_rank * share:(local_rank + 1) * share] or allowed


def launch_ranks(args):
    """`python bench.py --gpus N` with no rank environment: start N FRESH rank processes (the reference starts its own
    ranks the same way, src/training/train_distributed.py:172-178) -- before this process has made any GPU call; it
    never touches the GPU itself and never re-execs -- wait for them with a bound, relay rank 0's JSON line, and exit
    non-zero if any rank failed."""
    import subprocess
    import tempfile
    n = args.gpus
    env = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, len(os.sched_getaffinity(0)) // n)))     # cores this process may use
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    outs, logs, procs = [], [], []
    for r in range(n):
        outs.append(tempfile.TemporaryFile(mode="w+"))
        logs.append(tempfile.TemporaryFile(mode="w+"))
        procs.append(subprocess.Popen(cmd, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)), stdout=outs[r], stderr=logs[r]))
    deadline = time.time() + args.rank_timeout
    failed = None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:                              # one rank died: the others would sit in a collective until its timeout
            failed = f"rank {bad[0][0]} exited with code {bad[0][1]}"
        elif all(c == 0 for c in codes):
            break
        elif time.time() > deadline:
            failed = f"ranks did not finish within {args.rank_timeout:.0f} s"
        else:
            time.sleep(0.2)

    def text(f, tail):
        f.seek(0)
        return f.read()[-tail:]
    if failed is not None:
        for p in procs:                      # exactly the processes started above, by handle
            if p.poll() is None:
                p.kill()
        for r in range(n):
            sys.stderr.write(f"---- rank {r} ----\n{text(outs[r], 1500)}\n{text(logs[r], 3000)}\n")
        sys.stderr.write(f"bench.py: {failed}\n")
        sys.exit(1)
    line = [ln for ln in text(outs[0], 1 << 30).splitlines() if ln.startswith("{")]
    if not line:
        sys.stderr.write(f"bench.py: rank 0 printed no JSON line\n{text(outs[0], 2000)}\n{text(logs[0], 3000)}\n")
        sys.exit(1)
    print(line[-1], flush=True)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        # a rank that hangs (a collective one peer never joined) says where: all thread stacks go to stderr shortly
        # before the launcher's deadline, and the launcher relays every rank's stderr when it gives up
        import faulthandler
        faulthandler.dump_traceback_later(max(30.0, 0.8 * args.rank_timeout), exit=False)
    cpus = None
    if world > 1 and not args.no_pin:
        # every rank on the cores of its GPU's NUMA node, before any GPU call: the eager multi-rank step spends ~16 ms of
        # Python per 20 ms step, a rank that migrates between sockets is the scaling curve's jitter
        try:
            cpus = _cpus_of_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
            os.sched_setaffinity(0, cpus)
            torch.set_num_threads(max(1, min(torch.get_num_threads(), len(cpus))))      # no more threads than cores
        except (OSError, ValueError):
            cpus = None
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)      # ranks > devices only in the single-GPU gloo self-test below
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from cleanumamba_amd import hip
    from cleanumamba_amd.network import Net
    from cleanumamba_amd.training.train_distributed import apply_gradient_allreduce
    from cleanumamba_amd.training.train_step import TrainStep
    hip.lib()

    # CUM_EXCHANGE_ALONE=1 at --gpus 1: a one-rank RCCL group whose collectives all run (GradBuckets.exchanging) -- the
    # data-parallel step's own cost on one GPU: two graphs + one 165.5 MB all-reduce launch instead of one graph
    alone = world == 1 and os.environ.get("CUM_EXCHANGE_ALONE") == "1"
    if world > 1 or alone:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # RCCL over xGMI ("nccl" on ROCm).  CUM_DIST_BACKEND=gloo exists only to exercise this code path with
        # several ranks on a one-GPU box; it is never what the scaling numbers are measured with.
        backend = os.environ.get("CUM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    ac = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": None}[args.dtype]
    B = args.batch_per_gpu
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    clean = 0.05 * torch.randn(B, 1, CLIP, generator=g, device=dev)
    noisy = clean + 0.05 * torch.randn(B, 1, CLIP, generator=g, device=dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def run_mode(exchange, use_graph):
        """A fresh model + train step in one form, settled, warmed up, then EXACTLY args.steps steps between two
        barriers; -> dict(ms, elapsed (max over ranks), host ms by rank, loss, graph status, optimizer info, net)."""
        torch.manual_seed(0)                             # reference seeds 0 (src/training/train.py:51-53)
        net = Net("CleanUMamba", E8).to(dev).train()
        if exchange:
            net = apply_gradient_allreduce(net)
        # use_graph=None: TrainStep's default -- one graph for one process, the eager overlapped exchange for several ranks
        step = TrainStep(net, autocast_dtype=ac, use_graph=use_graph)
        settle = 0
        if ac == torch.float16:
            # Dynamic loss scaling starts at 65536 (GradScaler's default, as in the reference) and backs off while the
            # first scaled gradients overflow; those optimizer steps are skipped.  Let the scale settle before the W
            # warm-up steps so that the timed region holds real steps only (untimed, at most 24 extra steps).
            clean_run, skipped = 0, 0.0
            while clean_run < 2 and settle < 24:
                step(clean, noisy)
                settle += 1
                now = float(step.optimizer.state_vec[9])
                clean_run = clean_run + 1 if now == skipped else 0
                skipped = now
        for _ in range(args.warmup):
            loss, _ = step(clean, noisy)
        skipped_before = float(step.optimizer.state_vec[9]) if step.flat else 0.0
        barrier()
        host0 = step.host_seconds
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss, _ = step(clean, noisy)
        host_ms = 1e3 * (step.host_seconds - host0) / max(args.steps, 1)  # enqueue time only: nothing inside synchronises
        barrier()
        elapsed = time.perf_counter() - t0
        host_ms_ranks = [host_ms]
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([elapsed, host_ms], device=dev, dtype=torch.float64)
            gathered = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(gathered, t)
            elapsed = max(float(x[0]) for x in gathered)                  # MAX over ranks
            host_ms_ranks = [float(x[1]) for x in gathered]
        info = {"optimizer": "flat clip + Adam (csrc/optim.hip)" if step.flat else "torch.optim.Adam"}
        if step.flat:
            sv = step.optimizer.state_vec.cpu()
            info.update(loss_scale=float(sv[3]) if ac == torch.float16 else None, settle_steps=settle,
                        skipped_steps_in_timed_region=float(sv[9]) - skipped_before, adam_steps_total=float(sv[5]))
        return {"ms": 1e3 * elapsed / args.steps, "elapsed": elapsed, "host": host_ms_ranks, "loss": float(loss),
                "graph": step.graph_status, "optim": info, "net": net}

    exchanging = world > 1 or alone
    modes = {}
    if not exchanging:
        best = run_mode(False, False if args.no_graph else None)
    elif args.no_graph or args.graph:
        best = run_mode(True, not args.no_graph)
    else:
        # several ranks: BOTH step forms in one invocation (the first multi-GPU run must not measure half of what
        # matters): the eager step with the per-bucket exchange overlapped, then the three-graph form; the headline is the
        # faster one.  Then the same GPUs without any exchange: what the data-parallel step adds on this node.
        eager = run_mode(True, False)
        eager.pop("net")
        torch.cuda.empty_cache()
        graph = run_mode(True, True)
        modes = {"eager_overlapped": round(eager["ms"], 3), "three_graphs": round(graph["ms"], 3),
                 "three_graphs_status": graph["graph"]}
        best = graph if graph["ms"] < eager["ms"] and graph["graph"] == "captured" else dict(eager, net=graph["net"])
        if best is not graph:
            graph.pop("net", None)
        torch.cuda.empty_cache()
        if not args.no_baseline_mode:
            solo = run_mode(False, None)
            solo.pop("net")
            torch.cuda.empty_cache()
            modes["no_exchange"] = round(solo["ms"], 3)
            modes["exchange_exposed_ms"] = round(min(eager["ms"], graph["ms"]) - solo["ms"], 3)
    elapsed, host_ms_ranks, final_loss, graph_status, optim_info, net = (best["elapsed"], best["host"], best["loss"],
                                                                         best["graph"], best["optim"], best["net"])
    rccl_ranks = None
    if exchanging:
        import torch.distributed as dist
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                              # the rank count the collective itself observes
        rccl_ranks = int(one.item())

    if rank == 0:
        gb = B * world
        out = {"metric": "audio samples/sec/node (train step, E8, 10s@16kHz)",
               "value": round(gb * CLIP * args.steps / elapsed, 1), "unit": "audio samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "CleanUMamba-E8 (41.4M) full train step: fwd + L1 + multi-res STFT loss + bwd"
                                      " + grad all-reduce + clip + Adam; 10 s @ 16 kHz clips",
                          "global_batch": gb, "batch_per_gpu": B, "clip_samples": CLIP,
                          "parallelism": f"dp{world}", "weights": "random init (reference init, seed 0)"},
               "final_loss": round(final_loss, 5), "step_graph": graph_status, "optimizer": optim_info,
               "host_ms_per_step_by_rank": [round(h, 3) for h in host_ms_ranks],
               "modes": modes or None, "collective_ranks_observed": rccl_ranks,
               "backend": (os.environ.get("CUM_DIST_BACKEND", "nccl") if exchanging else None),
               "cpu_affinity_rank0": (f"{len(cpus)} cores: {cpus[0]}-{cpus[-1]}" if cpus else None),
               "exchange": ("none" if world == 1 and not alone else
                            "three graphs: all-reduce (AVG) of the decoder + bottleneck gradients (106 MB) beside the captured "
                            "encoder backward, all-reduce of the encoder's (59 MB) after it, then the captured optimizer section"
                            if graph_status == "captured" else "per-bucket all-reduce overlapped with the eager backward")}
        if not args.no_roofline:
            torch.cuda.empty_cache()
            kdt = ac if ac is not None else torch.bfloat16
            out["roofline"] = tn_roofline(dev, kdt)
            if world == 1:
                out["kernels"] = other_kernels(dev, kdt)
                out["scan"] = scan_rows(dev, kdt)
                # the kernel north_star names, against the HBM roof it nominates and the issue roof that binds at N = 64
                out["north_star_kernel"] = out["scan"][0]
                out["layers"] = layer_table(net, dev, kdt)
                del net
                torch.cuda.empty_cache()
                # the other single-GPU configurations of BASELINE.json, so that the driver's record carries them
                out["c2_e6_forward"] = c2_e6_forward(dev, kdt)
                out["c5_streaming"] = c5_streaming(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_clip)
        line = json.dumps(out)
    # The JSON line is the LAST line this job writes: libraries loaded by the ranks (RCCL prints its "Librccl path" line
    # through C stdio, which a pipe buffers until exit) are flushed first, by every rank, before the closing barrier.
    import ctypes
    ctypes.CDLL(None).fflush(None)
    sys.stdout.flush()
    if world > 1 or alone:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
        ctypes.CDLL(None).fflush(None)
    if rank == 0:
        print(line, flush=True)


if __name__ == "__main__":
    main()
