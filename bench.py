"""bench.py -- CleanUMamba-E8 train-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W            (N = 1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N > 1)

A step = one full optimisation step of the reference's hot loop (src/training/train.py:255-312):
forward, L1 + multi-resolution STFT loss, backward (gradient all-reduce over RCCL inside it for
N > 1), clip_grad_norm_(10), fused Adam, LR schedule -- on synthetic 10 s @ 16 kHz clips, 16 per GPU
(BASELINE configs[2]/[3]), random-init E8 weights.  Rank 0 prints ONE JSON line.

value = global_batch * 160000 * K / (max-over-ranks time of K steps).
roofline: the selective-scan forward kernel at the E8 bottleneck shape, algorithmic bytes
(B*T*4*(4*d_inner + 2*N)) / mean launch duration measured with HIP events.
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
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=16)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f16"],
                    help="autocast dtype of the GEMM/conv stack (reference trains under fp16 autocast); "
                         "scan / depthwise-conv kernels always compute in f32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-clip", type=int, default=CLIP, help="samples in the CPU-baseline clip")
    return ap.parse_args()


def scan_roofline(dev, iters=20):
    """Mean duration of the scan forward kernel at the E8 train shape, HIP events on the launch stream."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    bsz, dim, N, L = 16, 2048, 64, 624
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    xz = rn(bsz, L, 2 * dim)
    u, z = xz[..., :dim].transpose(1, 2), xz[..., dim:].transpose(1, 2)
    delta = (0.3 * rn(bsz, L, dim)).transpose(1, 2)
    A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float())[None].repeat(dim, 1)).contiguous()
    xd = rn(bsz, L, 32 + 2 * N)
    Bm, Cm = xd[..., 32:32 + N].transpose(1, 2), xd[..., 32 + N:].transpose(1, 2)
    D, bias = rn(dim), 0.3 * rn(dim)
    with torch.no_grad():
        for _ in range(3):
            selective_scan_fn(u, delta, A, Bm, Cm, D, z=z, delta_bias=bias, delta_softplus=True)
        torch.cuda.synchronize()
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(iters):
            selective_scan_fn(u, delta, A, Bm, Cm, D, z=z, delta_bias=bias, delta_softplus=True)
        end.record()
        torch.cuda.synchronize()
    ms = start.elapsed_time(end) / iters
    alg_bytes = bsz * L * 4 * (4 * dim + 2 * N)           # read u, delta, z, B, C; write out (fp32)
    achieved = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "kernel": "scan_fwd_lds_kernel<8> (B=16,D=2048,N=64,L=624,f32)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            # HBM bytes per launch from PMC passes of this kernel at this shape: 2 x FETCH_SIZE + WRITE_SIZE
            # (gfx950 corrections, calibrated on a known kernel): profiles/r01_scan_pmc.md
            "traffic": 368.7e6, "traffic_source": "profiles/r01_scan_pmc.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)",
            "launch_ms": round(ms, 4), "algorithmic_bytes": alg_bytes,
            "state_updates_per_s": round(bsz * L * dim * N / (ms * 1e-3) / 1e12, 3), "state_updates_unit": "T/s"}


def _time(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters


def other_kernels(dev):
    """Live HIP-event timings of the other heavy kernels of the step at an E8 layer shape, with their roofs
    (bf16 MFMA 2.5 PFLOP/s dense; HBM 8 TB/s).  Informational: `roofline` stays the scan forward kernel."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    from cleanumamba_amd.network import convstack as cs
    out = []
    # encoder layer 5 (768 -> 768, T = 2502, B = 16): conv k4 s2 as GEMM, forward and weight gradient
    M, N, K, lda = 16 * 2504, 768, 3072, 1536
    A = torch.randn(M * lda // 8 + K // 8 + 64, 8, device=dev).bfloat16()
    W = (torch.randn(N, K, device=dev) / K ** 0.5).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    bias = torch.zeros(N, device=dev)
    ms = _time(lambda: cs.gemm(A, 0, lda, W, bias, y, 0, N, M, 1 << 30, 1 << 30, hip.EPI_RELU, N))
    tf = 2.0 * M * N * K / ms / 1e9
    out.append({"kernel": "gemm_nt_kernel<bf16,relu> enc5 conv (M=40064,N=768,K=3072)", "bound": "mfma",
                "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                "launch_ms": round(ms, 4)})
    dz = torch.randn(M, N, device=dev).bfloat16()
    ms = _time(lambda: cs.wgrad(dz, 0, N, N, A, 0, lda, K, M))
    tf = 2.0 * M * N * K / ms / 1e9
    out.append({"kernel": "gemm_tn_kernel<bf16> + reduce, enc5 conv weight gradient", "bound": "mfma",
                "achieved": round(tf, 1), "peak": 2500.0, "unit": "TFLOP/s", "frac": round(tf / 2500.0, 4),
                "launch_ms": round(ms, 4)})
    # selective scan backward at the E8 bottleneck shape
    bsz, dim, Ns, L = 16, 2048, 64, 624
    g = torch.Generator(device=dev).manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)
    xz = rn(bsz, L, 2 * dim).requires_grad_(True)
    dl = (0.3 * rn(bsz, L, dim)).requires_grad_(True)
    Am = (-torch.exp(torch.log(torch.arange(1, Ns + 1, device=dev).float())[None].repeat(dim, 1))).requires_grad_(True)
    xd = rn(bsz, L, 32 + 2 * Ns).requires_grad_(True)
    Dv, bv = rn(dim).requires_grad_(True), (0.3 * rn(dim)).requires_grad_(True)
    dout = rn(bsz, dim, L)

    def fwd():
        return selective_scan_fn(xz[..., :dim].transpose(1, 2), dl.transpose(1, 2), Am, xd[..., 32:32 + Ns].transpose(1, 2),
                                 xd[..., 32 + Ns:].transpose(1, 2), Dv, z=xz[..., dim:].transpose(1, 2), delta_bias=bv,
                                 delta_softplus=True)
    t_f = _time(fwd)
    t_fb = _time(lambda: fwd().backward(dout))
    ms = t_fb - t_f
    byt = bsz * L * 4 * (7 * dim + 4 * Ns)
    gbs = byt / (ms * 1e-3) / 1e9
    out.append({"kernel": "scan_bwd_kernel<8,true> + finalize (B=16,D=2048,N=64,L=624,f32)", "bound": "hbm",
                "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "launch_ms": round(ms, 4), "algorithmic_bytes": byt})
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


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)      # ranks > devices only in the single-GPU gloo self-test below
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    from cleanumamba_amd import hip
    from cleanumamba_amd.network import Net
    from cleanumamba_amd.training.train_distributed import apply_gradient_allreduce, init_distributed
    from cleanumamba_amd.training.train_step import TrainStep
    hip.lib()

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI ("nccl" on ROCm).  CUM_DIST_BACKEND=gloo exists only to exercise this code path with
        # several ranks on a one-GPU box; it is never what the scaling numbers are measured with.
        backend = os.environ.get("CUM_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    torch.manual_seed(0)                                 # reference seeds 0 (src/training/train.py:51-53)
    net = Net("CleanUMamba", E8).to(dev).train()
    if world > 1:
        net = apply_gradient_allreduce(net)
    ac = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": None}[args.dtype]
    step = TrainStep(net, autocast_dtype=ac)

    B = args.batch_per_gpu
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    clean = 0.05 * torch.randn(B, 1, CLIP, generator=g, device=dev)
    noisy = clean + 0.05 * torch.randn(B, 1, CLIP, generator=g, device=dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss, _ = step(clean, noisy)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step(clean, noisy)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = float(loss)

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
               "final_loss": round(final_loss, 5)}
        if not args.no_roofline:
            out["roofline"] = scan_roofline(dev)
            if world == 1:
                out["kernels"] = other_kernels(dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.cpu_clip)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
