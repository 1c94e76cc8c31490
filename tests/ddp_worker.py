"""Child process of tests/test_train_gpu.py::test_two_ranks_real_model: one data-parallel rank.

Started as a FRESH process (never a re-exec of one that touched the GPU).  Both ranks share cuda:0 and exchange
gradients over gloo (CUM_DIST_BACKEND=gloo of bench.py: the code path of the RCCL run, minus the transport).
Runs the reference's sequence -- init_distributed, apply_gradient_allreduce, TrainStep (src/training/train.py:80-81,
123-124, 255-312) -- on a real CleanUMamba and writes what the parent compares:
  grads after the first backward (before the optimizer touches anything), parameters after `steps` steps.

usage: ddp_worker.py RANK WORLD PORT OUTDIR MODEL(442k|narrow_e8|e8) DTYPE(f32|f16|bf16) STFT(0|1) STEPS [graph]

  e8     the real 41.4 M-parameter E8 of bench.py, one 1 s clip per rank, the DEFAULT 32 MiB buckets
  graph  TrainStep(use_graph=True): with two ranks that is [captured fwd+loss+bwd] -> all-reduce -> [captured optimizer]
CUM_EXCHANGE_ALONE=1 with WORLD 1: a one-rank RCCL ("nccl") group whose collectives all run (see GradBuckets.exchanging).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

NARROW_E8 = dict(channels_input=1, channels_output=1, channels_H=16, max_H=48, encoder_n_layers=8, kernel_size=4,
                 stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=64, tsfm_d_inner=256)   # E8 shape, d_state 8
E8 = dict(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8, kernel_size=4,
          stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)        # bench.py's model


def build(model, dev):
    from cleanumamba_amd.network import CleanUMamba
    if model == "442k":
        with np.load(os.path.join(ROOT, "tests", "golden", "ckpt_442k.npz")) as f:
            cfg = json.loads(bytes(f["__network_config__"]).decode())
            sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}
        net = CleanUMamba(**cfg)
        net.load_state_dict(sd, strict=True)
    else:
        net = CleanUMamba(**(E8 if model == "e8" else NARROW_E8))
    return net.to(dev).train()


def batch(rank, n, length, dev):
    from oracle import synth            # test infrastructure: seeded waveforms only
    clean, noisy = synth.waveform(n, length, seed=500 + rank)
    return clean.to(dev), noisy.to(dev)


def main():
    rank, world, port, out, model, dtype, stft, steps = sys.argv[1:9]
    rank, world, steps = int(rank), int(world), int(steps)
    use_graph = len(sys.argv) > 9 and sys.argv[9] == "graph"
    length, per_rank = (16000, 1) if model == "e8" else (8000, 2)
    from cleanumamba_amd.training.train_distributed import apply_gradient_allreduce, init_distributed
    from cleanumamba_amd.training.train_step import TrainStep
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    alone = os.environ.get("CUM_EXCHANGE_ALONE") == "1"      # one rank of a real RCCL group: every collective runs, over one rank
    if world > 1 or alone:
        init_distributed(rank, world, None, "nccl" if alone else "gloo", f"tcp://127.0.0.1:{port}")
    torch.manual_seed(1000 + rank)                  # ranks initialise differently: the broadcast must fix it
    net = build(model, dev)
    n_buckets = 0
    if world > 1 or alone:
        if model == "e8":
            net = apply_gradient_allreduce(net)                           # the bench's configuration: 32 MiB buckets
        else:
            net = apply_gradient_allreduce(net, bucket_bytes=256 << 10)   # several buckets even on small models
        n_buckets = len(net.grad_buckets.buckets)
        assert n_buckets >= 2
    ac = {"f32": None, "f16": torch.float16, "bf16": torch.bfloat16}[dtype]
    step = TrainStep(net, optimization={"n_iters": 1000}, loss_config={"stft_lambda": int(stft)},
                     autocast_dtype=ac, use_graph=use_graph)
    if os.environ.get("CUM_TEST_BREAK_CAPTURE") == str(rank):
        # this rank's capture fails (the others' succeed): every rank must end up on the eager step
        real_loss = step._loss

        def broken(*a, **k):
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("capture broken on purpose (test)")
            return real_loss(*a, **k)
        step._loss = broken
    if world > 1:
        clean, noisy = batch(rank, per_rank, length, dev)
    else:                                            # the single-process run sees the concatenated batch
        parts = [batch(r, per_rank, length, dev) for r in range(int(os.environ.get("CUM_TEST_RANKS", "2")))]
        clean, noisy = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts])
    # first backward by hand (so the gradients can be dumped before the optimizer runs) ...
    step.zero_grad()
    announced = []                                   # sizes of the gradient groups the kernels hand to the exchange
    if world > 1:
        flat, inner = net.grad_buckets.flat, net.grad_buckets.flat.on_write
        flat.on_write = lambda ps: (announced.append(len(ps)), inner(ps))[1]
    loss0 = step.micro_step(clean, noisy)
    if world > 1:
        flat.on_write = inner
    scale = float(step.optimizer.loss_scale) if ac == torch.float16 else 1.0
    grads = {k: (p.grad.detach().float() / scale).cpu() for k, p in net.named_parameters()}
    step.optimizer_step()
    step.scheduler.step()
    # ... then whole steps
    losses = [float(loss0)]
    for _ in range(steps - 1):
        loss, _ = step(clean, noisy)
        losses.append(float(loss))
    torch.cuda.synchronize()
    params = {k: p.detach().float().cpu() for k, p in net.named_parameters()}
    torch.save({"grads": grads, "params": params, "losses": losses,
                "skipped": float(step.optimizer.state_vec[9]), "announced": announced, "buckets": n_buckets,
                "graph_status": step.graph_status, "numel": sum(p.numel() for p in net.parameters())},
               os.path.join(out, f"rank{rank}_of{world}.pt"))
    if world > 1 or alone:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
