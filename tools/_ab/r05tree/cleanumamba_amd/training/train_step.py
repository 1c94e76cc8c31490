"""One optimisation step with the semantics of the reference's hot loop
(src/training/train.py:255-312): zero grads, ``repeats`` micro-steps of loss_fn under optional autocast with
(scaled) backward of loss / repeats -- the gradient exchange inside the last one --, unscale,
clip_grad_norm_(10), Adam step (lr 1e-4, betas .9/.999, eps 1e-8), loss-scale update, LR schedule.

Host-side differences:
  * no per-step ``loss.item()`` -- the loss stays on the device and is synchronised only when the caller reads it;
  * on the GPU, parameters / gradients / Adam moments are flat buffers and the optimizer section is three launches
    with device-side loss scaling (training/flat_optim.py, csrc/optim.hip) instead of ~10 multi-tensor launches over
    103 tensors and GradScaler's device->host sync;
  * with static shapes the whole step (forward, loss, backward, optimizer section) is captured ONCE in a hipGraph and
    replayed: ~320 launches per step leave the host's critical path (``use_graph``).  With several ranks the default is
    the eager step with per-bucket overlap (training/train_distributed.py); ``use_graph=True`` opts into the captured
    form, in which no collective is ever captured: THREE graphs -- [zero_grad, forward, loss, backward of decoder +
    bottleneck] -> all-reduce of their gradients (the head of the flat buffer: 106 of 165.5 MB at E8), issued eagerly and
    running beside -> [backward of the encoder] -> all-reduce of the tail -> [clip + Adam] -- five host calls per step
    instead of ~320 launches, with two thirds of the exchange under the encoder's backward.  (Gradient accumulation or a
    model without the fused conv stack: two graphs around one whole-buffer all-reduce.)
"""
import os
import time
import warnings

import torch
import torch.nn as nn

from ..util.stft_loss import MultiResolutionSTFTLoss
from ..util.util import LinearWarmupCosineDecay, loss_fn

DEFAULT_OPTIM = {"n_iters": 1600000, "learning_rate": 1e-4, "betas": (0.9, 0.999), "eps": 1e-8,
                 "clip_grad_norm_max": 10, "weight_decay": 0, "fused_adam": True}
DEFAULT_LOSS = {"ell_p": 1, "ell_p_lambda": 1, "stft_lambda": 1,
                "stft_config": {"sc_lambda": 0.5, "mag_lambda": 0.5, "band": "full",
                                "hop_sizes": [50, 120, 240], "win_lengths": [240, 600, 1200],
                                "fft_sizes": [512, 1024, 2048]}}
GRAPH_WARMUP_STEPS = 3       # eager steps before the capture (lazy initialisation, allocator warm-up)


class TrainStep:
    """``step = TrainStep(net, ...); loss, grad_norm = step(clean, noisy)``.

    repeats: gradient-accumulation micro-steps per optimizer step (reference: ``repeats`` of
      src/training/train.py:282-300).  ``__call__`` splits the batch it is given into ``repeats`` equal micro-batches;
      ``micro_step`` / ``optimizer_step`` expose the two halves for loaders that deliver micro-batches one by one.
    flat_optimizer: None = on for fp32 CUDA models.  Off: torch.optim.Adam + clip_grad_norm_ + GradScaler.
    use_graph: None = on for CUDA runs with the flat optimizer (one graph for one process; two graphs around an eager
      whole-buffer all-reduce for several ranks, see the module docstring)."""

    def __init__(self, net, optimization=None, loss_config=None, autocast_dtype=None, iteration=0, repeats=1,
                 flat_optimizer=None, use_graph=None):
        self.net = net
        self.opt_cfg = dict(DEFAULT_OPTIM, **(optimization or {}))
        self.loss_cfg = dict(DEFAULT_LOSS, **(loss_config or {}))
        self.repeats = int(repeats)
        dev = next(net.parameters()).device
        self.buckets = getattr(net, "grad_buckets", None)
        if flat_optimizer is None:
            flat_optimizer = dev.type == "cuda" and all(p.dtype == torch.float32 for p in net.parameters())
        self.autocast_dtype = autocast_dtype
        fp16 = autocast_dtype == torch.float16
        self.scaler = None
        if flat_optimizer:
            from .flat_optim import FlatAdam
            from .train_distributed import GradBuckets
            if self.buckets is None:          # single process: flat storage only, no exchange
                self.buckets = net.grad_buckets = GradBuckets(net)
            self.optimizer = FlatAdam(self.buckets.flat, lr=self.opt_cfg["learning_rate"],
                                      betas=tuple(self.opt_cfg["betas"]), eps=self.opt_cfg["eps"],
                                      weight_decay=self.opt_cfg["weight_decay"],
                                      max_grad_norm=self.opt_cfg["clip_grad_norm_max"], loss_scaling=fp16)
        else:
            fused = bool(self.opt_cfg["fused_adam"]) and dev.type == "cuda"
            self.optimizer = torch.optim.Adam(net.parameters(), lr=self.opt_cfg["learning_rate"],
                                              betas=tuple(self.opt_cfg["betas"]), eps=self.opt_cfg["eps"],
                                              fused=fused, weight_decay=self.opt_cfg["weight_decay"])
            # fp16 autocast needs loss scaling (reference: GradScaler, train.py:158-160); bf16 does not
            self.scaler = torch.amp.GradScaler("cuda") if fp16 else None
        self.flat = flat_optimizer
        self.scheduler = LinearWarmupCosineDecay(self.optimizer, lr_max=self.opt_cfg["learning_rate"],
                                                 n_iter=self.opt_cfg["n_iters"], iteration=iteration, divider=25,
                                                 warmup_proportion=0.05, phase=("linear", "cosine"))
        self.mrstft = None
        if self.loss_cfg["stft_lambda"] > 0:
            self.mrstft = MultiResolutionSTFTLoss(**self.loss_cfg["stft_config"]).to(dev)
        world = self.buckets.world if self.buckets is not None else 1
        if use_graph is None:
            # One process: the whole step replays from one hipGraph.  Several ranks: the eager step, whose per-bucket
            # all-reduce overlaps the backward -- the captured form ([graph] -> one whole-buffer all-reduce -> [graph],
            # `use_graph=True`) trades that overlap for three host calls per step and has never run on two or more real
            # GPUs over RCCL, so it stays opt-in until a multi-GPU run shows it correct and faster.
            exchanging = self.buckets is not None and self.buckets.exchanging
            use_graph = self.flat and dev.type == "cuda" and not exchanging
        if use_graph and not self.flat:
            raise ValueError("use_graph needs the flat optimizer")
        self.use_graph = bool(use_graph)
        self.world = world
        self._graph = None              # {"graph", "optim_graph", "clean", "noisy", "loss"} | {"failed": error}
        self._eager_steps = 0
        self.host_seconds = 0.0         # wall time the host spent inside __call__ (enqueueing; nothing here synchronises)
        self.calls = 0
        if self.flat:
            self.optimizer.on_hyper_change(self.drop_graph)

    def drop_graph(self):
        """Forget the captured step: it bakes in betas / eps / weight decay / clip norm / repeats / the loss config."""
        self._graph = None
        self._eager_steps = 0

    # ------------------------------------------------------------------ pieces
    def zero_grad(self):
        if self.buckets is not None:
            self.buckets.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=True)

    def _loss(self, clean_audio, noisy_audio):
        kw = {k: v for k, v in self.loss_cfg.items() if k != "stft_config"}
        if self.autocast_dtype is not None:
            with torch.autocast(device_type="cuda", dtype=self.autocast_dtype):
                return loss_fn(self.net, (clean_audio, noisy_audio), mrstftloss=self.mrstft, **kw)[0]
        return loss_fn(self.net, (clean_audio, noisy_audio), mrstftloss=self.mrstft, **kw)[0]

    def micro_step(self, clean_audio, noisy_audio, last=True):
        """Forward + backward of one micro-batch; gradients accumulate.  ``last``: this backward closes the
        accumulation window, so it carries the gradient exchange."""
        if self.buckets is not None:
            self.buckets.require_sync = bool(last)
        loss = self._loss(clean_audio, noisy_audio)
        scaled = loss / self.repeats if self.repeats > 1 else loss
        if self.flat:
            fp = self.buckets.flat
            fp.armed = True                    # kernels may write parameter gradients straight into the flat buffer
            try:
                self.optimizer.scale_loss(scaled).backward()
            finally:
                fp.armed = False
        elif self.scaler is not None:
            self.scaler.scale(scaled).backward()
        else:
            scaled.backward()
        return loss.detach()

    def _backward(self, scaled, tensors=None, grads=None):
        """backward() with the kernels' gradient sinks armed (flat optimizer) / through the GradScaler."""
        def run():
            if tensors is not None:
                torch.autograd.backward(tensors, grads)
            else:
                scaled.backward()
        if self.flat:
            fp = self.buckets.flat
            fp.armed = True
            try:
                run()
            finally:
                fp.armed = False
        else:
            run()

    def _encoder_cut_offset(self):
        """Element offset in the flat buffers where the encoder's parameters start, if they form its tail (they do: the
        flat order is the reverse of the registration order and the encoder registers first) and the model runs the
        fused conv stack, which is where the cut lives; else None."""
        enc = getattr(self.net, "encoder", None)
        if enc is None or not self.flat or not getattr(self.net, "use_fused_convs", False):     # (absent: not our model)
            return None
        fp = self.buckets.flat
        ids = {id(p) for p in enc.parameters()}
        if not ids:
            return None
        inside = [o for p, o in zip(fp.params, fp.offsets) if id(p) in ids]
        outside = [o for p, o in zip(fp.params, fp.offsets) if id(p) not in ids]
        if len(inside) != len(ids) or not outside or max(outside) >= min(inside):
            return None
        return min(inside)

    def optimizer_step(self, write_lr=True):
        """unscale -> clip -> Adam -> scale update.  Returns the gradient norm (device scalar)."""
        if self.flat:
            self.optimizer.step(write_lr=write_lr)
            return self.optimizer.grad_norm
        if self.scaler is not None:
            self.scaler.unscale_(self.optimizer)
            grad_norm = nn.utils.clip_grad_norm_(self.net.parameters(), self.opt_cfg["clip_grad_norm_max"])
            self.scaler.step(self.optimizer)
            self.scaler.update()
        else:
            grad_norm = nn.utils.clip_grad_norm_(self.net.parameters(), self.opt_cfg["clip_grad_norm_max"])
            self.optimizer.step()
        return grad_norm

    def _body(self, clean_audio, noisy_audio, write_lr=True):
        loss = self._micro_steps(clean_audio, noisy_audio, exchange=True)
        return loss, self.optimizer_step(write_lr=write_lr)

    # ------------------------------------------------------------------ hipGraph
    @property
    def graph_status(self):
        if not self.use_graph:
            return "off"
        if self._graph is None:
            return "pending"
        return "failed: " + self._graph["failed"] if "failed" in self._graph else "captured"

    def _micro_steps(self, clean_audio, noisy_audio, exchange):
        """zero_grad + the ``repeats`` micro-steps, with or without the in-backward gradient exchange."""
        self.zero_grad()
        if self.repeats == 1:
            if exchange:
                return self.micro_step(clean_audio, noisy_audio)
            return self.micro_step(clean_audio, noisy_audio, last=False)
        if clean_audio.shape[0] % self.repeats:
            raise ValueError(f"batch of {clean_audio.shape[0]} clips does not split into {self.repeats} micro-batches")
        parts = zip(clean_audio.chunk(self.repeats), noisy_audio.chunk(self.repeats))
        losses = [self.micro_step(c, n, last=(exchange and i == self.repeats - 1)) for i, (c, n) in enumerate(parts)]
        return torch.stack(losses).mean()

    def _capture(self, clean_audio, noisy_audio):
        g = {"clean": clean_audio.clone(), "noisy": noisy_audio.clone()}
        try:
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            if not (self.buckets is not None and self.buckets.exchanging):
                with torch.cuda.graph(graph):
                    loss, norm = self._body(g["clean"], g["noisy"], write_lr=False)
                g.update(graph=graph, loss=loss, norm=norm)
            else:
                # several ranks: no collective inside a capture.  Graph 1 leaves this rank's gradient in the flat buffer,
                # the exchange runs eagerly between the replays, graph 2 is the optimizer section.
                # (thread_local: the process group's watchdog thread may query events while this thread captures;
                #  in the default "global" mode such a call from another thread invalidates the capture)
                cut = self._encoder_cut_offset()
                if cut is not None and self.repeats == 1:
                    # THREE graphs: [zero_grad, forward, loss, backward of decoder + bottleneck] -> all-reduce of their
                    # gradients (the head of the flat buffer) issued eagerly, running beside -> [backward of the encoder]
                    # -> all-reduce of the tail -> [clip + Adam].  The autograd graph is cut at the encoder's outputs
                    # (network/CleanUMamba.py _forward_fused): graph A ends with their gradients in hand, graph B feeds
                    # them to the encoder's backward.
                    holder = []
                    self.net.__dict__["_encoder_cut"] = holder
                    try:
                        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                            self.zero_grad()
                            self.buckets.require_sync = False
                            loss = self._loss(g["clean"], g["noisy"])
                            if len(holder) != 1:
                                raise RuntimeError("the forward did not pass the encoder cut exactly once")
                            outs, leaves = holder[0]
                            self._backward(self.optimizer.scale_loss(loss))
                            dys = [t.grad for t in leaves]
                    finally:
                        self.net.__dict__.pop("_encoder_cut", None)
                    enc_graph = torch.cuda.CUDAGraph()
                    try:                       # (require_sync stays off: the encoder's backward announces its gradients too)
                        with torch.cuda.graph(enc_graph, pool=graph.pool(), capture_error_mode="thread_local"):
                            live = [(o, d) for o, d in zip(outs, dys) if d is not None]
                            self._backward(None, tensors=[o for o, _ in live], grads=[d for _, d in live])
                    finally:
                        self.buckets.require_sync = True
                    loss = loss.detach()
                    g.update(enc_graph=enc_graph, cut=cut, keep=(outs, leaves, dys))
                else:
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        loss = self._micro_steps(g["clean"], g["noisy"], exchange=False)
                optim_graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(optim_graph, pool=graph.pool(), capture_error_mode="thread_local"):
                    norm = self.optimizer_step(write_lr=False)
                g.update(graph=graph, optim_graph=optim_graph, loss=loss, norm=norm)
        except Exception as exc:          # noqa: BLE001 - capture is an optimisation; stay eager
            if os.environ.get("CUM_DEBUG_CAPTURE") == "1":        # debugging: surface the capture error instead of going eager
                raise
            g = {"failed": repr(exc)}
            warnings.warn(f"TrainStep: hipGraph capture of the train step failed ({exc!r}); steps run eagerly")
        if self.buckets is not None and self.buckets.exchanging:
            # The replayed step issues ONE whole-buffer all-reduce, the eager step one per bucket: ranks that disagree
            # about which of the two they run would pair different collectives (hang, or averages of the wrong buffers).
            # Every rank reaches this point at the same step (the warm-up count is the same everywhere), so an eager
            # MIN-reduction of "my capture worked" -- issued outside any capture -- makes the choice collective.
            if not self.buckets.all_ranks_agree("graph" in g):
                if "graph" in g:
                    g = {"failed": "another rank could not capture the train step"}
                    warnings.warn("TrainStep: another rank failed to capture the train step; every rank runs eagerly")
        self._graph = g

    def __call__(self, clean_audio, noisy_audio):
        """Returns (loss tensor on device, grad_norm tensor)."""
        t0 = time.perf_counter()
        g = self._graph
        if self.use_graph and g is None and self._eager_steps >= GRAPH_WARMUP_STEPS:
            self._capture(clean_audio, noisy_audio)       # capture does not execute: the replay below is this step
            g = self._graph
        replay = self.use_graph and g is not None and "graph" in g
        if replay and not ((g["clean"].shape, g["clean"].dtype) == (clean_audio.shape, clean_audio.dtype)
                           and (g["noisy"].shape, g["noisy"].dtype) == (noisy_audio.shape, noisy_audio.dtype)):
            if "optim_graph" in g:
                # several ranks: falling back to the eager step on THIS rank alone would mismatch the collectives
                raise RuntimeError(
                    f"TrainStep: the captured multi-rank step takes batches of {tuple(g['clean'].shape)} "
                    f"{g['clean'].dtype}; got {tuple(clean_audio.shape)} {clean_audio.dtype}.  Keep the batch shape "
                    "fixed (drop the last partial batch) or build the step with use_graph=False")
            replay = False
        if replay:
            self.buckets.flat.require_intact()
            g["clean"].copy_(clean_audio)
            g["noisy"].copy_(noisy_audio)
            self.optimizer.write_lr()
            g["graph"].replay()
            if "enc_graph" in g:
                wait_head = self.buckets.exchange_range(0, g["cut"])        # beside the encoder's backward
                g["enc_graph"].replay()
                wait_tail = self.buckets.exchange_range(g["cut"], self.buckets.flat.numel)
                wait_head()
                wait_tail()
                g["optim_graph"].replay()
            elif "optim_graph" in g:
                self.buckets.exchange_all()
                g["optim_graph"].replay()
            self.buckets.flat.bump_versions()             # the replay moved the parameters behind autograd's back
            loss, grad_norm = g["loss"].clone(), g["norm"].clone()
        else:
            loss, grad_norm = self._body(clean_audio, noisy_audio)
            self._eager_steps += 1
        self.scheduler.step()
        self.host_seconds += time.perf_counter() - t0
        self.calls += 1
        return loss, grad_norm
