"""Flat parameter / gradient storage and the clip + Adam (+ loss scaling) step on it.

Optimizer section of the reference's hot loop (src/training/train.py:145-160, 303-310):
``scaler.unscale_`` -> ``clip_grad_norm_(10)`` -> ``scaler.step(Adam)`` -> ``scaler.update``.  There it runs over
103 separate tensors; here every parameter is a view of ONE flat fp32 buffer and every ``p.grad`` a view of a
second one, so that

  * the gradient exchange all-reduces contiguous slices (training/train_distributed.py) with no flatten / copy-back,
  * the norm, the clip, the unscale, the inf check, the loss-scale update and Adam are three kernel launches
    (csrc/optim.hip) with nothing returning to the host -- which is also what lets the whole train step be
    captured in a hipGraph (training/train_step.py).

The host only writes the learning rate into the device-side state vector before each step.
"""
import ctypes
import weakref

import torch

from .. import hip

# layout of the device-side state vector (csrc/optim.hip)
ST_NORM, ST_MULT, ST_FOUND_INF, ST_SCALE, ST_TRACKER, ST_STEP, ST_BC1, ST_BC2_SQRT, ST_LR, ST_SKIPPED = range(10)


_SINKS = {}          # parameter data_ptr -> weakref(FlatParams) that owns it


def _forget(ptrs, ref):
    """A FlatParams was collected: drop its entries (unless a newer FlatParams has taken the address over)."""
    for ptr in ptrs:
        if _SINKS.get(ptr) is ref:
            del _SINKS[ptr]


def sink_of(param):
    """The FlatParams whose flat buffers hold ``param`` (None if the parameter is not flat-managed)."""
    ref = _SINKS.get(param.data_ptr())
    flat = ref() if ref is not None else None
    if flat is None or flat.by_ptr.get(param.data_ptr()) is None:
        return None
    return flat


class FlatParams:
    """Moves the parameters of ``module`` into one flat fp32 buffer (``p.data`` becomes a view) and gives every
    parameter a gradient view of a second flat buffer.  Parameters are laid out in REVERSE registration order:
    that is the order backward produces gradients in (decoder first), so slices from the front of the gradient
    buffer complete first.  Every parameter starts 16-byte aligned (zero padding in between)."""

    ALIGN = 4      # elements

    def __init__(self, module):
        params = [p for p in module.parameters() if p.requires_grad]
        if not params:
            raise ValueError("no trainable parameters")
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("FlatParams needs fp32 parameters on one device (master weights are fp32, "
                                   "as in the reference's autocast training)")
        self.params = list(reversed(params))
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.grad_views = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = self.data[o:o + p.numel()].view_as(p)
                view.copy_(p.detach())
                p.data = view
                gview = self.grad[o:o + p.numel()].view_as(p)
                if p.grad is not None:
                    gview.copy_(p.grad)
                p.grad = gview
                self.grad_views.append(gview)
        self.index = {id(p): i for i, p in enumerate(self.params)}
        self.by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        # fresh[i]: nothing has been accumulated into gradient i since zero_grad() -- a producer may then WRITE its
        # result into the view (GradSink below) instead of handing it to autograd's AccumulateGrad (read-read-write)
        self.fresh = [False] * len(self.params)
        ref = weakref.ref(self)
        ptrs = [p.data_ptr() for p in self.params]
        for ptr in ptrs:
            _SINKS[ptr] = ref
        weakref.finalize(self, _forget, ptrs, ref)

    def intact(self):
        """False once somebody re-pointed a parameter (``.to()``, pruning, load_pruned_state_dict ...)."""
        base = self.data.data_ptr()
        return all(p.data_ptr() == base + 4 * o for p, o in zip(self.params, self.offsets))

    def require_intact(self):
        """The optimizer, the gradient sinks and the exchange all work on the flat buffers: once a parameter points
        somewhere else (``net.to()`` / ``.half()``, ``load_pruned_state_dict``, pruning, ``p.data = ...``) they would
        silently keep training an orphaned copy.  Raise instead."""
        if not self.intact():
            raise RuntimeError("a parameter was re-allocated after FlatParams took it over (net.to() / .half() / pruning / "
                               "load_pruned_state_dict after the TrainStep was built): its flat view is orphaned and the "
                               "live parameter would silently stop training; build a new TrainStep / GradBuckets")

    def bump_versions(self):
        """The kernels update the flat buffer through raw pointers, which no autograd version counter sees.  Caches
        keyed on ``p._version`` (packed conv weights, the captured streaming hop's weight copies, -exp(A_log)) would
        go stale: count the update on every parameter (6 us for 103 tensors)."""
        torch._C._increment_version(self.params)

    def attach_grads(self):
        """Make every ``p.grad`` the bucket view again (after ``zero_grad(set_to_none=True)`` or a foreign ``.grad``)."""
        for p, g in zip(self.params, self.grad_views):
            if p.grad is not g:
                if p.grad is not None and p.grad.data_ptr() != g.data_ptr():
                    g.copy_(p.grad)
                p.grad = g

    def zero_grad(self):
        self.grad.zero_()
        self.attach_grads()
        self.fresh = [True] * len(self.params)

    # ---- gradient sink: kernels write parameter gradients straight into the flat buffer
    def slot(self, param):
        """(index, element offset) of ``param`` if its gradient view may be overwritten now, else None."""
        i = self.by_ptr.get(param.data_ptr())
        if not self.armed or i is None or not self.fresh[i] or param.grad is not self.grad_views[i]:
            return None
        return i, self.offsets[i]

    # The sink is taken only inside a backward that its owner started (TrainStep.micro_step arms it): any other
    # backward -- torch.autograd.grad(loss, params), a user's own loss.backward() -- sees plain autograd semantics
    # (gradients returned / accumulated by AccumulateGrad), not None results and silently overwritten .grad views.
    armed = False

    def wrote(self, indices):
        """The gradients of these parameters now sit in their views (they are no longer fresh); tell the exchange."""
        for i in indices:
            self.fresh[i] = False
        if self.on_write is not None:
            self.on_write([self.params[i] for i in indices])

    on_write = None

    def slices(self, max_bytes):
        """Contiguous (start, end, [param indices]) ranges of at most ``max_bytes`` (cut at parameter boundaries)."""
        cap = max(1, max_bytes // 4)
        out, start, members = [], 0, []
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            end = o + (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            if members and end - start > cap:
                out.append((start, o, members))
                start, members = o, []
            members.append(i)
        out.append((start, self.numel, members))
        return out


class FlatAdam:
    """Adam with gradient-norm clipping and (optionally) dynamic loss scaling on a FlatParams.

    Arithmetic of ``torch.optim.Adam`` (no amsgrad; weight decay in the L2 form), ``clip_grad_norm_`` and
    ``torch.amp.GradScaler`` (init_scale 65536, growth 2 every 2000 clean steps, backoff 0.5; a step whose
    gradients hold inf / nan is skipped).  ``param_groups`` has one group so that the LR schedulers written
    for torch optimizers (``group["lr"] = ...``) drive it unchanged."""

    def __init__(self, flat, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, max_grad_norm=0.0,
                 loss_scaling=False, init_scale=65536.0, growth_factor=2.0, backoff_factor=0.5, growth_interval=2000):
        self.flat = flat
        self.param_groups = [{"params": flat.params, "lr": lr, "betas": tuple(betas), "eps": eps,
                              "weight_decay": weight_decay}]
        self.max_grad_norm = float(max_grad_norm or 0.0)
        self.loss_scaling = bool(loss_scaling)
        self.growth, self.backoff, self.growth_interval = growth_factor, backoff_factor, int(growth_interval)
        dev = flat.data.device
        lib = hip.lib()
        self.exp_avg = torch.zeros_like(flat.data)
        self.exp_avg_sq = torch.zeros_like(flat.data)
        self.state_vec = torch.zeros(lib.cum_optim_state_elems(), dtype=torch.float32, device=dev)
        self.state_vec[ST_SCALE] = init_scale if loss_scaling else 1.0
        self.state_vec[ST_LR] = lr
        self.nparts = lib.cum_optim_sumsq_parts(flat.numel)
        self.partials = torch.zeros(self.nparts, dtype=torch.float32, device=dev)
        self._lr_written = lr
        self._listeners = []

    # ---- loss scaling (device scalars: no host sync)
    def scale_loss(self, loss):
        return loss * self.state_vec[ST_SCALE] if self.loss_scaling else loss

    @property
    def loss_scale(self):
        return self.state_vec[ST_SCALE]

    @property
    def grad_norm(self):
        """Total norm of the (unscaled, unclipped) gradient of the last step -- a device scalar."""
        return self.state_vec[ST_NORM]

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def write_lr(self):
        """Push the host-side learning rate to the device (one fill kernel; call outside a captured graph)."""
        lr = float(self.param_groups[0]["lr"])
        self.state_vec[ST_LR:ST_LR + 1].fill_(lr)
        self._lr_written = lr

    def step(self, write_lr=True):
        """unscale + clip + Adam + scale update; three launches on the current stream."""
        g = self.param_groups[0]
        if write_lr:
            self.write_lr()
        lib, f = hip.lib(), self.flat
        f.require_intact()
        st = hip.stream_ptr()
        with torch.cuda.device(f.data.device):
            hip.check(lib.cum_optim_sumsq(hip.ptr(f.grad), f.numel, hip.ptr(self.partials), st))
            hip.check(lib.cum_optim_prepare(hip.ptr(self.state_vec), hip.ptr(self.partials), self.nparts,
                                            self.max_grad_norm, g["betas"][0], g["betas"][1], int(self.loss_scaling),
                                            self.growth, self.backoff, self.growth_interval, st))
            hip.check(lib.cum_optim_adam(hip.ptr(f.data), hip.ptr(f.grad), hip.ptr(self.exp_avg),
                                         hip.ptr(self.exp_avg_sq), f.numel, hip.ptr(self.state_vec), g["betas"][0],
                                         g["betas"][1], g["eps"], g["weight_decay"], st))
        f.bump_versions()

    # ---- checkpoint format of torch.optim.Adam (src/training/train.py:183-186, 367: optimizer_state_dict)
    def state_dict(self):
        f = self.flat
        order = {id(p): i for i, p in enumerate(reversed(f.params))}       # registration order, as torch numbers them
        step = self.state_vec[ST_STEP].detach().clone()
        state = {}
        for p, o in zip(f.params, f.offsets):
            n = p.numel()
            state[order[id(p)]] = {"step": step.clone(), "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                                   "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        g = dict(self.param_groups[0])
        g["params"] = list(range(len(f.params)))
        return {"state": state, "param_groups": [g], "flat_state": self.state_vec.detach().clone(),
                "loss_scaling": bool(self.loss_scaling)}

    def load_state_dict(self, sd):
        """Moments and the step count come from the checkpoint; the loss scale and its growth tracker only when both the
        checkpoint and this optimizer use loss scaling (a bf16 / f32 run saved scale 1.0: loaded into an fp16 run it
        would underflow the gradients for thousands of steps); the learning rate is the scheduler's to write."""
        f = self.flat
        order = {id(p): i for i, p in enumerate(reversed(f.params))}
        step = None
        with torch.no_grad():
            for p, o in zip(f.params, f.offsets):
                ent = sd["state"].get(order[id(p)])
                if ent is None:
                    continue
                n = p.numel()
                self.exp_avg[o:o + n].copy_(ent["exp_avg"].reshape(-1))
                self.exp_avg_sq[o:o + n].copy_(ent["exp_avg_sq"].reshape(-1))
                if step is None:
                    step = float(ent["step"])                     # one host read, not one per parameter
            saved = sd.get("flat_state")
            if saved is not None:
                saved = saved.detach().float().cpu()
                if step is None:
                    step = float(saved[ST_STEP])
                # the checkpoint scaled its loss too: said by its flag (a scale that had backed off to <= 1.0 is still a
                # loss scale and is restored); checkpoints written before the flag existed: inferred from scale > 1
                scaled = sd.get("loss_scaling")
                if scaled is None:
                    scaled = float(saved[ST_SCALE]) > 1.0
                if self.loss_scaling and scaled:
                    self.state_vec[ST_SCALE] = float(saved[ST_SCALE])
                    self.state_vec[ST_TRACKER] = float(saved[ST_TRACKER])
            if step is not None:
                self.state_vec[ST_STEP] = step
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
        t = float(step or 0.0)
        b1, b2 = self.param_groups[0]["betas"]
        if t > 0:
            self.state_vec[ST_BC1] = 1.0 - b1 ** t
            self.state_vec[ST_BC2_SQRT] = (1.0 - b2 ** t) ** 0.5
        self.hyper_changed()

    # ---- a captured train step bakes host-side values in (betas, eps, weight decay, clip norm): tell whoever captured
    def hyper_changed(self):
        for fn in list(self._listeners):
            fn()

    def on_hyper_change(self, fn):
        self._listeners.append(fn)
