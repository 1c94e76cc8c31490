"""Data-parallel gradient exchange over RCCL/xGMI -- interface of
src/training/train_distributed.py:44-60, 97-149 (init_distributed, reduce_tensor,
apply_gradient_allreduce).

The reference flattens all 41.4 M gradients into one buffer AFTER backward, runs a
single blocking all-reduce, divides, and copies everything back (two extra
full-gradient passes over HBM, no overlap).  Here, one process per GPU:
  * gradients live permanently in a few flat fp32 buckets (``p.grad`` is a view),
    so there is no flatten and no copy-back;
  * a bucket's all-reduce is issued from a post-accumulate hook as soon as its last
    gradient has been produced, on the communicator's own stream, overlapping the
    rest of backward; buckets are filled in reverse registration order, which is
    the order backward produces gradients (decoder first);
  * averaging uses ReduceOp.AVG on RCCL (SUM + scale on gloo); backward's end
    callback only waits for the outstanding handles.
xGMI is point-to-point (7 links per GPU), so the default 32 MiB bucket keeps each
collective large enough to be link-bandwidth bound rather than latency bound while
still giving ~5 overlappable pieces for the 165 MB of E8 gradients.
"""
import os

import torch
import torch.distributed as dist
from torch.autograd import Variable


def reduce_tensor(tensor, num_gpus):
    rt = tensor.clone()
    dist.all_reduce(rt, op=dist.ReduceOp.SUM)
    rt /= num_gpus
    return rt


def init_distributed(rank, num_gpus, group_name=None, dist_backend="nccl", dist_url="tcp://127.0.0.1:54321"):
    """One process per GPU.  ``dist_backend="nccl"`` is RCCL on ROCm; "gloo" runs on CPU (tests)."""
    if dist_backend == "nccl":
        assert torch.cuda.is_available(), "Distributed mode on the nccl/RCCL backend requires a GPU."
        torch.cuda.set_device(rank % torch.cuda.device_count())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not dist.is_initialized():
        dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank)


class GradBuckets:
    """Flat gradient storage + overlapped all-reduce for one module."""

    def __init__(self, module, bucket_bytes=32 << 20, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.params = [p for p in module.parameters() if p.requires_grad]
        backend = dist.get_backend(process_group)
        self.use_avg = backend == "nccl"
        self.buckets = []          # (flat, [params], pending_count)
        self.where = {}            # id(param) -> (bucket index, view)
        self.handles = []
        self._armed = False
        order = list(reversed(self.params))
        by_key = {}
        for p in order:
            by_key.setdefault((p.dtype, p.device), []).append(p)
        for (dtype, device), plist in by_key.items():
            cap = max(1, bucket_bytes // torch.empty((), dtype=dtype).element_size())
            cur, n = [], 0
            for p in plist:
                if cur and n + p.numel() > cap:
                    self._make_bucket(cur, dtype, device)
                    cur, n = [], 0
                cur.append(p)
                n += p.numel()
            if cur:
                self._make_bucket(cur, dtype, device)
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    def _make_bucket(self, plist, dtype, device):
        flat = torch.zeros(sum(p.numel() for p in plist), dtype=dtype, device=device)
        idx, off = len(self.buckets), 0
        for p in plist:
            view = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self.where[id(p)] = (idx, view)
            p.grad = view
        self.buckets.append([flat, plist, len(plist)])

    def zero_grad(self):
        for flat, plist, _ in self.buckets:
            flat.zero_()
            for p in plist:
                p.grad = self.where[id(p)][1]

    def _hook(self, p):
        idx, view = self.where[id(p)]
        if p.grad is not view:
            # someone replaced .grad (e.g. zero_grad(set_to_none=True)): fold it back into the bucket
            if p.grad is not None and p.grad.data_ptr() != view.data_ptr():
                view.copy_(p.grad)
            p.grad = view
        if not self._armed:
            self._armed = True
            for b in self.buckets:
                b[2] = len(b[1])
            Variable._execution_engine.queue_callback(self._finish)
        b = self.buckets[idx]
        b[2] -= 1
        if b[2] == 0:
            self._launch(b[0])

    def _launch(self, flat):
        if self.world == 1:
            return
        if self.use_avg:
            self.handles.append((dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None))
        else:
            self.handles.append((dist.all_reduce(flat, group=self.group, async_op=True), flat))

    def _finish(self):
        # buckets whose parameters did not all receive a gradient this pass are still reduced,
        # so that every rank issues the same sequence of collectives
        for b in self.buckets:
            if b[2] != 0:
                self._launch(b[0])
                b[2] = 0
        for h, flat in self.handles:
            h.wait()
            if flat is not None:
                flat /= self.world
        self.handles = []
        self._armed = False


def apply_gradient_allreduce(module, bucket_bytes=32 << 20):
    """Broadcast rank 0's parameters/buffers, then all-reduce (average) gradients during every
    backward.  Does not change the module's class; returns the module (reference contract).
    The bucket manager is exposed as ``module.grad_buckets`` (use its zero_grad())."""
    with torch.no_grad():
        for t in module.state_dict().values():
            if torch.is_tensor(t):
                dist.broadcast(t, 0)
    module.grad_buckets = GradBuckets(module, bucket_bytes)
    return module
