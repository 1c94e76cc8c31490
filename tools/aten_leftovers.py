"""Which Python lines of the train step still launch ATen kernels (GPU box): a TorchDispatchMode over one eager step records
every aten op that touches a CUDA tensor with the innermost frame inside this repo; ops that only make views / allocate
are skipped.  (Autograd's own accumulation runs without a Python frame: reported as <autograd engine>.)"""
import collections
import sys
import traceback
sys.path.insert(0, ".")
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from cleanumamba_amd.network import Net
from cleanumamba_amd.training.train_step import TrainStep
import bench
dev = torch.device("cuda")
torch.manual_seed(0)
net = Net("CleanUMamba", bench.E8).to(dev).train()
step = TrainStep(net, autocast_dtype=torch.float16, use_graph=False)
g = torch.Generator(device=dev).manual_seed(1234)
clean = 0.05 * torch.randn(16, 1, bench.CLIP, generator=g, device=dev)
noisy = clean + 0.05 * torch.randn(16, 1, bench.CLIP, generator=g, device=dev)
for _ in range(6):
    step(clean, noisy)
torch.cuda.synchronize()
SKIP = ("view", "reshape", "empty", "as_strided", "slice", "select", "transpose", "permute", "expand", "t.default", "detach",
        "alias", "unsqueeze", "squeeze", "_unsafe_view", "split", "unbind", "narrow", "record_stream", "is_", "size", "stride",
        "_local_scalar_dense", "unfold", "lift_fresh", "chunk", "movedim", "diagonal", "resize_", "set_", "_to_copy_meta", "sym_")
acc = collections.Counter()
sizes = {}
shapes = collections.defaultdict(set)      # for the ops without a Python frame: which tensors they touch


class Mode(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not any(s in name for s in SKIP):
            ts = [a for a in list(args) + list((kwargs or {}).values()) if isinstance(a, torch.Tensor)]
            if isinstance(out, torch.Tensor):
                ts.append(out)
            if any(t.is_cuda for t in ts):
                fr = "<autograd engine>"
                for f in reversed(traceback.extract_stack()[:-1]):
                    if ("cleanumamba_amd" in f.filename or f.filename.endswith("bench.py")) and "aten_leftovers" not in f.filename:
                        fr = f"{f.filename.split('cleanumamba_amd/')[-1]}:{f.lineno} {f.name}"
                        break
                acc[(name, fr)] += 1
                sizes[(name, fr)] = max(sizes.get((name, fr), 0), max((t.numel() for t in ts), default=0))
                if fr == "<autograd engine>":
                    shapes[(name, fr)].add(" ".join(f"{tuple(t.shape)}:{str(t.dtype).replace('torch.', '')}" for t in ts))
        return out


with Mode():
    step(clean, noisy)
torch.cuda.synchronize()
for (name, fr), n in sorted(acc.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"{n:4d}  {name:34s} max numel {sizes[(name, fr)]:>11d}  {fr}")
    for sh in sorted(shapes.get((name, fr), ())):
        print(f"          {sh}")
print("total", sum(acc.values()))
