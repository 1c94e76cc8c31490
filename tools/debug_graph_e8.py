"""Debug: E8 f16 graph replay vs eager, per-step losses (tests/test_train_gpu.py::test_benched_configuration_...)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import synth
from cleanumamba_amd.network import Net
from cleanumamba_amd.training.train_step import TrainStep
E8 = dict(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8, kernel_size=4,
          stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)
cuda = torch.device("cuda:0")
from cleanumamba_amd.training import train_step as ts
_orig = ts.loss_fn
KEEP = {}
import torch.nn.functional as F
class _NetSpy:
    def __init__(self, net): self.net = net
    def __call__(self, x):
        y = self.net(x)
        KEEP[("y", id(self.net))] = y.detach()
        return y
def _spy(net, X, **kw):
    loss, dic = _orig(_NetSpy(net), X, **kw)
    KEEP[id(net)] = (dic, loss.detach())
    KEEP[("clean", id(net))] = X[0]
    return loss, dic
ts.loss_fn = _spy
import cleanumamba_amd.util.util as uu
VAR = os.environ.get("VARIANT", "torch")
class _F:
    def __getattr__(self, k): return getattr(F, k)
    @staticmethod
    def l1_loss(y, c):
        if VAR == "twostage":
            return (y - c).abs().view(-1, 500).sum(1).sum() / y.numel()
        if VAR == "clone_y":
            return F.l1_loss(y.clone(), c)
        if VAR == "clone_c":
            return F.l1_loss(y, c.clone())
        if VAR == "sum":
            return (y - c).abs().sum() / y.numel()
        return F.l1_loss(y, c)
uu.F = _F()
nan_at = int(os.environ.get("NAN_AT", "5"))
L = int(os.environ.get("CLIP", "160000"))
nets, steps = [], []
for graph in (True, False):
    torch.manual_seed(0)
    nets.append(Net("CleanUMamba", E8).to(cuda).train())
    steps.append(TrainStep(nets[-1], autocast_dtype=torch.float16, use_graph=graph))
for it in range(5):
    clean, noisy = synth.waveform(2, L, seed=40 + it)
    if it == nan_at:
        noisy[1, 0, 777] = float("nan")
    row = []
    for k in range(2):
        loss, gn = steps[k](clean.to(cuda), noisy.to(cuda))
        sv = steps[k].optimizer.state_vec.cpu()
        dic, lraw = KEEP[id(nets[k])]
        yk, ck = KEEP[("y", id(nets[k]))], KEEP[("clean", id(nets[k]))]
        with torch.no_grad():
            re = float(F.l1_loss(yk, ck)); re2 = float(F.l1_loss(yk, clean.to(cuda)))
        row.append((float(loss), {kk: round(float(v), 5) for kk, v in dic.items()}, "recomputed l1 from kept buffers", round(re, 5), "vs fed clean", round(re2, 5), yk.shape, ck.data_ptr() == (steps[k]._graph or {}).get("clean", ck).data_ptr() if k == 0 else None))
    worst = max(((pa - pb).norm() / pb.norm()).item() for pa, pb in zip(nets[0].parameters(), nets[1].parameters()))
    print(it, row, "param rel", worst, flush=True)
print(steps[0].graph_status)
