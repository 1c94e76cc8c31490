"""Does a captured torch reduction replay correctly on this ROCm build?  (l1_loss over 320 000 elements inside the train-step graph)"""
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
for n in (16000, 160000):
    a = torch.randn(2, 1, n, device=dev)
    b = torch.randn(2, 1, n, device=dev)
    outs = {}
    def body():
        outs["l1"] = F.l1_loss(a, b)
        outs["absmean"] = (a - b).abs().mean()
        outs["abssum"] = (a - b).abs().sum()
        outs["two_stage"] = (a - b).abs().view(-1, 500).sum(1).sum() / (2 * n)
        outs["std"] = a.std(dim=2)
    with torch.no_grad():
        for _ in range(2):
            body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body()
        for it in range(3):
            a.copy_(torch.randn(2, 1, n, device=dev)); b.copy_(torch.randn(2, 1, n, device=dev))
            g.replay()
            torch.cuda.synchronize()
            got = {k: v.detach().flatten()[0].item() for k, v in outs.items()}
            want = {"l1": F.l1_loss(a, b).item(), "absmean": (a - b).abs().mean().item(), "abssum": (a - b).abs().sum().item(),
                    "two_stage": ((a - b).abs().view(-1, 500).sum(1).sum() / (2 * n)).item(), "std": a.std(dim=2).flatten()[0].item()}
            print(n, it, {k: (round(got[k], 6), round(want[k], 6)) for k in got}, flush=True)
