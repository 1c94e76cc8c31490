import sys, torch
sys.path.insert(0, '.')
from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
from oracle import mamba_ref as M
dev = torch.device('cuda')
for shape in [(1,1,1,1),(1,1,8,1),(1,64,8,1),(1,64,16,3),(2,70,20,20)]:
    bsz, dim, N, L = shape
    g = torch.Generator().manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g)
    cpu = dict(u=rn(bsz, L, dim).transpose(1, 2), delta=0.5*rn(bsz, L, dim).transpose(1, 2), A=-torch.exp(0.5*rn(dim, N)),
               B=rn(bsz, L, N).transpose(1, 2), C=rn(bsz, L, N).transpose(1, 2))
    dout = rn(bsz, L, dim).transpose(1, 2)
    ref = {k: v.double().detach().requires_grad_(True) for k, v in cpu.items()}
    yr = M.selective_scan_ref(ref['u'], ref['delta'], ref['A'], ref['B'], ref['C'], delta_softplus=True)
    (yr*dout.double()).sum().backward()
    d = {k: v.to(dev).requires_grad_(True) for k, v in cpu.items()}
    y = selective_scan_fn(d['u'], d['delta'], d['A'], d['B'], d['C'], delta_softplus=True)
    (y*dout.to(dev)).sum().backward()
    for k in ('B','C'):
        a, b = d[k].grad.cpu().double(), ref[k].grad
        ratio = (a/b).flatten()
        print(shape, k, 'ratio min/max', ratio.min().item(), ratio.max().item(), 'first', ratio[:10].tolist())
