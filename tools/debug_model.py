import sys, json, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import load_golden, golden_json, load_ckpt, rel_l2
from oracle import synth
from cleanumamba_amd.network import CleanUMamba
dev = torch.device('cuda')
T = torch.from_numpy
if 'grads' in sys.argv:
    for name in ['e8_synth']:
        g = load_golden('e2e_' + name); meta = golden_json(g['meta'])
        net = CleanUMamba(**meta['cfg'])
        sd = synth.fill_state_dict(dict(zip(meta['keys'], meta['shapes'])), seed=meta['seed'])
        net.load_state_dict(sd); net = net.to(dev).train()
        clean, noisy = synth.waveform(2, meta['L'], seed=meta['wave_seed'])
        y = net(noisy.to(dev)); print('out rel', rel_l2(y, g['out64']))
        (y * clean.to(dev)).sum().backward()
        named = dict(net.named_parameters())
        for k in g:
            if k.startswith('grad:'):
                p = named[k[5:]]; gn = float(g['gradnorm:' + k[5:]])
                err = (p.grad.flatten()[:4096].double().cpu() - T(g[k]).double()).norm().item()
                print(f'{k[5:]:50s} relerr(head)={err/ (T(g[k]).double().norm().item()+1e-30):.3e} norm ratio={p.grad.double().norm().item()/gn:.6f}')
if 'stream' in sys.argv:
    sd, cfg = load_ckpt('442k')
    net = CleanUMamba(**cfg); net.load_state_dict(sd); net = net.to(dev).eval(); net.normalize_input = False
    x = T(load_golden('e2e_442k')['input']).to(dev)[0]
    with torch.no_grad():
        par = net(x.unsqueeze(0))[0][:, :16000]
        seq = torch.cat([net.feed(x), net.flush()], 1)
    d = (seq - par).abs()[0]
    print('shape', seq.shape, par.shape)
    for i in range(0, 16000, 1024):
        print(i, d[i:i+1024].max().item(), par[0, i:i+1024].abs().max().item())
