import sys, json, time, numpy as np, torch
sys.path.insert(0, ".")
from cleanumamba_amd.network import CleanUMamba
dev = torch.device("cuda")
with np.load("tests/golden/ckpt_442k.npz") as f:
    cfg = json.loads(bytes(f["__network_config__"]).decode())
    sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}
net = CleanUMamba(**cfg); net.load_state_dict(sd); net = net.to(dev).eval()
x = 0.1 * torch.randn(1, 1, 16000, device=dev)
with torch.no_grad():
    for _ in range(5): net(x)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(50): net(x)
    torch.cuda.synchronize()
print("C1 (442K model, B=1, 1 s @ 16 kHz) forward: %.3f ms" % ((time.time() - t0) / 50 * 1e3))
