import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import os
if os.environ.get("CUM_HOPPLAN_OLD"):        # same-box A/B against an older plan format: tools/_ab/hopplan_old.py + its CUM_LIB
    import importlib.util
    import cleanumamba_amd.hip  # noqa: F401
    spec = importlib.util.spec_from_file_location("cleanumamba_amd.network.hopplan", os.environ["CUM_HOPPLAN_OLD"])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cleanumamba_amd.network.hopplan"] = mod
    import cleanumamba_amd.network.convstack  # noqa: F401
    spec.loader.exec_module(mod)
from cleanumamba_amd.network import CleanUMamba
dev = torch.device("cuda")
with np.load("tests/golden/ckpt_pruned500k.npz") as f:
    cfg = json.loads(bytes(f["__network_config__"]).decode())
    sd = {k: torch.from_numpy(f[k].astype(np.float32)) for k in f.files if k != "__network_config__"}
net = CleanUMamba(**cfg); net.load_pruned_state_dict(sd); net = net.to(dev).eval()
S, n = 256, 160000
x = 0.05 * torch.randn(S, n, device=dev)
hop = net.total_stride
with torch.no_grad():
    net.feed_batch(x[:, :4 * hop + net.frame_length]); net.reset_stream(); torch.cuda.synchronize()
    ts = []
    t00 = time.perf_counter()
    for i in range(0, n, 16 * hop):
        t0 = time.perf_counter()
        net.feed_batch(x[:, i:i + 16 * hop])
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t0))
    t0 = time.perf_counter(); net.flush_batch(); torch.cuda.synchronize(); tf = time.perf_counter() - t0
print("calls", len(ts), "first", [round(1e3*a,3) for a in ts[0]], "second", [round(1e3*a,3) for a in ts[1]])
print("median host ms", 1e3*np.median([a for a,b in ts[2:]]), "median host+gpu ms", 1e3*np.median([b for a,b in ts[2:]]), "flush ms", 1e3*tf, "total", time.perf_counter()-t00)
import cProfile, pstats
net.reset_stream()
with torch.no_grad():
    net.feed_batch(x[:, :16*hop]); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    for i in range(16*hop, 16*hop*11, 16 * hop):
        net.feed_batch(x[:, i:i + 16 * hop])
    pr.disable()
    torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
# where flush spends its time
net.reset_stream()
with torch.no_grad():
    net.feed_batch(x[:, :40 * hop]); torch.cuda.synchronize()
    pr = cProfile.Profile(); pr.enable()
    net.flush_batch(); torch.cuda.synchronize()
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
net.reset_stream()
with torch.no_grad():
    net.feed_batch(x[:, :40 * hop]); torch.cuda.synchronize()
    t0 = time.perf_counter(); net.flush_batch(); torch.cuda.synchronize(); print("second flush ms", 1e3 * (time.perf_counter() - t0))
