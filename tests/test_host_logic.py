"""Host-side contract of the drop-in module (CPU): constructor, state-dict keys/shapes, checkpoint
loading, geometry, and the no-fallback rule."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import golden_json, load_ckpt, load_golden
from cleanumamba_amd.network import CleanUMamba, Net
from cleanumamba_amd.util.util import LinearWarmupCosineDecay

E8 = dict(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8, kernel_size=4,
          stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)


def test_net_factory_and_unknown_names():
    _, cfg = load_ckpt("442k")
    net = Net("CleanUMamba", cfg)
    assert isinstance(net, CleanUMamba)
    with pytest.raises(NotImplementedError):
        Net("CleanUNet", cfg)
    with pytest.raises(TypeError):
        Net("CleanUMamba", dict(cfg, encoder_norm=None))      # unknown keys raise, as in the reference
    with pytest.raises(NotImplementedError):
        CleanUMamba(**dict(cfg, LSTM=True))


@pytest.mark.parametrize("name", ["e8_synth", "e6_synth"])
def test_state_dict_keys_and_shapes_match_reference(name):
    meta = golden_json(load_golden("e2e_" + name)["meta"])
    with torch.device("meta"):
        net = CleanUMamba(**meta["cfg"])
    sd = net.state_dict()
    assert sorted(sd) == meta["keys"]
    assert [list(sd[k].shape) for k in meta["keys"]] == meta["shapes"]
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == (41376385 if "e8" in name else 27211393)


def test_442k_checkpoint_loads_strict_and_geometry():
    sd, cfg = load_ckpt("442k")
    net = CleanUMamba(**cfg)
    res = net.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert sum(p.numel() for p in net.parameters()) == 441601
    g = load_golden("e2e_442k")
    assert net.frame_length == int(g["frame_length"]) == 766
    assert net.total_stride == int(g["total_stride"]) == 256
    assert net.valid_length(16000) == int(g["valid_length"])
    assert net.pad_signal(torch.zeros(1, 1, 16000)).shape[-1] == 16126
    e8 = CleanUMamba.__new__(CleanUMamba)
    e8.encoder_n_layers, e8.kernel_size, e8.stride = 8, 4, 2
    assert e8.valid_length(160000) == 160254 and e8.valid_length(1) == 766
    e8.encoder_n_layers = 6
    assert e8.valid_length(160000) == 160062 and e8.valid_length(1) == 190


def test_pruned_checkpoint_loader_adopts_odd_shapes():
    sd, cfg = load_ckpt("pruned500k")
    net = CleanUMamba(**cfg)
    net.load_pruned_state_dict(sd)
    assert sum(p.numel() for p in net.parameters()) == 491655
    mix = [b.mixer for b in net.tsfm_Mamba_layers]
    assert [(m.d_inner, m.d_state, m.dt_rank) for m in mix] == [(8, 8, 32), (8, 8, 32), (48, 8, 32)]
    assert mix[0].d_model == 114 and net.norm_f.normalized_shape == (114,)
    assert net.encoder[1][0].in_channels == 53 and net.encoder[1][0].out_channels == 74
    assert type(mix[0]).__name__ == "Mamba"
    assert isinstance(mix[0].in_proj, nn.Linear) and isinstance(mix[0].conv1d, nn.Conv1d)
    assert mix[0].conv1d.groups == 8


def test_submodule_contract_used_by_pruning_code():
    _, cfg = load_ckpt("442k")
    net = CleanUMamba(**cfg)
    assert isinstance(net.encoder[0][0], nn.Conv1d) and isinstance(net.encoder[0][2], nn.Conv1d)
    assert isinstance(net.decoder[0][0], nn.Conv1d) and isinstance(net.decoder[0][2], nn.ConvTranspose1d)
    assert isinstance(net.decoder[0][3], nn.ReLU) and len(net.decoder[-1]) == 3
    assert isinstance(net.norm_f, nn.LayerNorm)
    blk = net.tsfm_Mamba_layers[0]
    assert isinstance(blk.norm, nn.LayerNorm) and blk.layer_idx == 0 and blk.mixer.layer_idx == 0
    for name in ("in_proj", "x_proj", "dt_proj", "out_proj"):
        assert isinstance(getattr(blk.mixer, name), nn.Linear)
    assert blk.mixer.dt_proj.bias._no_reinit
    # init semantics: A_log = log(1..N), D = 1
    assert torch.allclose(blk.mixer.A_log[0], torch.log(torch.arange(1, blk.mixer.d_state + 1).float()))
    assert torch.all(blk.mixer.D == 1)
    import copy, pickle
    copy.deepcopy(net)
    pickle.loads(pickle.dumps(net))


def test_no_cpu_fallback_for_the_hot_path():
    sd, cfg = load_ckpt("442k")
    net = CleanUMamba(**cfg).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 1, 2000))


def test_lr_schedule_matches_reference_formula():
    p = [nn.Parameter(torch.zeros(1))]
    opt = torch.optim.SGD(p, lr=1.0)
    sch = LinearWarmupCosineDecay(opt, lr_max=1e-4, n_iter=1000, iteration=0, divider=25, warmup_proportion=0.05)
    lrs = [sch.step() for _ in range(1000)]
    assert abs(lrs[0] - (4e-6 + (1e-4 - 4e-6) / 50)) < 1e-12
    assert abs(lrs[49] - 1e-4) < 1e-12
    assert abs(lrs[-1] - 4e-10) < 1e-12
    assert lrs[500] < lrs[100]


def test_integration_recipe_aliases_reference_import_paths():
    """The sys.modules recipe of INTEGRATION.md makes the reference's own import lines resolve to this package."""
    import subprocess
    import sys
    code = r'''
import sys, cleanumamba_amd.network, cleanumamba_amd.network.network, cleanumamba_amd.network.CleanUMamba
import cleanumamba_amd.network.layers, cleanumamba_amd.mamba_ssm as _m, cleanumamba_amd.causal_conv1d as _c
import cleanumamba_amd.mamba_ssm.models.mixer_seq_simple, cleanumamba_amd.mamba_ssm.utils.generation
import cleanumamba_amd.mamba_ssm.modules.mamba_simple, cleanumamba_amd.mamba_ssm.ops.selective_scan_interface
for name in ("network", "network.network", "network.CleanUMamba", "network.layers"):
    sys.modules["src." + name] = sys.modules["cleanumamba_amd." + name]
for name, mod in list(sys.modules.items()):
    if name.startswith("cleanumamba_amd.mamba_ssm"):
        sys.modules[name.replace("cleanumamba_amd.", "", 1)] = mod
sys.modules["causal_conv1d"] = _c
from src.network.network import Net
from src.network.layers import Activation
from mamba_ssm.models.mixer_seq_simple import create_block, _init_weights
from mamba_ssm.utils.generation import InferenceParams
from mamba_ssm.ops.selective_scan_interface import selective_scan_fn
from causal_conv1d import causal_conv1d_fn, causal_conv1d_update
net = Net("CleanUMamba", dict(channels_H=8, max_H=16, encoder_n_layers=3, tsfm_d_model=16, tsfm_d_inner=32, tsfm_n_head=2))
assert type(net.tsfm_Mamba_layers[0].mixer).__name__ == "Mamba"
print("ok")
'''
    from conftest import ROOT
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_graft_entry_build_is_consistent_with_the_header():
    """build() compiles (incrementally) and checks the library's ABI version against include/cleanumamba_hip.h."""
    import __graft_entry__ as g
    g.build()
    from cleanumamba_amd import hip
    assert set(hip.SIGNATURES) and hip.lib().cum_abi_version() >= 3


def test_bench_launcher_starts_ranks_and_fails_fast_without_a_gpu():
    """`python bench.py --gpus 2` with no rank environment must start its own two rank processes and relay a failure:
    here (no GPU) every rank stops at "bench.py needs a GPU", and the launcher must exit non-zero promptly, naming a
    rank, instead of asserting on WORLD_SIZE as it did before or hanging in a rendezvous."""
    import os
    import subprocess
    import sys
    import time
    if torch.cuda.is_available():
        pytest.skip("the failing-rank path needs a box without a GPU (tests/test_train_gpu.py covers the working one)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    t0 = time.time()
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--rank-timeout", "120"], env=env, capture_output=True, text=True, timeout=150)
    assert res.returncode == 1
    assert "exited with code" in res.stderr and "bench.py needs a GPU" in res.stderr, res.stderr[-1500:]
    assert "---- rank 1 ----" in res.stderr
    assert time.time() - t0 < 100


def test_bench_pins_ranks_to_the_numa_node_of_their_gpu(tmp_path):
    """bench.py --gpus N: every rank's CPU set comes from the NUMA node of its GPU (AMD display-class PCI devices in bus
    order), shared with the other ranks of that node; an even split when sysfs does not tell."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    sysr = tmp_path / "sys"
    nodes = [0, 0, 1, 1]
    for i, node in enumerate(nodes):
        d = sysr / "bus" / "pci" / "devices" / f"0000:{0x10 + i:02x}:00.0"
        d.mkdir(parents=True)
        (d / "vendor").write_text("0x1002\n")
        (d / "class").write_text("0x038000\n")
        (d / "numa_node").write_text(f"{node}\n")
    d = sysr / "bus" / "pci" / "devices" / "0000:01:00.0"          # an unrelated device in front of them
    d.mkdir(parents=True)
    (d / "vendor").write_text("0x8086\n")
    (d / "class").write_text("0x020000\n")
    (d / "numa_node").write_text("0\n")
    for node, cl in ((0, "0-7,16-23"), (1, "8-15,24-31")):
        nd = sysr / "devices" / "system" / "node" / f"node{node}"
        nd.mkdir(parents=True)
        (nd / "cpulist").write_text(cl + "\n")
    allowed = set(range(32))
    got = [bench._cpus_of_rank(r, 4, str(sysr), allowed) for r in range(4)]
    assert got[0] == [0, 1, 2, 3, 4, 5, 6, 7] and got[1] == [16, 17, 18, 19, 20, 21, 22, 23]
    assert got[2] == [8, 9, 10, 11, 12, 13, 14, 15] and got[3] == [24, 25, 26, 27, 28, 29, 30, 31]
    # a device mask re-numbers the GPUs: HIP device i is physical device visible[i]
    info = {}
    got = [bench._cpus_of_rank(r, 2, str(sysr), allowed, visible=[3, 0], info=info) for r in range(2)]
    assert got[0] == list(range(8, 16)) + list(range(24, 32)) and got[1] == list(range(0, 8)) + list(range(16, 24))
    assert info["numa_node"] == 0 and info["bdf"] == "0000:10:00.0" and "mask" in info["how"]
    # a mask that cannot be mapped (UUIDs, an index beyond the lower mask): the even split, and the line says so
    got = [bench._cpus_of_rank(r, 4, str(sysr), allowed, visible="unknown", info=info) for r in range(4)]
    assert got == [list(range(8 * r, 8 * r + 8)) for r in range(4)] and info["how"].startswith("even split")
    assert bench._visible_devices({}) is None
    assert bench._visible_devices({"HIP_VISIBLE_DEVICES": "2,3"}) == [2, 3]
    assert bench._visible_devices({"ROCR_VISIBLE_DEVICES": "1"}) == [1]
    assert bench._visible_devices({"HIP_VISIBLE_DEVICES": "GPU-abcdef"}) == "unknown"
    assert bench._visible_devices({"HIP_VISIBLE_DEVICES": "1,0", "ROCR_VISIBLE_DEVICES": "4,6"}) == [6, 4]
    assert bench._visible_devices({"HIP_VISIBLE_DEVICES": "2", "ROCR_VISIBLE_DEVICES": "4,6"}) == "unknown"
    # no sysfs: contiguous even split of what the process may use
    got = [bench._cpus_of_rank(r, 4, str(tmp_path / "nothing"), set(range(8))) for r in range(4)]
    assert got == [[0, 1], [2, 3], [4, 5], [6, 7]]


def test_flat_adam_checkpoint_rules_and_sink_registry():
    """FlatAdam.load_state_dict: a checkpoint written by a run WITHOUT loss scaling must not overwrite the live loss
    scale of an fp16 run (scale 1.0 would underflow the gradients for thousands of steps), one written with scaling
    restores it; the step count is restored; listeners (a captured train step) hear about the change.  FlatParams leaves
    no entry in the gradient-sink registry once it is collected."""
    import gc
    from cleanumamba_amd.training import flat_optim as fo
    torch.manual_seed(0)
    mk = lambda: nn.Sequential(nn.Linear(5, 7), nn.Linear(7, 3))
    a = mk()
    opt_plain = fo.FlatAdam(fo.FlatParams(a), lr=1e-3, loss_scaling=False)
    opt_plain.state_vec[fo.ST_STEP] = 41.0
    opt_plain.exp_avg.fill_(0.25)
    sd_plain = opt_plain.state_dict()
    b = mk()
    flat_b = fo.FlatParams(b)
    opt16 = fo.FlatAdam(flat_b, lr=1e-3, loss_scaling=True)
    heard = []
    opt16.on_hyper_change(lambda: heard.append(1))
    sd_plain["param_groups"][0]["betas"] = (0.8, 0.99)
    opt16.load_state_dict(sd_plain)
    assert float(opt16.state_vec[fo.ST_SCALE]) == 65536.0          # kept: the checkpoint did not scale its loss
    assert float(opt16.state_vec[fo.ST_STEP]) == 41.0 and float(opt16.exp_avg[0]) == 0.25
    assert opt16.param_groups[0]["betas"] == (0.8, 0.99) and heard == [1]
    opt16.state_vec[fo.ST_SCALE] = 8192.0
    opt16.state_vec[fo.ST_TRACKER] = 17.0
    c = mk()
    opt16c = fo.FlatAdam(fo.FlatParams(c), lr=1e-3, loss_scaling=True)
    opt16c.load_state_dict(opt16.state_dict())
    assert float(opt16c.state_vec[fo.ST_SCALE]) == 8192.0 and float(opt16c.state_vec[fo.ST_TRACKER]) == 17.0
    # re-pointed parameters are refused by name
    flat_b.require_intact()
    b[0].weight.data = b[0].weight.data.clone()
    with pytest.raises(RuntimeError, match="re-allocated"):
        flat_b.require_intact()
    # the registry forgets collected FlatParams
    ptrs = [p.data_ptr() for p in c.parameters()]
    assert all(ptr in fo._SINKS for ptr in ptrs)
    del opt16c, c
    gc.collect()
    assert not any(ptr in fo._SINKS for ptr in ptrs)


def test_pack_layouts_separate_into_row_and_column_tables():
    """network/convstack.py _separable: the host-side decision behind the index-free weight re-pack (cum_pack2d)."""
    from cleanumamba_amd.network.convstack import _separable, PACK_PAD
    # a [Cout, Cin, K] conv weight packed as [Cout_padded, K * Cin_padded] (tap-major), at offset 1000 of the flat buffer
    co, ci, k, cop, cip = 5, 3, 4, 8, 8
    g = torch.full((cop, k * cip), -1, dtype=torch.int64)
    for o in range(co):
        for t in range(k):
            for i in range(ci):
                g[o, t * cip + i] = 1000 + (o * ci + i) * k + t
    ro, cl, tr = _separable(g)
    assert not tr                                     # source stride k along columns (inside a tap), k * ci along rows
    full = ro.long()[:, None] + cl.long()[None, :]
    ok = (ro != PACK_PAD)[:, None] & (cl != PACK_PAD)[None, :]
    assert torch.equal(ok, g >= 0) and torch.equal(full[ok], g[ok])
    # its transpose: rows are now the near neighbours
    ro2, cl2, tr2 = _separable(g.t().contiguous())
    gt = g.t()
    full2 = ro2.long()[:, None] + cl2.long()[None, :]
    ok2 = (ro2 != PACK_PAD)[:, None] & (cl2 != PACK_PAD)[None, :]
    assert tr2 and torch.equal(ok2, gt >= 0) and torch.equal(full2[ok2], gt[ok2])
    # not separable: one element moved
    h = g.clone()
    h[1, 1] += 1
    assert _separable(h) is None
    # a hole that is not a whole row / column
    h = g.clone()
    h[0, 0] = -1
    assert _separable(h) is None
    assert _separable(torch.full((4, 8), -1, dtype=torch.int64)) is None
    # the contiguous-run promise of cum_pack2d (runs8): plain [N, K] weights yes, tap-major conv weights no
    from cleanumamba_amd.network.convstack import _runs8
    assert _runs8(ro, cl) == 0
    w = torch.full((6, 24), -1, dtype=torch.int64)
    w[:5, :16] = 2000 + torch.arange(5 * 16).view(5, 16)
    r3, c3, _ = _separable(w)
    assert _runs8(r3, c3) == 2                        # runs of 8 starting at multiples of 4 elements
    r4, c4, _ = _separable(torch.where(w >= 0, w + 1, w))
    assert _runs8(r4, c4) == 1                        # same runs, odd start: scalar loads
    w5 = w.clone()
    w5[:, 12:16] = -1                                 # a group of 8 that is half padding
    r5, c5, _ = _separable(w5)
    assert _runs8(r5, c5) == 0


def test_loss_terms_combination_matches_the_chain_of_scalar_ops():
    """util._Combine folds loss_fn's scalar chain (src/util/util.py:300-327: term * lambda, sums over the resolutions, / n,
    the total, the three logged components) into one matrix-vector product; values and gradients equal the chain's."""
    from cleanumamba_amd.util.util import _Combine
    torch.manual_seed(3)
    vals = [torch.rand((), dtype=torch.float32).requires_grad_(True) for _ in range(7)]      # ae, (sc, mag) x 3
    lam, stft_lam, sc_l, mag_l, n = 1.0, 1.0, 0.5, 0.5, 3
    cs_, cm_ = stft_lam * sc_l / n, stft_lam * mag_l / n
    sc_w, mag_w = [cs_, 0.0] * n, [0.0, cm_] * n
    rows = ((lam,) + tuple(a + b for a, b in zip(sc_w, mag_w)), (lam,) + (0.0,) * (2 * n), (0.0,) + tuple(sc_w), (0.0,) + tuple(mag_w))
    out = _Combine.apply(rows, *vals)
    ref_vals = [v.detach().clone().requires_grad_(True) for v in vals]
    sc = (ref_vals[1] + ref_vals[3] + ref_vals[5]) * sc_l / n
    mag = (ref_vals[2] + ref_vals[4] + ref_vals[6]) * mag_l / n
    loss = ref_vals[0] * lam + (sc + mag) * stft_lam
    assert torch.allclose(out[0], loss, rtol=1e-6) and torch.allclose(out[1], ref_vals[0] * lam, rtol=1e-6)
    assert torch.allclose(out[2], sc * stft_lam, rtol=1e-6) and torch.allclose(out[3], mag * stft_lam, rtol=1e-6)
    (3.0 * out[0]).backward()
    (3.0 * loss).backward()
    for a, b in zip(vals, ref_vals):
        assert torch.allclose(a.grad, b.grad, rtol=1e-6)


def test_xor_scatter_lane_algebra_of_the_backward_scan():
    """csrc/scan_bwd.hip keeps state k ^ h(lane) in register slot k so that one DPP add eliminates one register of the
    per-step dB / dC reduce-scatter; tools/xor_scatter_model.py restates the exchanges (quad_perm, row_half_mirror,
    row_ror:8, permlane32/16 swaps), the owner lane of every (array, state, step) total and the four staged B / C
    variants in numpy and checks them against plain sums."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("xor_scatter_model", os.path.join(root, "tools", "xor_scatter_model.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    for seed in range(3):
        assert m.check(seed)


def test_lazy_zero_grad_never_leaves_a_stale_gradient():
    """FlatParams.zero_grad() skips the memset under gradients that kernels WRITE through the sink; whatever else happens
    to such a view in the next cycle -- autograd accumulating into it, nothing touching it at all -- must see zeros."""
    import torch.nn as nn
    from cleanumamba_amd.training import flat_optim as fo
    m = nn.Sequential(nn.Linear(4, 4), nn.Linear(4, 2))
    f = fo.FlatParams(m)
    params = list(m.parameters())
    f.armed = True
    f.zero_grad()                                    # first cycle: everything zeroed, everything written by "kernels"
    for p in params:
        i, o = f.slot(p)
        f.grad[o:o + p.numel()] = 7.0
        f.wrote([i])
    f.settle()
    assert float(f.grad.sum()) == 7.0 * sum(p.numel() for p in params)
    f.zero_grad()                                    # lazy: nothing cleared yet, all four views stale
    assert len(f.stale) == 4 and float(f.grad.abs().sum()) > 0
    # parameter 0: sink-written again; parameter 1: autograd accumulates (the hook clears the view first);
    # parameters 2, 3: nothing reaches them -> zero at settle()
    i, o = f.slot(params[0])
    f.grad[o:o + params[0].numel()] = 1.0
    f.wrote([i])
    f.armed = False
    (params[1] * 2.0).sum().backward()
    f.settle()
    assert float(params[0].grad.sum()) == params[0].numel()
    assert torch.equal(params[1].grad, torch.full_like(params[1], 2.0))
    assert float(params[2].grad.abs().sum()) == 0 and float(params[3].grad.abs().sum()) == 0
    # the next zero_grad() may skip only what was sink-written in THIS cycle (parameter 0)
    f.zero_grad()
    assert f.stale == {f.index[id(params[0])]}
    assert float(params[1].grad.abs().sum()) == 0
    # under a gradient exchange (whole bucket slices are read before settle) nothing is left stale
    f.settle()
    f.exchange_reads_buffer = True
    f.armed = True
    i, o = f.slot(params[0])
    f.wrote([i])
    f.settle()
    f.zero_grad()
    assert not f.stale and float(f.grad.abs().sum()) == 0
