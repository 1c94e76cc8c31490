"""Host side of the one-launch streaming hop (cleanumamba_amd/network/hopplan.py, csrc/hop.hip): the op list compiled
from a model, on the CPU -- every shipped checkpoint gets a plan, every op stays inside its LDS / weight-blob / state
ranges, the MFMA fragment packing round-trips, the k-split mirrors the kernel's rule.  (The kernel itself: -m gpu,
tests/test_model_gpu.py::test_one_launch_hop_equals_per_layer_hop.)

Reference: CleanUMamba.feed / _denoise_frame, src/network/CleanUMamba.py:370-490."""
import numpy as np
import pytest
import torch

from conftest import load_ckpt

CKPTS = ["442k", "pruned500k", "e8_pruned200k", "e8_pruned1m", "e8_pruned2m", "e6_pruned200k", "e6_pruned500k",
         "e6_pruned1m", "e6_pruned2m"]


def _net(name):
    from cleanumamba_amd.network import CleanUMamba
    sd, cfg = load_ckpt(name)
    net = CleanUMamba(**cfg)
    (net.load_state_dict if name == "442k" else net.load_pruned_state_dict)(sd)
    return net.eval()


def test_fragment_packing_round_trips():
    from cleanumamba_amd.network import hopplan
    g = torch.Generator().manual_seed(0)
    for N, K in ((5, 7), (16, 16), (33, 100), (106, 56)):
        W = torch.randn(N, K, generator=g)
        frag = hopplan._frag(W)
        Np, Kp = hopplan._rup(N, 16), hopplan._rup(K, 16)
        assert frag.numel() == Np * Kp
        # lane l of tile (nt, kc) holds W[16 nt + l % 16][16 kc + 4 (l // 16) + j], j = 0..3 (csrc/hop.hip::hop_gemm)
        f = frag.view(Np // 16, Kp // 16, 64, 4)
        full = torch.zeros(Np, Kp)
        full[:N, :K] = W
        for nt, kc, lane in ((0, 0, 0), (Np // 16 - 1, Kp // 16 - 1, 63), (0, Kp // 16 - 1, 17), (Np // 16 - 1, 0, 46)):
            n, k = 16 * nt + lane % 16, 16 * kc + 4 * (lane // 16)
            assert torch.equal(f[nt, kc, lane], full[n, k:k + 4])


@pytest.mark.parametrize("name", CKPTS)
def test_every_shipped_checkpoint_gets_a_plan_inside_its_ranges(name):
    from cleanumamba_amd import hip
    from cleanumamba_amd.network import hopplan
    net = _net(name)
    assert hopplan.unsupported_reason(net) is None
    plan = hopplan.HopPlan(net)
    ints = plan.plan.numpy()
    assert ints.size == hip.lib().cum_stream_hop_plan_ints() and ints[0] == hopplan._MAGIC
    n_ops, frame_len, hop, lds_floats, phase_off, ops_lds = (int(v) for v in ints[1:7])
    assert n_ops == len(plan.ops) <= hopplan._MAX_OPS
    assert frame_len == net.frame_length and hop == net.total_stride
    assert plan.lds_bytes <= hip.lib().cum_stream_hop_max_lds_bytes() and 4 * lds_floats <= plan.lds_bytes
    assert ops_lds + n_ops * hopplan._OP_INTS <= lds_floats
    nw, ns = plan.weights.numel(), plan.state_stride
    E = net.encoder_n_layers
    kinds = [op[0] for op in plan.ops]
    assert kinds.count(hopplan._OP_GEMM) == (2 * E - 1) + 2 + 4 * len(net.tsfm_Mamba_layers) + 2 * E
    assert kinds.count(hopplan._OP_RING) == 0 and kinds.count(hopplan._OP_OVERLAP) == E
    # every encoder layer's 1x1 + GLU product also appends its rows to the layer's ring
    rings = [op[23] for op in plan.ops if op[0] == hopplan._OP_GEMM and op[23] >= 0]
    assert rings == [e["ring"] for e in plan.encs]
    for op in plan.ops:
        if op[0] == hopplan._OP_GEMM and op[23] >= 0:
            assert op[19] == 2 and op[16] == 2 and 16 * op[4] == plan.encs[rings.index(op[23])]["ld_out"]
    assert kinds.count(hopplan._OP_LN) == len(net.tsfm_Mamba_layers) + 1
    for op in plan.ops:
        if op[0] != hopplan._OP_GEMM:
            continue
        (_, w, x, scratch, ntg, kcn, xs, kpr, seg, M, cap, dst, bias, bias2, add, pitch, row_off, act, nlimit, nacc, ks,
         kcs, mt, ring) = op
        assert w % 4 == 0 and w + ntg * nacc * kcn * 256 <= nw           # a fragment chunk: 64 lanes x 4 floats = 1 KiB
        assert 0 <= x < ops_lds and x % 4 == 0 and xs % 4 == 0 and (xs == 0 or xs % 8 == 4 or xs % 8 == 0)
        assert kcn <= 4 * kpr and 0 <= dst < ops_lds and pitch % 4 == 0
        for b in (bias, bias2):
            assert b == -1 or (b % 4 == 0 and b + 16 * ntg <= nw)
        # the split (16-row tiles per item, k slices) is one the kernel has a body for and fits the scratch it names
        assert mt in (1, 2, 4) and 16 * mt <= max(16, hopplan._rup(M, 16))
        base = ntg * ((M + 16 * mt - 1) // (16 * mt))
        assert ks >= 1 and ks * kcs >= kcn and (ks - 1) * kcs < kcn
        if ks > 1:
            assert scratch + ks * base * nacc * mt * 256 <= scratch + cap <= ops_lds
            lo, hi = scratch, scratch + ks * base * nacc * mt * 256
            rows_read = M if xs else 1
            assert hi <= x or lo >= x + (rows_read - 1) * xs + 16 * kcn or seg, "scratch overlaps the operand it reads"
    # state block: rings, tails, Mamba states laid out without overlap behind the 4 header floats
    spans = [(0, 4)]
    spans += [(e["ring"], e["ring"] + 3 * e["n"] * e["ld_out"]) for e in plan.encs]
    spans += [(d["tail"], d["tail"] + 2 * d["cq"]) for d in plan.decs]
    for b in plan.blks:
        spans += [(b["conv_state"], b["conv_state"] + b["di"] * b["W"]), (b["ssm_state"], b["ssm_state"] + b["di"] * b["N"])]
    spans.sort()
    for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
        assert a1 <= b0
    assert spans[-1][1] <= ns
    # the deepest encoder window is one row; every decoder layer's skip names the ring of its mirror layer
    assert plan.encs[-1]["n"] == 1
    for j, d in enumerate(plan.decs[:-1]):
        assert d["skip_ring"] == plan.encs[E - 2 - j]["ring"] and d["skip_n"] == 2 * d["L"]
    assert plan.flops_per_hop > 0


@pytest.mark.parametrize("name", ["442k", "pruned500k", "e6_pruned2m"])
def test_stage_lists_compute_every_product(name):
    """The per-wave stage lists the kernel walks (hopplan._stages), emulated in numpy on the packed weight blob and a
    random LDS image: every product of the hop comes out as W . x of its op (k order, k split and epilogue placement
    included), every (tile, row) is produced exactly once, and the lists fit the kernel's tables."""
    from cleanumamba_amd.network import hopplan
    plan = hopplan.HopPlan(_net(name))
    ints = plan.plan.numpy()
    blob = plan.weights.numpy().astype(np.float64)
    lds_floats = int(ints[4])
    rng = np.random.default_rng(0)
    lds = rng.standard_normal(lds_floats + 4096)
    wtab0 = hopplan._HDR_INTS + hopplan._MAX_OPS * hopplan._OP_INTS
    stg0 = wtab0 + hopplan._MAX_OPS * hopplan._WAVES
    total = 0
    for k, op in enumerate(plan.ops):
        if op[0] != hopplan._OP_GEMM:
            assert not ints[wtab0 + hopplan._WAVES * k:wtab0 + hopplan._WAVES * (k + 1)].any()
            continue
        (_, w, x, scratch, ntg, kcn, xs, kpr, seg, M, cap, dst, bias, bias2, add, pitch, row_off, act, nlimit, nacc, ks,
         kcs, mt, _ring) = op
        # reference: out[a][n][m] = sum_k W_a[n][k] X[m][k] from the op's own description
        frag = blob[w:w + ntg * nacc * kcn * 256].reshape(ntg, nacc, kcn, 4, 16, 4)        # [tile][acc][kc][g][r][j]
        W = frag.transpose(1, 0, 4, 2, 3, 5).reshape(nacc, ntg * 16, kcn * 16)             # k = kc * 16 + 4 g + j
        X = np.empty((M, kcn * 16))
        for kc in range(kcn):
            sg = kc // kpr
            for m in range(M):
                a0 = x + m * xs + sg * seg + (kc - sg * kpr) * 16
                X[m, kc * 16:kc * 16 + 16] = lds[a0:a0 + 16]
        ref = np.einsum("ank,mk->anm", W, X)
        # the kernel's walk
        mgs = (M + 16 * mt - 1) // (16 * mt)
        got = np.full((nacc, ntg * 16, mgs * mt * 16), np.nan)
        part = {}
        for wave in range(hopplan._WAVES):
            wt = int(ints[wtab0 + hopplan._WAVES * k + wave])
            start, count = wt & 0xffff, wt >> 16
            assert count <= hopplan._MAX_WAVE_STAGES and count == len(plan.stages[k][wave])
            total += count
            acc = None
            for t in ints[stg0 + 4 * start:stg0 + 4 * (start + count)].reshape(count, 4):
                woff, xoff, meta, out = (int(v) for v in t)
                nb, first, last = meta & 7, meta >> 3 & 1, meta >> 4 & 1
                assert 1 <= nb <= 4 and woff % 4 == 0 and xoff % 4 == 0
                if first:
                    acc = np.zeros((nacc, 16, mt * 16))
                for d in range(nb):
                    for a in range(nacc):
                        ch = blob[woff + (a * kcn + d) * 256:woff + (a * kcn + d) * 256 + 256].reshape(4, 16, 4)   # [g][r][j]
                        Wc = ch.transpose(1, 0, 2).reshape(16, 16)
                        for r in range(mt * 16):
                            a0 = xoff + r * xs + d * 16
                            acc[a, :, r] += Wc @ lds[a0:a0 + 16]
                if last:
                    if ks == 1:
                        n0, m0 = out & 0xffff, out >> 16
                        assert np.isnan(got[:, n0:n0 + 16, m0:m0 + mt * 16]).all(), "a tile produced twice"
                        got[:, n0:n0 + 16, m0:m0 + mt * 16] = acc
                    else:
                        assert scratch <= out and out + nacc * mt * 256 <= scratch + cap and out not in part
                        part[out] = acc
                    acc = None
            assert acc is None
        if ks > 1:
            base, blk = ntg * mgs, nacc * mt * 256
            for b in range(base):
                bm, tg = b // ntg, b % ntg
                tot = sum(part.pop(scratch + (sl * base + b) * blk) for sl in range(ks))
                got[:, tg * 16:tg * 16 + 16, bm * mt * 16:(bm + 1) * mt * 16] = tot
            assert not part
        assert not np.isnan(got).any()
        np.testing.assert_allclose(got[:, :, :M], ref, rtol=1e-9, atol=1e-9)
    assert total == int(ints[7]) <= hopplan._MAX_STAGES


def test_models_outside_the_kernel_keep_the_per_layer_hop():
    from cleanumamba_amd.network import CleanUMamba, hopplan
    big = CleanUMamba(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8, kernel_size=4,
                      stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)
    assert "too large" in hopplan.unsupported_reason(big)          # 41 M parameters: streams batched as GEMM rows instead
    with pytest.raises(ValueError):
        hopplan.HopPlan(big)
    net = _net("442k")
    net.half()
    assert hopplan.unsupported_reason(net) == "parameters are not f32"
