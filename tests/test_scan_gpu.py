"""HIP selective scan / causal conv / step kernels vs the oracle and the golden vectors (GPU)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_l2
from oracle import mamba_ref as M

pytestmark = pytest.mark.gpu
T = torch.from_numpy
FWD_TOL, BWD_TOL = 1e-5, 1e-4        # per-op rel-L2, fp32


def _scan_inputs(g, dev, layout):
    """layout 'bdl': upstream (B, D, L) contiguous; 'bld': channel-contiguous memory viewed as (B, D, L)."""
    def lay(a):
        t = T(a).to(dev)
        if layout == "bld":
            t = t.transpose(1, 2).contiguous().transpose(1, 2)
        return t.requires_grad_(True)
    ins = {k: lay(g[k]) for k in ("u", "delta", "B", "C") + (("z",) if "z" in g else ())}
    for k in ("A", "D", "delta_bias"):
        ins[k] = T(g[k]).to(dev).requires_grad_(True) if k in g else None
    return ins


@pytest.mark.parametrize("layout", ["bld", "bdl"])
@pytest.mark.parametrize("idx", range(6))
def test_scan_fwd_bwd_vs_golden(cuda, idx, layout):
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    g = load_golden(f"scan_{idx}")
    ins = _scan_inputs(g, cuda, layout)
    y, last = selective_scan_fn(ins["u"], ins["delta"], ins["A"], ins["B"], ins["C"], ins["D"], z=ins.get("z"),
                                delta_bias=ins["delta_bias"], delta_softplus=True, return_last_state=True)
    assert y.shape == ins["u"].shape
    assert rel_l2(y, g["out64"]) < FWD_TOL
    assert rel_l2(last, g["last64"]) < FWD_TOL
    (y * T(g["dout"]).to(cuda)).sum().backward()
    for k in ("u", "delta", "A", "B", "C", "D", "z", "delta_bias"):
        if ins.get(k) is not None:
            assert rel_l2(ins[k].grad, g[f"d{k}64"]) < BWD_TOL, k


@pytest.mark.parametrize("shape", [(1, 1, 1, 1), (3, 70, 20, 47), (2, 130, 64, 16), (1, 64, 37, 17), (2, 9, 5, 100),
                                   # d_state <= 16: the wave-specialised backward (csrc/scan_bwd_small.hip) -- the d_state
                                   # values of the shipped checkpoints (8, 12, 13, 14, 16), lengths around the 8-step half
                                   (2, 130, 16, 41), (2, 70, 12, 23), (1, 200, 14, 8), (3, 64, 9, 7), (2, 136, 13, 64),
                                   (1, 8, 8, 129), (2, 48, 8, 9),
                                   # grids with more than 3 (2) waves per SIMD at d_state <= 8 (<= 16): the half-chunk form
                                   # of the one-wave forward (scan_fwd_small_kernel<.., SUB>), ragged last half
                                   (100, 2048, 8, 41), (70, 2000, 13, 23)])
def test_scan_odd_shapes_vs_oracle(cuda, shape):
    """ragged sizes: channels not a multiple of 64, d_state not a multiple of 8, L not a multiple of 16."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    bsz, dim, N, L = shape
    gen = torch.Generator().manual_seed(sum(shape))
    rn = lambda *s: torch.randn(*s, generator=gen)
    cpu = dict(u=rn(bsz, L, dim).transpose(1, 2), delta=0.5 * rn(bsz, L, dim).transpose(1, 2),
               A=-torch.exp(0.5 * rn(dim, N)), B=rn(bsz, L, N).transpose(1, 2), C=rn(bsz, L, N).transpose(1, 2),
               D=rn(dim), z=rn(bsz, L, dim).transpose(1, 2), delta_bias=0.5 * rn(dim))
    dout = rn(bsz, L, dim).transpose(1, 2)
    ref = {k: v.double().detach().requires_grad_(True) for k, v in cpu.items()}
    yr = M.selective_scan_ref(ref["u"], ref["delta"], ref["A"], ref["B"], ref["C"], ref["D"], z=ref["z"],
                              delta_bias=ref["delta_bias"], delta_softplus=True)
    (yr * dout.double()).sum().backward()
    dev = {k: v.to(cuda).requires_grad_(True) for k, v in cpu.items()}
    y = selective_scan_fn(dev["u"], dev["delta"], dev["A"], dev["B"], dev["C"], dev["D"], z=dev["z"],
                          delta_bias=dev["delta_bias"], delta_softplus=True)
    assert rel_l2(y, yr) < FWD_TOL
    (y * dout.to(cuda)).sum().backward()
    for k in cpu:
        assert rel_l2(dev[k].grad, ref[k].grad) < BWD_TOL, k


def _segments(shape):
    from cleanumamba_amd import hip
    return hip.lib().cum_scan_fwd_workspace_elems(*shape)


@pytest.mark.parametrize("io", [torch.float32, torch.float16])
@pytest.mark.parametrize("shape,opts", [
    ((1, 2048, 64, 624), "plain"),          # file denoising at batch 1, E8 bottleneck (src/examples/denoise.py)
    ((16, 128, 16, 624), "plain"),          # the 442K model at the training batch
    ((8, 48, 8, 500), "plain"),             # a pruned-E8 block (d_inner 48, d_state 8)
    ((2, 70, 13, 257), "no_z"),             # ragged everything: channels, states, last chunk, last segment
    ((1, 100, 37, 200), "no_softplus_no_bias"),
    ((3, 8, 8, 129), "no_D"),
    ((2, 136, 12, 97), "plain")])
def test_time_parallel_scan_vs_oracle_and_sequential(cuda, shape, opts, io):
    """SURVEY 8 row a9' (north_star: "wavefront shuffle / prefix-sum for the scan recurrence"): the segmented forward
    (csrc/scan_seg.hip) on grids the sequential kernels would leave mostly idle -- against the f64 oracle (output,
    last_state, and EVERY gradient through the backward kernels, which consume the checkpoints the segmented pass wrote:
    the TIME-PARALLEL backward of csrc/scan_bwd_small.hip where its plan segments the shape (d_state <= 16, long enough),
    else the sequential kernels) and against the sequential forward + backward on the same inputs."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.mamba_ssm.ops import selective_scan_interface as ssi
    bsz, dim, N, L = shape
    assert _segments(shape) > 0, "the plan does not segment this shape"
    tp_bwd = hip.lib().cum_scan_bwd_tp_workspace_elems(*shape) > 0
    assert tp_bwd == (N <= 16 and (L + 15) // 16 >= 6), "time-parallel backward: every small-state shape long enough"
    gen = torch.Generator().manual_seed(sum(shape))
    rn = lambda *s: torch.randn(*s, generator=gen)
    cpu = dict(u=rn(bsz, L, dim).transpose(1, 2), delta=0.5 * rn(bsz, L, dim).transpose(1, 2),
               A=-torch.exp(0.5 * rn(dim, N)), B=rn(bsz, L, N).transpose(1, 2), C=rn(bsz, L, N).transpose(1, 2),
               D=rn(dim), z=rn(bsz, L, dim).transpose(1, 2), delta_bias=0.5 * rn(dim))
    if opts == "no_softplus_no_bias":
        cpu["delta"] = cpu["delta"].abs()
    drop = {"plain": (), "no_z": ("z",), "no_softplus_no_bias": ("delta_bias",), "no_D": ("D",)}[opts]
    softplus = opts != "no_softplus_no_bias"
    for k in ("u", "delta", "z"):
        cpu[k] = cpu[k].to(io).float()                      # the values the kernels read
    dout = rn(bsz, L, dim).transpose(1, 2)

    def run(t, fn, cast=None):
        c = (lambda v: v.to(cast)) if cast is not None else (lambda v: v)
        y, last = fn(c(t["u"]), c(t["delta"]), t["A"], t["B"], t["C"], t.get("D"), z=None if "z" not in t else c(t["z"]),
                     delta_bias=t.get("delta_bias"), delta_softplus=softplus, return_last_state=True)
        return y, last
    ref = {k: v.double().detach().requires_grad_(True) for k, v in cpu.items() if k not in drop}
    yr, lastr = run(ref, M.selective_scan_ref)
    (yr * dout.double()).sum().backward()
    out = {}
    for mode in (True, False):
        ssi.TIME_PARALLEL = mode
        try:
            dev = {k: v.to(cuda).requires_grad_(True) for k, v in cpu.items() if k not in drop}
            y, last = run(dev, ssi.selective_scan_fn, cast=io)
            (y.float() * dout.to(cuda)).sum().backward()
        finally:
            ssi.TIME_PARALLEL = True
        out[mode] = (y.float(), last, {k: dev[k].grad.float() for k in dev})
    y, last, grads = out[True]
    ftol, btol = (FWD_TOL, BWD_TOL) if io == torch.float32 else (6e-4, 2e-3)      # f16: output / gradient rounding
    assert rel_l2(y, yr) < ftol
    assert rel_l2(last, lastr) < FWD_TOL
    for k in grads:
        assert rel_l2(grads[k], ref[k].grad) < btol, k
    # against the sequential kernels: only the rounding of the segment decay exp2(A' sum delta') differs
    ys, lasts, gs = out[False]
    assert rel_l2(y, ys) < (3e-6 if io == torch.float32 else 6e-4) and rel_l2(last, lasts) < 3e-6
    for k in grads:
        assert rel_l2(grads[k], gs[k]) < (3e-5 if io == torch.float32 else 2e-3), k
    # bit-reproducible
    ssi.TIME_PARALLEL = True
    dev = {k: v.to(cuda) for k, v in cpu.items() if k not in drop}
    y2, last2 = run(dev, ssi.selective_scan_fn, cast=io)
    assert torch.equal(y2.float(), y) and torch.equal(last2, last)


def test_time_parallel_backward_is_bit_reproducible_and_takes_16_bit_io(cuda):
    """The time-parallel backward on the two bench shapes (442K model at B = 16; pruned-E8 block at B = 256 streams-worth
    of clips is past the plan's limit, so 64 clips): two runs are bit-identical (slabs + fixed-order finalize, no atomics), f16
    I/O agrees with f32 I/O."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.mamba_ssm.ops import selective_scan_interface as ssi
    for shape in ((16, 128, 16, 624), (64, 48, 8, 1875)):
        bsz, dim, N, L = shape
        assert hip.lib().cum_scan_bwd_tp_workspace_elems(*shape) > 0
        g = torch.Generator(device=cuda).manual_seed(L)
        rn = lambda *s: torch.randn(*s, generator=g, device=cuda)
        base = dict(u=rn(bsz, L, dim).transpose(1, 2), delta=0.5 * rn(bsz, L, dim).transpose(1, 2),
                    A=-torch.exp(0.5 * rn(dim, N)), B=rn(bsz, L, N).transpose(1, 2), C=rn(bsz, L, N).transpose(1, 2),
                    D=rn(dim), z=rn(bsz, L, dim).transpose(1, 2), delta_bias=0.5 * rn(dim))
        dout = rn(bsz, L, dim).transpose(1, 2)
        runs = []
        for io in (torch.float32, torch.float32, torch.float16):
            t = {k: (v.to(io) if k in ("u", "delta", "z") else v).detach().requires_grad_(True) for k, v in base.items()}
            y = ssi.selective_scan_fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t["D"], z=t["z"],
                                      delta_bias=t["delta_bias"], delta_softplus=True)
            (y.float() * dout).sum().backward()
            runs.append({k: v.grad.float() for k, v in t.items()})
        for k in runs[0]:
            assert torch.equal(runs[0][k], runs[1][k]), k
            assert rel_l2(runs[2][k], runs[0][k]) < 3e-3, k


def test_time_parallel_plan():
    """The plan: segmented when the sequential grid brings fewer than two waves per SIMD and the sequence has at least
    three segments of two 16-step chunks; never for the training shapes that fill the chip."""
    assert _segments((16, 2048, 64, 624)) == 0 and _segments((32, 2048, 64, 2499)) == 0      # E8 / E6 training
    assert _segments((1, 2048, 64, 624)) > 0 and _segments((16, 128, 16, 624)) > 0 and _segments((256, 48, 8, 1875)) > 0
    from cleanumamba_amd import hip
    tp = hip.lib().cum_scan_bwd_tp_workspace_elems
    assert tp(16, 2048, 64, 624) == 0 and tp(1, 2048, 64, 624) == 0          # d_state > 16: sequential backward
    assert tp(16, 2048, 8, 2499) == 0                                        # 512 workgroups: the chip is busy
    assert tp(16, 128, 16, 624) > 0 and tp(128, 48, 8, 1875) > 0 and tp(2, 70, 13, 257) > 0
    assert tp(256, 48, 8, 1875) == 0                                         # 256 workgroups: 0.9-1.1 x measured, not taken
    assert _segments((1, 128, 16, 61)) == 0                                                      # 4 chunks: too short
    assert _segments((0, 8, 8, 100)) == 0 and _segments((1, 8, 8, 0)) == 0


@pytest.mark.parametrize("N", [8, 16])
@pytest.mark.parametrize("opts", ["plain", "no_z", "no_softplus_no_bias", "no_D"])
def test_scan_small_state_optional_arguments(cuda, N, opts):
    """d_state <= 16 backward (producer / consumer waves): the optional operands of selective_scan_fn -- no gate, no
    softplus, no bias, no skip term -- against the f64 oracle."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    bsz, dim, L = 2, 96, 37
    gen = torch.Generator().manual_seed(N + len(opts))
    rn = lambda *s: torch.randn(*s, generator=gen)
    cpu = dict(u=rn(bsz, L, dim).transpose(1, 2), delta=(0.5 * rn(bsz, L, dim)).abs().transpose(1, 2),
               A=-torch.exp(0.5 * rn(dim, N)), B=rn(bsz, L, N).transpose(1, 2), C=rn(bsz, L, N).transpose(1, 2),
               D=rn(dim), z=rn(bsz, L, dim).transpose(1, 2), delta_bias=0.5 * rn(dim))
    drop = {"plain": (), "no_z": ("z",), "no_softplus_no_bias": ("delta_bias",), "no_D": ("D",)}[opts]
    softplus = opts != "no_softplus_no_bias"
    dout = rn(bsz, L, dim).transpose(1, 2)

    def run(t, fn):
        kw = dict(z=t.get("z"), delta_bias=t.get("delta_bias"), delta_softplus=softplus)
        y = fn(t["u"], t["delta"], t["A"], t["B"], t["C"], t.get("D"), **kw)
        (y * dout.to(y.device).to(y.dtype)).sum().backward()
        return y
    ref = {k: v.double().detach().requires_grad_(True) for k, v in cpu.items() if k not in drop}
    dev = {k: v.to(cuda).requires_grad_(True) for k, v in cpu.items() if k not in drop}
    yr, y = run(ref, M.selective_scan_ref), run(dev, selective_scan_fn)
    assert rel_l2(y, yr) < FWD_TOL
    for k in ref:
        assert rel_l2(dev[k].grad, ref[k].grad) < BWD_TOL, k


@pytest.mark.parametrize("io", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 128, 64, 50), (1, 70, 64, 33), (2, 64, 32, 40)])
def test_scan_backward_with_the_forwards_y_equals_the_rebuilding_one(cuda, shape, io, monkeypatch):
    """d_state > 16, gated: the forward keeps y before the gate (cum_scan_fwd_keeps_y, y_pre) and the backward reads it
    instead of rebuilding sum_n C x_t + D u (csrc/scan_bwd.hip YIN).  Every gradient against the f64 oracle and against
    the backward that rebuilds y (keeps_y forced off); the kept y against the oracle's; really taken."""
    from cleanumamba_amd.mamba_ssm.ops import selective_scan_interface as ssi
    bsz, dim, N, L = shape
    g = torch.Generator().manual_seed(5)
    rn = lambda *sh: torch.randn(*sh, generator=g)
    cpu = dict(u=rn(bsz, L, dim).transpose(1, 2), delta=0.5 * rn(bsz, L, dim).transpose(1, 2), A=-torch.exp(0.5 * rn(dim, N)),
               B=rn(bsz, L, N).transpose(1, 2), C=rn(bsz, L, N).transpose(1, 2), D=rn(dim), z=rn(bsz, L, dim).transpose(1, 2),
               delta_bias=0.3 * rn(dim))
    dout = rn(bsz, L, dim).transpose(1, 2)
    lowp = ("u", "delta", "z")
    rnd = lambda k, v: (v.to(io).double() if k in lowp else v.double())
    ref = {k: rnd(k, v).detach().requires_grad_(True) for k, v in cpu.items()}
    yr = M.selective_scan_ref(ref["u"], ref["delta"], ref["A"], ref["B"], ref["C"], ref["D"], z=ref["z"],
                              delta_bias=ref["delta_bias"], delta_softplus=True)
    (yr * dout.to(io).double()).sum().backward()
    kept = []
    real = ssi.scan_forward

    def spy(*a, **k):
        kept.append(k.get("y_pre"))
        return real(*a, **k)
    monkeypatch.setattr(ssi, "scan_forward", spy)

    def run(keep):
        if not keep:
            monkeypatch.setattr(ssi, "keeps_y", lambda *a, **k: False)
        dev = {k: (v.to(io) if k in lowp else v).to(cuda).detach().requires_grad_(True) for k, v in cpu.items()}
        y = ssi.selective_scan_fn(dev["u"], dev["delta"], dev["A"], dev["B"], dev["C"], dev["D"], z=dev["z"],
                                  delta_bias=dev["delta_bias"], delta_softplus=True)
        (y.float() * dout.to(io).float().to(cuda)).sum().backward()
        return y.detach().float(), {k: dev[k].grad.detach().float().cpu() for k in dev}
    y1, g1 = run(True)
    assert kept[-1] is not None and kept[-1].dtype == io                      # the forward was asked for y
    yk = kept[-1].float().cpu()
    ypre_ref = M.selective_scan_ref(ref["u"], ref["delta"], ref["A"], ref["B"], ref["C"], ref["D"], z=None,
                                    delta_bias=ref["delta_bias"], delta_softplus=True).detach()
    low = io != torch.float32
    assert rel_l2(yk, ypre_ref) < (6e-3 if low else FWD_TOL)
    y0, g0 = run(False)
    assert kept[-1] is None
    assert torch.equal(y1, y0)
    for k in g1:
        assert rel_l2(g1[k], g0[k]) < (6e-3 if low else 1e-5), k
        assert rel_l2(g1[k], ref[k].grad) < (2e-2 if low else BWD_TOL), k


def test_scan_softplus_threshold_and_empty(cuda):
    """delta + bias > 20 takes the identity branch; zero-length and zero-batch inputs are accepted."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    gen = torch.Generator().manual_seed(3)
    u, delta = torch.randn(1, 8, 12, generator=gen), torch.full((1, 8, 12), 25.0)
    delta[:, :, ::2] = -30.0
    A = -torch.rand(8, 8, generator=gen) * 0.01
    Bm, Cm = torch.randn(1, 8, 12, generator=gen), torch.randn(1, 8, 12, generator=gen)
    yr = M.selective_scan_ref(u.double(), delta.double(), A.double(), Bm.double(), Cm.double(), delta_softplus=True)
    y = selective_scan_fn(u.to(cuda), delta.to(cuda), A.to(cuda), Bm.to(cuda), Cm.to(cuda), delta_softplus=True)
    assert rel_l2(y, yr) < FWD_TOL
    y0 = selective_scan_fn(u[:, :, :0].to(cuda), delta[:, :, :0].to(cuda), A.to(cuda), Bm[:, :, :0].to(cuda),
                           Cm[:, :, :0].to(cuda))
    assert y0.shape == (1, 8, 0)
    with pytest.raises(RuntimeError):
        selective_scan_fn(u.to(cuda), delta.to(cuda), torch.zeros(8, 65, device=cuda), Bm.to(cuda), Cm.to(cuda))


def test_scan_linearity_and_determinism_at_e8_size(cuda):
    """Size-independent properties at the BASELINE E8 bottleneck shape (B=16, D=2048, N=64, L=624):
    out is linear in C; two runs are bit-identical (no float atomics); chunk carry (L > 16) is exact."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    g = torch.Generator(device="cuda").manual_seed(0)
    bsz, dim, N, L = 16, 2048, 64, 624
    rn = lambda *s: torch.randn(*s, generator=g, device=cuda)
    u, delta, z = (rn(bsz, L, dim).transpose(1, 2) for _ in range(3))
    delta = 0.3 * delta
    A = -torch.exp(torch.log(torch.arange(1, N + 1, device=cuda).float())[None].repeat(dim, 1))
    xd = rn(bsz, L, 2 * N)
    Bm, Cm = xd[..., :N].transpose(1, 2), xd[..., N:].transpose(1, 2)
    bias = 0.3 * rn(dim)
    f = lambda Cx, zz: selective_scan_fn(u, delta, A, Bm, Cx, None, z=zz, delta_bias=bias, delta_softplus=True)
    y1, y2 = f(Cm, None), f(2.5 * Cm, None)
    assert rel_l2(y2, 2.5 * y1) < 1e-6
    ins = [t.detach().clone().requires_grad_(True) for t in (u, delta, A, xd)]
    def run():
        for t in ins:
            t.grad = None
        y = selective_scan_fn(ins[0], ins[1], ins[2], ins[3][..., :N].transpose(1, 2),
                              ins[3][..., N:].transpose(1, 2), None, z=z, delta_bias=bias, delta_softplus=True)
        y.square().mean().backward()
        return [y.detach().clone()] + [t.grad.clone() for t in ins]
    a, b = run(), run()
    for ta, tb in zip(a, b):
        assert torch.equal(ta, tb), "scan is not bit-reproducible"
    # a prefix of the sequence gives the prefix of the output (causality across chunk boundaries)
    yp = selective_scan_fn(u[..., :333], delta[..., :333], A, Bm[..., :333], Cm[..., :333], None, z=z[..., :333],
                           delta_bias=bias, delta_softplus=True)
    assert torch.equal(yp, a[0][..., :333].contiguous().view_as(yp)) or rel_l2(yp, a[0][..., :333]) < 1e-6
    # spot-check 3 channels of one batch element against the fp64 oracle
    sel = [0, 777, 2047]
    yr = M.selective_scan_ref(u[3:4, sel].double().cpu(), delta[3:4, sel].double().cpu(), A[sel].double().cpu(),
                              Bm[3:4].double().cpu(), Cm[3:4].double().cpu(), None, z=z[3:4, sel].double().cpu(),
                              delta_bias=bias[sel].double().cpu(), delta_softplus=True)
    assert rel_l2(a[0][3:4, sel], yr) < FWD_TOL


def test_scan_properties_at_e6_size_bf16_io(cuda):
    """BASELINE config 2 bottleneck shape (E6: B=32, D=2048, N=64, L=2499), bf16 I/O as under autocast: the
    mid-chunk checkpoints and the LDS-parked decay factors of the backward at a length that is not a multiple of the
    16-step chunk.  Size-independent checks: bit-reproducible forward and gradients, causality (a prefix of the input
    gives the prefix of the output), and three channels of one clip against the float64 oracle, forward and dA."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    g = torch.Generator(device="cuda").manual_seed(1)
    bsz, dim, N, L = 32, 2048, 64, 2499
    rn = lambda *s: torch.randn(*s, generator=g, device=cuda)
    u, z = (rn(bsz, L, dim).bfloat16().transpose(1, 2) for _ in range(2))
    delta = (0.3 * rn(bsz, L, dim)).bfloat16().transpose(1, 2)
    A = -torch.exp(torch.log(torch.arange(1, N + 1, device=cuda).float())[None].repeat(dim, 1))
    xd = rn(bsz, L, 2 * N)
    Bm, Cm = xd[..., :N].transpose(1, 2), xd[..., N:].transpose(1, 2)
    bias = 0.3 * rn(dim)
    dout = rn(bsz, L, dim).bfloat16().transpose(1, 2)

    def run():
        leaves = [t.detach().clone().requires_grad_(True) for t in (u, delta, A)]
        y = selective_scan_fn(leaves[0], leaves[1], leaves[2], Bm, Cm, None, z=z, delta_bias=bias, delta_softplus=True)
        y.backward(dout)
        return [y.detach()] + [t.grad for t in leaves]
    a, b = run(), run()
    for ta, tb in zip(a, b):
        assert torch.equal(ta, tb), "scan is not bit-reproducible at the E6 size"
    with torch.no_grad():
        yp = selective_scan_fn(u[..., :1000], delta[..., :1000], A, Bm[..., :1000], Cm[..., :1000], None,
                               z=z[..., :1000], delta_bias=bias, delta_softplus=True)
    assert torch.equal(yp, a[0][..., :1000])
    sel, clip = [5, 1024, 2040], 7
    ur, dr = u[clip:clip + 1, sel].double().cpu(), delta[clip:clip + 1, sel].double().cpu()
    Ar = A[sel].double().cpu().requires_grad_(True)
    yr = M.selective_scan_ref(ur, dr, Ar, Bm[clip:clip + 1].double().cpu(), Cm[clip:clip + 1].double().cpu(), None,
                              z=z[clip:clip + 1, sel].double().cpu(), delta_bias=bias[sel].double().cpu(),
                              delta_softplus=True)
    assert rel_l2(a[0][clip:clip + 1, sel].float(), yr) < 4e-3          # bf16 output rounding
    # dA of those channels from this clip alone: run the kernel on the one-clip, three-channel problem
    leaves = [t.detach().clone().requires_grad_(True) for t in (u[clip:clip + 1, sel].contiguous(),
                                                                 delta[clip:clip + 1, sel].contiguous(), A[sel].clone())]
    y1 = selective_scan_fn(leaves[0].transpose(1, 2).contiguous().transpose(1, 2), leaves[1].transpose(1, 2).contiguous().transpose(1, 2),
                           leaves[2], Bm[clip:clip + 1], Cm[clip:clip + 1], None,
                           z=z[clip:clip + 1, sel].transpose(1, 2).contiguous().transpose(1, 2), delta_bias=bias[sel],
                           delta_softplus=True)
    y1.backward(dout[clip:clip + 1, sel].transpose(1, 2).contiguous().transpose(1, 2))
    yr.backward(dout[clip:clip + 1, sel].double().cpu())
    assert rel_l2(leaves[2].grad, Ar.grad) < 4e-3


@pytest.mark.parametrize("idx", range(4))
@pytest.mark.parametrize("layout", ["bld", "bdl"])
def test_dwconv_fwd_bwd_vs_golden(cuda, idx, layout):
    from cleanumamba_amd.causal_conv1d import causal_conv1d_fn
    g = load_golden(f"dwconv_{idx}")
    x = T(g["x"]).to(cuda)
    if layout == "bld":
        x = x.transpose(1, 2).contiguous().transpose(1, 2)
    x.requires_grad_(True)
    w, b = T(g["w"]).to(cuda).requires_grad_(True), T(g["b"]).to(cuda).requires_grad_(True)
    y = causal_conv1d_fn(x, w, b, "silu")
    assert rel_l2(y, g["y64"]) < FWD_TOL
    (y * T(g["dout"]).to(cuda)).sum().backward()
    assert rel_l2(x.grad, g["dx64"]) < BWD_TOL
    assert rel_l2(w.grad, g["dw64"]) < BWD_TOL
    assert rel_l2(b.grad, g["db64"]) < BWD_TOL
    # no activation, no bias, width 2
    y2 = causal_conv1d_fn(x.detach(), w.detach()[:, :2].contiguous(), None, None)
    assert rel_l2(y2, M.causal_conv1d_ref(T(g["x"]).double(), T(g["w"]).double()[:, :2])) < FWD_TOL


def test_streaming_step_kernels_vs_golden(cuda):
    from cleanumamba_amd.causal_conv1d import causal_conv1d_update
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_state_update
    g = {k: T(v).to(cuda) for k, v in load_golden("step").items()}
    cs = torch.zeros(3, 48, 4, device=cuda)
    ss = torch.zeros(3, 48, 13, device=cuda)
    for s in range(g["xs"].shape[0]):
        xc = causal_conv1d_update(g["xs"][s], cs, g["w"], g["conv_bias"], "silu")
        assert rel_l2(xc, g["xconv64"][s]) < FWD_TOL
        y = selective_state_update(ss, xc, g["dts"][s], g["A"], g["Bs"][s], g["Cs"][s], g["D"], z=g["zs"][s],
                                   dt_bias=g["dt_bias"], dt_softplus=True)
        assert rel_l2(y, g["y64"][s]) < 2e-5
    assert rel_l2(ss, g["ssm_state64"]) < 2e-5
    assert rel_l2(cs, g["conv_state64"]) < 1e-6


@pytest.mark.parametrize("io,tol", [(torch.bfloat16, 4e-3), (torch.float16, 5e-4)])
@pytest.mark.parametrize("shape", [(2, 192, 64, 150), (1, 70, 20, 33), (2, 53, 8, 40)])   # 53: odd dim, one channel per lane
def test_scan_and_dwconv_16bit_io_vs_oracle(cuda, shape, io, tol):
    """bf16 / f16 element type for u / delta / z / out (what autocast hands over; cum_scan_shape.io_dtype): against
    the f64 oracle on the SAME rounded inputs the only differences are the kernels' f32 arithmetic and the final
    rounding of each output (2^-9 relative per element in bf16, 2^-12 in f16), hence rel-L2 <= 4e-3 / 5e-4."""
    from cleanumamba_amd.causal_conv1d import causal_conv1d_fn
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import selective_scan_fn
    bsz, dim, N, L = shape
    g = torch.Generator().manual_seed(sum(shape))
    rn = lambda *s: torch.randn(*s, generator=g)
    bf = lambda t: t.to(io).float()
    u, delta, z = bf(rn(bsz, L, dim)), bf(0.5 * rn(bsz, L, dim)), bf(rn(bsz, L, dim))
    A = -torch.exp(rn(dim, N) * 0.5)
    Bm, Cm, D, bias = rn(bsz, L, N), rn(bsz, L, N), rn(dim), 0.3 * rn(dim)
    dout = bf(rn(bsz, L, dim))

    def run(dev, dt_io, dt_ref):
        leaves = [t.to(dev).to(dt_io).requires_grad_(True) for t in (u, delta, z)]
        rest = [t.to(dev).to(dt_ref).requires_grad_(True) for t in (A, Bm, Cm, D, bias)]
        uu, dd, zz = (t.transpose(1, 2) for t in leaves)
        args = (uu, dd, rest[0], rest[1].transpose(1, 2), rest[2].transpose(1, 2), rest[3])
        if dev.type == "cuda":
            out = selective_scan_fn(*args, z=zz, delta_bias=rest[4], delta_softplus=True)
        else:
            out = M.selective_scan_ref(*args, z=zz, delta_bias=rest[4], delta_softplus=True)
        out.backward(dout.to(dev).to(out.dtype).transpose(1, 2))
        return [out] + [t.grad for t in leaves + rest]

    got = run(cuda, io, torch.float32)
    ref = run(torch.device("cpu"), torch.float64, torch.float64)
    assert got[0].dtype == io and all(t.dtype == io for t in got[1:4])
    for name, a, b in zip(("out", "du", "ddelta", "dz", "dA", "dB", "dC", "dD", "dbias"), got, ref):
        assert rel_l2(a.float(), b) < tol, name

    # depthwise conv + SiLU, same comparison
    w, cb = rn(dim, 4), rn(dim)
    xg = u.to(cuda).to(io).requires_grad_(True)
    wg, bg = w.to(cuda).requires_grad_(True), cb.to(cuda).requires_grad_(True)
    y = causal_conv1d_fn(xg.transpose(1, 2), wg, bg, "silu")
    y.backward(dout.to(cuda).to(io).transpose(1, 2))
    xr = u.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), cb.double().requires_grad_(True)
    yr = M.causal_conv1d_ref(xr.transpose(1, 2), wr, br, "silu")
    yr.backward(dout.double().transpose(1, 2))
    assert y.dtype == io and xg.grad.dtype == io and wg.grad.dtype == torch.float32
    for name, a, b in (("y", y, yr), ("dx", xg.grad, xr.grad), ("dw", wg.grad, wr.grad), ("db", bg.grad, br.grad)):
        assert rel_l2(a.float(), b) < tol, name


@pytest.mark.parametrize("dim,dt", [(512, torch.float32), (512, torch.bfloat16), (72, torch.float32), (2048, torch.bfloat16),
                                    (512, torch.float16), (2048, torch.float16)])
def test_add_layernorm_vs_torch(cuda, dim, dt):
    """Residual add + LayerNorm (csrc/layernorm.hip) against the separate torch ops the reference executes
    (Block.forward, fused_add_norm=False), evaluated in float64: forward 1e-6 (f32 output) / bf16 rounding,
    gradients 1e-5 (f32) / 5e-3 (bf16 hidden gradient)."""
    from cleanumamba_amd.mamba_ssm.ops.layernorm import AddLayerNormFn
    g = torch.Generator().manual_seed(dim)
    bsz, L = 3, 37
    wide = torch.randn(bsz, L + 2, dim + 8, generator=g)          # strided (batch, time) view like a row buffer
    h0 = wide[:, :L, :dim].to(dt)
    r0, w0, b0 = torch.randn(bsz, L, dim, generator=g), 1 + 0.1 * torch.randn(dim, generator=g), 0.1 * torch.randn(dim, generator=g)
    gy, gr = torch.randn(bsz, L, dim, generator=g).to(dt).float(), torch.randn(bsz, L, dim, generator=g)

    base = wide.to(dt).to(cuda).requires_grad_(True)
    h = base[:, :L, :dim]
    r, w, b = (t.to(cuda).requires_grad_(True) for t in (r0, w0, b0))
    y, res = AddLayerNormFn.apply(h, r, w, b, 1e-5, dt)
    assert y.dtype == dt and res.dtype == torch.float32
    ((y.float() * gy.to(cuda)).sum() + (res * gr.to(cuda)).sum()).backward()

    hd = h0.double().requires_grad_(True)
    rd, wd, bd = (t.double().requires_grad_(True) for t in (r0, w0, b0))
    res_d = hd + rd
    y_d = torch.nn.functional.layer_norm(res_d, (dim,), wd, bd, 1e-5)
    ((y_d * gy.double()).sum() + (res_d * gr.double()).sum()).backward()
    ytol, gtol = {torch.float32: (1e-6, 1e-5), torch.bfloat16: (4e-3, 5e-3), torch.float16: (5e-4, 7e-4)}[dt]
    assert rel_l2(res, res_d) < 1e-6
    assert rel_l2(y.float(), y_d) < ytol
    assert rel_l2(base.grad[:, :L, :dim].float(), hd.grad) < gtol
    assert float(base.grad[:, L:].abs().max()) == 0.0
    for a, bb in ((r.grad, rd.grad), (w.grad, wd.grad), (b.grad, bd.grad)):
        assert rel_l2(a, bb) < 1e-5


@pytest.mark.parametrize("dtype,tol", [(None, 2e-5), (torch.float16, 4e-3)])
@pytest.mark.parametrize("d_model,d_state", [(64, 16), (512, 64), (56, 12)])
def test_mamba_inner_single_node_equals_separate_ops(cuda, monkeypatch, dtype, tol, d_model, d_state):
    """Mamba.forward with everything between in_proj and out_proj as ONE autograd node (_MambaInnerFn: no slice / cat /
    cast glue, gradients written into shared buffers) against the same block built from the separate Functions
    (causal_conv1d_fn, _ProjFn, selective_scan_fn): same output, same input and parameter gradients."""
    from cleanumamba_amd.mamba_ssm.modules import mamba_simple as ms
    torch.manual_seed(d_model + d_state)
    blk = ms.Mamba(d_model, d_state=d_state, d_conv=4, expand=2).to(cuda)
    x = torch.randn(3, 37, d_model, generator=torch.Generator().manual_seed(1)).to(cuda)
    dout = torch.randn(3, 37, d_model, generator=torch.Generator().manual_seed(2)).to(cuda)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(ms, "_FUSED_INNER", fused)
        blk.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        if dtype is None:
            y = blk(xi)
        else:
            with torch.autocast("cuda", dtype=dtype):
                y = blk(xi)
        (y.float() * dout).sum().backward()
        res[fused] = (y.detach().float(), xi.grad.clone(), {k: p.grad.clone() for k, p in blk.named_parameters()})
    assert rel_l2(res[True][0], res[False][0]) < tol
    assert rel_l2(res[True][1], res[False][1]) < 3 * tol
    for k in res[True][2]:
        assert rel_l2(res[True][2][k], res[False][2][k]) < 3 * tol, k


@pytest.mark.parametrize("shape", [(2, 130, 64, 50), (2, 70, 20, 47), (3, 130, 16, 41), (2, 48, 8, 100), (1, 2048, 64, 624),
                                   (1, 128, 16, 624)])
def test_scan_with_a_log_flag_equals_minus_exp_outside(cuda, shape):
    """cum_scan_shape.delta_softplus bit 1 (CUM_SCAN_A_IS_LOG): `A` holds A_log, the kernels form A = -exp(A_log) (upstream
    Mamba.forward: `A = -torch.exp(self.A_log.float())`) and the backward returns dA_log = dA * A -- against the same op fed
    -exp(A_log) with torch's chain rule around it; sequential kernels of every d_state class and the time-parallel forms
    (the last two shapes)."""
    from cleanumamba_amd.mamba_ssm.ops.selective_scan_interface import A_IS_LOG, selective_scan_fn
    bsz, dim, N, L = shape
    gen = torch.Generator().manual_seed(7 + sum(shape))
    rn = lambda *s: torch.randn(*s, generator=gen)
    base = dict(u=rn(bsz, L, dim).transpose(1, 2), delta=0.5 * rn(bsz, L, dim).transpose(1, 2), A_log=0.5 * rn(dim, N),
                B=rn(bsz, L, N).transpose(1, 2), C=rn(bsz, L, N).transpose(1, 2), D=rn(dim),
                z=rn(bsz, L, dim).transpose(1, 2), delta_bias=0.5 * rn(dim))
    dout = rn(bsz, L, dim).transpose(1, 2).to(cuda)
    res = {}
    for flag in (False, True):
        t = {k: v.to(cuda).requires_grad_(True) for k, v in base.items()}
        A = t["A_log"] if flag else -torch.exp(t["A_log"])
        y = selective_scan_fn(t["u"], t["delta"], A, t["B"], t["C"], t["D"], z=t["z"], delta_bias=t["delta_bias"],
                              delta_softplus=(1 | A_IS_LOG) if flag else True)
        (y * dout).sum().backward()
        res[flag] = (y.detach(), {k: v.grad for k, v in t.items()})
    assert rel_l2(res[True][0], res[False][0]) < 2e-6
    for k in base:
        assert rel_l2(res[True][1][k], res[False][1][k]) < 2e-5, k
