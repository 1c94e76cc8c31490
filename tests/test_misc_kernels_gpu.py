"""Small C-ABI entry points against their torch equivalents (GPU): operand gather, hipFFT wrappers, the waveform-end
kernels of csrc/loss.hip (time-domain loss term, per-clip std, input / output framing)."""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src_dt,dst_dt", [(torch.float32, torch.bfloat16), (torch.float32, torch.float32),
                                           (torch.bfloat16, torch.float32), (torch.bfloat16, torch.bfloat16),
                                           (torch.float32, torch.float16), (torch.float16, torch.float32),
                                           (torch.float16, torch.float16)])
@pytest.mark.parametrize("n", [1, 7, 8, 1000, 100003])
def test_gather_with_zero_padding(cuda, src_dt, dst_dt, n):
    from cleanumamba_amd.network import convstack as cs
    g = torch.Generator().manual_seed(n)
    src = torch.randn(5000, generator=g).to(src_dt)
    idx = torch.randint(-1, 5000, (n,), generator=g, dtype=torch.int32)
    out = cs.gather(src.to(cuda), idx.to(cuda), dst_dt)
    want = torch.where(idx >= 0, src[idx.clamp_min(0).long()].float(), torch.zeros(n)).to(dst_dt)
    assert out.dtype == dst_dt and torch.equal(out.cpu(), want)      # pure data movement + one rounding: bit-exact


@pytest.mark.parametrize("n_fft", [512, 1024, 2048])
def test_rfft_irfft_match_torch_fft(cuda, n_fft):
    """Real transforms through caller-owned plans (cum_fft_plan_create / cum_fft_exec: hipFFT, unnormalised, inputs may be
    overwritten, work area passed in) vs torch.fft on a copy."""
    from cleanumamba_amd import hip
    batch = 37
    x = torch.randn(batch, n_fft, generator=torch.Generator().manual_seed(n_fft)).to(cuda)
    want = torch.fft.rfft(x.clone(), dim=-1)
    spec = torch.empty(batch, n_fft // 2 + 1, dtype=torch.complex64, device=cuda)
    scratch = x.clone()
    hip.fft(hip.FFT_R2C, n_fft, batch, scratch, torch.view_as_real(spec))
    assert rel_l2(torch.view_as_real(spec), torch.view_as_real(want)) < 1e-6
    back_want = torch.fft.irfft(want.clone(), n=n_fft, dim=-1, norm="forward")
    back = torch.empty(batch, n_fft, device=cuda)
    z = want.clone()
    hip.fft(hip.FFT_C2R, n_fft, batch, torch.view_as_real(z), back)
    assert rel_l2(back, back_want) < 1e-6
    assert rel_l2(back / n_fft, x) < 1e-5                            # round trip


@pytest.mark.parametrize("n", [256, 1024])
def test_cfft_matches_torch_fft(cuda, n):
    """Complex plan (hipFFT c2c, in place, unnormalised both ways) against torch.fft in float64."""
    from cleanumamba_amd import hip
    g = torch.Generator().manual_seed(n)
    z = torch.randn(37, n, 2, generator=g)
    ref = torch.fft.fft(torch.view_as_complex(z.double()))
    buf = z.to(cuda).contiguous()
    with torch.cuda.device(cuda):
        hip.fft(hip.FFT_C2C, n, 37, buf, buf)
    got = torch.view_as_complex(buf.double().cpu())
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    with torch.cuda.device(cuda):
        hip.fft(hip.FFT_C2C, n, 37, buf, buf, inverse=True)
    back = buf.cpu() / n
    assert float((back - z).abs().max()) < 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_stream_window_update_and_tail_rows(cuda, dt):
    """The streaming encoder glue against plain slicing: window <- cat(window[n_new:], new rows) with `fresh` either a
    whole recomputed window or only the new rows; tail_rows copies the newest rows into a compact clip buffer."""
    from cleanumamba_amd import hip
    S, rows, n_new, Cp, pitch = 5, 22, 8, 24, 24
    g = torch.Generator().manual_seed(3)
    window = torch.randn(S, pitch, Cp, generator=g).to(dt).to(cuda)
    full = torch.randn(S, pitch, Cp, generator=g).to(dt).to(cuda)
    want = torch.cat([window[:, n_new:rows], full[:, rows - n_new:rows]], 1)
    lib, dc = hip.lib(), hip.dtype_code(dt)
    for compact in (False, True):
        w = window.clone()
        fresh = full[:, rows - n_new:rows].contiguous() if compact else full
        tmp = torch.empty(S * rows * Cp, dtype=dt, device=cuda)
        with torch.cuda.device(cuda):
            nxt = torch.zeros(S, n_new + 4, Cp, dtype=dt, device=cuda)
            hip.check(lib.cum_stream_window_update(dc, S, rows, n_new, Cp, hip.ptr(w), hip.ptr(fresh), pitch,
                                                   n_new if compact else pitch, rows - n_new if compact else 0,
                                                   hip.ptr(tmp), hip.ptr(nxt) if compact else None, n_new + 4,
                                                   hip.stream_ptr()))
        assert torch.equal(w[:, :rows], want)
        assert torch.equal(w[:, rows:], window[:, rows:])                 # rows beyond the window are untouched
        if compact:       # the next layer's compact input: the n_new + 2 newest rows of the updated window
            assert torch.equal(nxt[:, :n_new + 2], want[:, rows - n_new - 2:]) and float(nxt[:, n_new + 2:].abs().max()) == 0
    dst = torch.zeros(S, 12, Cp, dtype=dt, device=cuda)
    with torch.cuda.device(cuda):
        hip.check(lib.cum_stream_tail_rows(dc, S, 10, Cp, hip.ptr(window), pitch, rows - 10, hip.ptr(dst), 12,
                                           hip.stream_ptr()))
    assert torch.equal(dst[:, :10], window[:, rows - 10:rows]) and float(dst[:, 10:].abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        hip.check(lib.cum_stream_tail_rows(dc, S, 10, Cp, hip.ptr(window), pitch, rows, hip.ptr(dst), 12,
                                           hip.stream_ptr()))                # source rows out of range


@pytest.mark.parametrize("rows,cols", [(114, 85), (85, 114), (300, 1000), (7, 3)])
def test_small_linear_matches_torch(cuda, rows, cols):
    """cum_small_linear (one workgroup per stream; LDS-staged matrix when it is small, wave-per-row otherwise)."""
    from cleanumamba_amd import hip
    g = torch.Generator().manual_seed(rows)
    x, W, b = torch.randn(37, cols, generator=g), torch.randn(rows, cols, generator=g) / cols ** 0.5, torch.randn(rows, generator=g)
    out = torch.empty(37, rows, device=cuda)
    xd, Wd, bd = x.to(cuda), W.to(cuda), b.to(cuda)
    with torch.cuda.device(cuda):
        hip.check(hip.lib().cum_small_linear(37, rows, cols, hip.ptr(xd), cols, hip.ptr(Wd), hip.ptr(bd), hip.ptr(out), rows,
                                             hip.stream_ptr()))
    ref = x.double() @ W.double().t() + b.double()
    assert rel_l2(out, ref) < 1e-6


@pytest.mark.parametrize("p", [1, 2])
@pytest.mark.parametrize("shape", [(1, 1, 1), (2, 1, 4099), (16, 1, 160000)])
def test_lp_loss_matches_torch_and_replays_in_a_graph(cuda, p, shape):
    """cum_lp_loss_fwd / _bwd (F.l1_loss / F.mse_loss of src/util/util.py:262-268) against torch in f64, and the value
    a captured graph returns on NEW data against the eager value -- the ATen reduction it replaces returned a stale sum
    from the second replay of the train-step graph on (tools/debug_graph_e8.py)."""
    import torch.nn.functional as F
    from cleanumamba_amd.network.convstack import LpLoss
    g = torch.Generator().manual_seed(sum(shape) + p)
    y = torch.randn(*shape, generator=g).to(cuda).requires_grad_(True)
    c = torch.randn(*shape, generator=g).to(cuda)
    if y.numel() > 10:
        with torch.no_grad():
            y[0, 0, 3] = c[0, 0, 3]                               # an exact tie: gradient 0, as torch.sgn
    loss = LpLoss.apply(y, c, p)
    (loss * 3.0).backward()
    yd = y.detach().double().requires_grad_(True)
    want = (F.l1_loss if p == 1 else F.mse_loss)(yd, c.double())
    (want * 3.0).backward()
    assert abs(float(loss) - float(want)) < 2e-6 * abs(float(want)) + 1e-12
    assert rel_l2(y.grad, yd.grad) < 1e-6
    two = LpLoss.apply(y.detach(), c, p)
    assert torch.equal(two, loss.detach())                         # fixed summation order: bit-reproducible
    ys, cs_ = y.detach().clone(), c.clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = LpLoss.apply(ys, cs_, p)
    for it in range(3):
        ys.copy_(torch.randn(*shape, generator=g).to(cuda))
        cs_.copy_(torch.randn(*shape, generator=g).to(cuda))
        graph.replay()
        assert torch.equal(out, LpLoss.apply(ys, cs_, p)), it


@pytest.mark.parametrize("shape", [(1, 2), (3, 777), (16, 160000), (2, 480000)])
def test_clip_std_and_framing(cuda, shape):
    """cum_clip_std against torch.std (f64) -- `noisy_audio.std(dim=2, keepdim=True) + 1e-3`, src/network/CleanUMamba.py:
    260-262 -- on signals with a large offset (one-pass Welford must not cancel); cum_frame_rows / cum_unframe_rows
    against the torch construction of the row buffer."""
    from cleanumamba_amd.network import convstack as cs
    B, L = shape
    g = torch.Generator().manual_seed(L)
    x = (0.05 * torch.randn(B, 1, L, generator=g) + torch.linspace(-3, 3, B).view(B, 1, 1)).to(cuda)
    std = cs.clip_std(x, 1e-3)
    want = x.double().std(dim=2, keepdim=True) + 1e-3
    assert std.shape == (B, 1, 1) and rel_l2(std, want) < 2e-6
    assert torch.equal(std, cs.clip_std(x, 1e-3))
    T = L + 5
    for dt in (torch.float32, torch.float16, torch.bfloat16):
        geo = cs.Geo(B, T, 1)
        buf = cs.frame_input(x, std, T, dt)
        ref = cs.to_rows(torch.nn.functional.pad(x / std, (0, T - L)), geo, dt)
        assert buf.shape == ref.shape and torch.equal(buf, ref)
        y = cs.Unframe.apply(buf.clone().requires_grad_(True), std, geo, L)
        assert torch.equal(y, cs.from_rows(ref, geo).float()[:, :, :L] * std)
        leaf = buf.clone().float().requires_grad_(True)
        dy = torch.randn(B, 1, L, generator=g).to(cuda)
        (cs.Unframe.apply(leaf.to(dt), std, geo, L) * dy).sum().backward()
        wleaf = ref.clone().float().requires_grad_(True)
        ((cs.from_rows(wleaf.to(dt), geo).float()[:, :, :L] * std) * dy).sum().backward()
        assert rel_l2(leaf.grad, wleaf.grad) < (1e-6 if dt == torch.float32 else 4e-3)
