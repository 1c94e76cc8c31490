"""Small C-ABI entry points against their torch equivalents (GPU): operand gather, hipFFT wrappers."""
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
    """cum_rfft / cum_irfft (hipFFT, unnormalised, inputs may be overwritten) vs torch.fft on a copy."""
    from cleanumamba_amd import hip
    batch = 37
    x = torch.randn(batch, n_fft, generator=torch.Generator().manual_seed(n_fft)).to(cuda)
    want = torch.fft.rfft(x.clone(), dim=-1)
    spec = torch.empty(batch, n_fft // 2 + 1, dtype=torch.complex64, device=cuda)
    scratch = x.clone()
    hip.check(hip.lib().cum_rfft(n_fft, batch, hip.ptr(scratch), hip.ptr(torch.view_as_real(spec)), hip.stream_ptr()))
    assert rel_l2(torch.view_as_real(spec), torch.view_as_real(want)) < 1e-6
    back_want = torch.fft.irfft(want.clone(), n=n_fft, dim=-1, norm="forward")
    back = torch.empty(batch, n_fft, device=cuda)
    z = want.clone()
    hip.check(hip.lib().cum_irfft(n_fft, batch, hip.ptr(torch.view_as_real(z)), hip.ptr(back), hip.stream_ptr()))
    assert rel_l2(back, back_want) < 1e-6
    assert rel_l2(back / n_fft, x) < 1e-5                            # round trip


@pytest.mark.parametrize("n", [256, 1024])
def test_cfft_matches_torch_fft(cuda, n):
    """cum_cfft (hipFFT c2c, in place, unnormalised both ways) against torch.fft in float64."""
    from cleanumamba_amd import hip
    g = torch.Generator().manual_seed(n)
    z = torch.randn(37, n, 2, generator=g)
    ref = torch.fft.fft(torch.view_as_complex(z.double()))
    buf = z.to(cuda).contiguous()
    with torch.cuda.device(cuda):
        hip.check(hip.lib().cum_cfft(n, 37, hip.ptr(buf), hip.ptr(buf), 0, hip.stream_ptr()))
    got = torch.view_as_complex(buf.double().cpu())
    assert float((got - ref).abs().max() / ref.abs().max()) < 2e-6
    with torch.cuda.device(cuda):
        hip.check(hip.lib().cum_cfft(n, 37, hip.ptr(buf), hip.ptr(buf), 1, hip.stream_ptr()))
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
