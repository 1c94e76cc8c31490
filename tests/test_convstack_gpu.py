"""Fused encoder / decoder layers (csrc/gemm.hip via network/convstack.py) vs the oracle's
F.conv1d / F.conv_transpose1d restatement, forward and backward, fp32 and bf16 (GPU)."""
import pytest
import torch

from conftest import record, rel_l2
from oracle import cleanumamba_ref as R

pytestmark = pytest.mark.gpu

# (forward, backward) rel-L2 bounds of one fused layer against the f64 oracle evaluated on the SAME 16-bit-rounded
# inputs and weights the kernels read: what is left is the rounding of the intermediate H-channel activation and of
# the outputs (bf16: 2^-9 per element, f16: 2^-12) and f32 accumulation.  (Against UNROUNDED operands the input
# gradient is dominated by ReLU gates that flip where |pre-activation| < one rounding step -- sqrt(2^-9) ~ 5 % -- which
# says nothing about the kernels.)  Bounds are ~3x the largest value measured over the shapes below
# (gpurun_out/test_measured.jsonl); a dropped K tile or a mis-packed weight row moves these by O(1 / K tiles) >> bound.
LAYER_TOL = {torch.float32: (2e-5, 1e-4), torch.bfloat16: (6e-3, 1.2e-2), torch.float16: (8e-4, 1.6e-3)}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def _rounded(t, dtype):
    """The value a 16-bit kernel path sees for an f32 operand."""
    return t.to(dtype).double()


def _store(dtype):
    """Straight-through rounding of the H-channel intermediate to the type the kernels store it in: without it the
    ReLU behind the decoder's transposed conv gates on pre-activations that differ by one rounding step of its INPUT,
    and the flipped gates (not the kernels) set the input-gradient error (3e-2 bf16 / 2e-2 f16 measured)."""
    if dtype == torch.float32:
        return None
    return lambda t: t + (t.to(dtype).double() - t).detach()


def _layer_params(cin, h, cout, seed):
    g = torch.Generator().manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g)
    return {"encoder.0.0.weight": rn(h, cin, 4) / (4 * cin) ** 0.5, "encoder.0.0.bias": 0.1 * rn(h),
            "encoder.0.2.weight": rn(2 * h, h, 1) / h ** 0.5, "encoder.0.2.bias": 0.1 * rn(2 * h),
            "decoder.0.0.weight": rn(2 * h, h, 1) / h ** 0.5, "decoder.0.0.bias": 0.1 * rn(2 * h),
            "decoder.0.2.weight": rn(h, cout, 4) / (2 * h) ** 0.5, "decoder.0.2.bias": 0.1 * rn(cout)}


@pytest.mark.parametrize("cin,h,tin", [(1, 64, 62), (53, 74, 30), (128, 256, 126), (768, 768, 14)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_encoder_layer_fwd_bwd(cuda, cin, h, tin, dtype):
    tol, btol = LAYER_TOL[dtype]
    tag = f"enc_layer[{cin}-{h}-{tin}-{dtype}]"
    from cleanumamba_amd.network import convstack as cs
    B = 3
    sd = _layer_params(cin, h, h, seed=cin + h)
    x = torch.randn(B, cin, tin, generator=torch.Generator().manual_seed(1))
    tout = (tin - 4) // 2 + 1
    dout = torch.randn(B, h, tout, generator=torch.Generator().manual_seed(2))
    ref = {k: (_rounded(v, dtype) if k.endswith("weight") else v.double()).requires_grad_(True)
           for k, v in sd.items() if k.startswith("encoder")}          # biases stay f32 in the kernels
    xr = _rounded(x, dtype).requires_grad_(True)
    yr = R.encoder_layer(ref, 0, xr, store=_store(dtype))
    (yr * dout.double()).sum().backward()

    dev = {k: v.to(cuda).requires_grad_(True) for k, v in sd.items() if k.startswith("encoder")}
    xd = x.to(cuda).requires_grad_(True)
    gi, gm, go = cs.Geo(B, tin, cin), cs.Geo(B, tout, h), cs.Geo(B, tout, h)
    buf = cs.to_rows(xd, gi, dtype)
    y1 = cs.ConvK4S2ReLU.apply(buf, dev["encoder.0.0.weight"], dev["encoder.0.0.bias"], gi, gm)
    ybuf = cs.PointwiseGLU.apply(y1, dev["encoder.0.2.weight"], dev["encoder.0.2.bias"], gm, go, True)
    y = cs.from_rows(ybuf, go).float()
    assert record(tag + ".fwd", rel_l2(y, yr)) < tol
    # closing rows / padded channels stay zero
    rows = go.rows(ybuf)
    assert float(rows[:, go.T:].abs().max()) == 0 and float(rows[:, :, go.C:].abs().max() if go.Cp > go.C else 0) == 0
    (y * dout.to(cuda)).sum().backward()
    assert record(tag + ".dx", rel_l2(xd.grad, xr.grad)) < btol
    for k in dev:
        assert record(tag + ".d" + k, rel_l2(dev[k].grad, ref[k].grad)) < btol, k


@pytest.mark.parametrize("h,cout,t,relu,with_skip", [(64, 1, 30, False, False), (74, 53, 14, True, True),
                                                     (256, 128, 62, True, True), (768, 768, 6, True, True)])
@pytest.mark.parametrize("dtype", DTYPES)
def test_decoder_layer_fwd_bwd(cuda, h, cout, t, relu, with_skip, dtype):
    tol, btol = LAYER_TOL[dtype]
    tag = f"dec_layer[{h}-{cout}-{t}-{dtype}]"
    from cleanumamba_amd.network import convstack as cs
    B = 2
    sd = _layer_params(h, h, cout, seed=h + cout)
    x = torch.randn(B, h, t, generator=torch.Generator().manual_seed(3))
    skip = torch.randn(B, cout, 2 * t + 2, generator=torch.Generator().manual_seed(4))
    dout = torch.randn(B, cout, 2 * t + 2, generator=torch.Generator().manual_seed(5))
    ref = {k: (_rounded(v, dtype) if k.endswith("weight") else v.double()).requires_grad_(True)
           for k, v in sd.items() if k.startswith("decoder")}
    xr, sr = _rounded(x, dtype).requires_grad_(True), _rounded(skip, dtype).requires_grad_(True)
    yr = R.decoder_layer(ref, 0, xr, last=not relu, store=_store(dtype))
    if with_skip:
        yr = yr + sr
    (yr * dout.double()).sum().backward()

    dev = {k: v.to(cuda).requires_grad_(True) for k, v in sd.items() if k.startswith("decoder")}
    xd, sk = x.to(cuda).requires_grad_(True), skip.to(cuda).requires_grad_(True)
    gi, gg, go = cs.Geo(B, t, h), cs.Geo(B, t, h), cs.Geo(B, 2 * t + 2, cout)
    buf = cs.to_rows(xd, gi, dtype)
    sbuf = cs.to_rows(sk, go, dtype) if with_skip else None
    g = cs.PointwiseGLU.apply(buf, dev["decoder.0.0.weight"], dev["decoder.0.0.bias"], gi, gg, True)
    ybuf = cs.ConvT4S2.apply(g, dev["decoder.0.2.weight"], dev["decoder.0.2.bias"], sbuf, gg, go, relu)
    y = cs.from_rows(ybuf, go).float()
    assert record(tag + ".fwd", rel_l2(y, yr)) < tol
    assert float(go.rows(ybuf)[:, go.T:].abs().max()) == 0
    (y * dout.to(cuda)).sum().backward()
    assert record(tag + ".dx", rel_l2(xd.grad, xr.grad)) < btol
    if with_skip:
        assert record(tag + ".dskip", rel_l2(sk.grad, sr.grad)) < btol
    for k in dev:
        assert record(tag + ".d" + k, rel_l2(dev[k].grad, ref[k].grad)) < btol, k


def test_pointwise_with_residual(cuda):
    from cleanumamba_amd.network import convstack as cs
    B, cin, cout, t = 2, 512, 768, 10
    g = torch.Generator().manual_seed(9)
    w, b = torch.randn(cout, cin, 1, generator=g) / cin ** 0.5, torch.randn(cout, generator=g)
    x, s = torch.randn(B, cin, t, generator=g), torch.randn(B, cout, t, generator=g)
    yr = torch.nn.functional.conv1d(x.double(), w.double(), b.double()) + s.double()
    gi, go = cs.Geo(B, t, cin), cs.Geo(B, t, cout)
    wd, bd = w.to(cuda).requires_grad_(True), b.to(cuda).requires_grad_(True)
    xd = x.to(cuda).requires_grad_(True)
    ybuf = cs.Pointwise.apply(cs.to_rows(xd, gi, torch.float32), wd, bd, cs.to_rows(s.to(cuda), go, torch.float32),
                              gi, go)
    y = cs.from_rows(ybuf, go)
    assert rel_l2(y, yr) < 2e-5
    y.square().sum().backward()
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (torch.nn.functional.conv1d(xr, wr, b.double()) + s.double()).square().sum().backward()
    assert rel_l2(xd.grad, xr.grad) < 1e-4 and rel_l2(wd.grad, wr.grad) < 1e-4


def _ref_gemm(A, W, bias):
    return A.double() @ W.double().t() + (0 if bias is None else bias.double())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 6e-3), (torch.float16, 8e-4)])   # output rounding only
@pytest.mark.parametrize("M,N,K", [(300, 64, 64), (1000, 192, 512), (60001, 256, 256), (140003, 192, 256)])
def test_gemm_backward_epilogues_against_torch(cuda, M, N, K, dtype, tol):
    """The backward epilogues of cum_gemm_nt through the C ABI, ragged M (edge tiles).  All four shapes run the
    128 x 128 kernel (f32: the 256 x 128 one on the largest; asserted below -- the 256 x 256 ping-pong kernel's epilogues
    are pinned by test_dispatch_map_gpu.py on shapes the library reports as taking it):
    3 = ReLU gate from a full activation and from sign nibbles (+ ungated second output);
    4 = GLU backward from the packed (a | b) pre-activation and from the gate-only form (+ residual)."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.network import convstack as cs
    g = torch.Generator().manual_seed(M + N + K)
    rn = lambda *s: torch.randn(*s, generator=g)
    A = rn(M, K).to(cuda).to(dtype)
    W = (rn(N, K) / K ** 0.5).to(cuda).to(dtype)
    acc = _ref_gemm(A.cpu().float(), W.cpu().float(), None)            # what the MFMA accumulates (inputs already rounded)
    import ctypes
    desc = hip.GemmDesc()
    desc.dtype, desc.M, desc.N, desc.K = hip.dtype_code(dtype), M, N, K
    tile = hip.lib().cum_gemm_nt_tile(ctypes.byref(desc))
    assert tile == 128 or (dtype == torch.float32 and tile == 256)      # f32: 256 x 128 from 1 024 tiles on

    # ---- epilogue 3
    Y = rn(M, N).to(cuda).to(dtype)                                     # activation whose sign gates
    out = torch.full((M, N), float("nan"), device=cuda, dtype=dtype)
    ung = torch.full((M, N), float("nan"), device=cuda, dtype=dtype)
    cs.gemm(A, 0, K, W, None, out, 0, N, M, 1 << 30, 1 << 30, hip.EPI_MASK, N, res=Y, r_off=0, ldr=N, aux=ung, x_off=0, ldz=N)
    want = torch.where(Y.cpu().double() > 0, acc, torch.zeros_like(acc))
    assert rel_l2(out.float(), want) < tol and rel_l2(ung.float(), acc) < tol
    bits = torch.zeros(M * N // 4, dtype=torch.uint8, device=cuda)
    yq = (Y.float().cpu() > 0).view(M, N // 4, 4).to(torch.uint8)
    bits.copy_((yq[..., 0] | yq[..., 1] << 1 | yq[..., 2] << 2 | yq[..., 3] << 3).reshape(-1).to(cuda))
    out2 = torch.full((M, N), float("nan"), device=cuda, dtype=dtype)
    cs.gemm(A, 0, K, W, None, out2, 0, N, M, 1 << 30, 1 << 30, hip.EPI_MASK, N, res=bits, r_off=0, ldr=N, mask_bits=True)
    assert torch.equal(out2, out)

    # ---- producer of the sign nibbles: epilogue 1 with mask_bits
    bias = rn(N).to(cuda)
    relu_out = torch.empty(M, N, device=cuda, dtype=dtype)
    nib = torch.full((M * N // 4,), 255, dtype=torch.uint8, device=cuda)
    cs.gemm(A, 0, K, W, bias, relu_out, 0, N, M, 1 << 30, 1 << 30, hip.EPI_RELU, N, aux=nib, x_off=0, ldz=N, mask_bits=True)
    rq = (relu_out.float().cpu() > 0).view(M, N // 4, 4).to(torch.uint8)
    assert torch.equal(nib.cpu(), (rq[..., 0] | rq[..., 1] << 1 | rq[..., 2] << 2 | rq[..., 3] << 3).reshape(-1))

    # ---- epilogue 4: d = acc + ext is the gradient of y = a * sig(b)
    ext = rn(M, N).to(cuda).to(dtype)
    a, b = rn(M, N), rn(M, N)
    Z = torch.empty(M, 2 * N)
    Zv = Z.view(M, N // 16, 2, 16)
    Zv[:, :, 0], Zv[:, :, 1] = a.view(M, N // 16, 16), b.view(M, N // 16, 16)
    Z = Z.to(cuda).to(dtype)
    ar, br = Z.float().cpu().view(M, N // 16, 2, 16)[:, :, 0].reshape(M, N).double(), Z.float().cpu().view(M, N // 16, 2, 16)[:, :, 1].reshape(M, N).double()
    d = acc + ext.float().cpu().double()
    sg = torch.sigmoid(br)
    want_dz = torch.empty(M, N // 16, 2, 16, dtype=torch.float64)
    want_dz[:, :, 0], want_dz[:, :, 1] = (d * sg).view(M, N // 16, 16), (d * ar * sg * (1 - sg)).view(M, N // 16, 16)
    dz = torch.full((M, 2 * N), float("nan"), device=cuda, dtype=dtype)
    cs.gemm(A, 0, K, W, None, dz, 0, 2 * N, M, 1 << 30, 1 << 30, hip.EPI_GLU_BWD, N, res=ext, r_off=0, ldr=N, aux=Z, x_off=0, ldz=2 * N)
    assert rel_l2(dz.float(), want_dz.view(M, 2 * N)) < tol
    # gate-only form: b [M][N] and the saved output y
    bg = br.to(dtype).to(cuda)
    y = (ar * sg).to(dtype).to(cuda)
    sgq = torch.sigmoid(bg.float().cpu().double())
    want2 = torch.empty(M, N // 16, 2, 16, dtype=torch.float64)
    want2[:, :, 0], want2[:, :, 1] = (d * sgq).view(M, N // 16, 16), (d * y.float().cpu().double() * (1 - sgq)).view(M, N // 16, 16)
    dz2 = torch.full((M, 2 * N), float("nan"), device=cuda, dtype=dtype)
    cs.gemm(A, 0, K, W, None, dz2, 0, 2 * N, M, 1 << 30, 1 << 30, hip.EPI_GLU_BWD, N, res=ext, r_off=0, ldr=N, aux=bg, x_off=0, ldz=N,
            aux2=y, y_off=0, ldy=N, gate_only=True)
    assert rel_l2(dz2.float(), want2.view(M, 2 * N)) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.bfloat16, 2e-6), (torch.float16, 2e-6)])   # f32 accumulation of exact products
@pytest.mark.parametrize("M,N,K,ldx", [(300, 64, 64, 64), (999, 256, 512, 256), (70001, 512, 256, 256), (20003, 768, 1024, 512),
                                        (4100, 256, 256, 256)])
def test_weight_gradient_gemm_against_torch(cuda, M, N, K, ldx, dtype, tol):
    """cum_gemm_tn through the C ABI: dW = dZ^T X with overlapping X rows (ldx < K: the k=4/s=2 window) and the fused
    bias gradient, ragged M.  The first shape runs the 128x128 kernel, the others (16-bit types, N and K multiples of
    256) the 256x256 one; inputs are already rounded, so only the f32 summation order differs from the f64 reference."""
    from cleanumamba_amd.network import convstack as cs
    g = torch.Generator().manual_seed(M + N + K)
    dz = torch.randn(M, N, generator=g).to(cuda).to(dtype)
    xflat = torch.randn(M * ldx + K, generator=g).to(cuda).to(dtype)
    dw, db = cs.wgrad(dz, 0, N, N, xflat, 0, ldx, K, M)
    X = torch.as_strided(xflat.cpu().double(), (M, K), (ldx, 1))
    want_w, want_b = dz.cpu().double().t() @ X, dz.cpu().double().sum(0)
    assert dw.shape == (N, K) and db.shape == (N,)
    assert record(f"wgrad[{M}x{N}x{K}-{dtype}].dW", rel_l2(dw, want_w)) < tol
    assert record(f"wgrad[{M}x{N}x{K}-{dtype}].db", rel_l2(db, want_b)) < tol


@pytest.mark.parametrize("tin", [62, 510, 4100])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_first_encoder_layer(cuda, tin, dtype, monkeypatch):
    """csrc/enc0.hip -- the first encoder layer (1 -> 64 channels) with its ReLU intermediate rebuilt from the input
    instead of stored (src/network/CleanUMamba.py:108-113 at channels_input 1, channels_H 64) -- through EncoderStack:
    forward and all four parameter gradients against the f64 oracle (the per-layer bounds), against the generic GEMM
    route on the same inputs, bit-reproducible, and really taken."""
    from cleanumamba_amd.network import convstack as cs
    tol, btol = LAYER_TOL[dtype]
    B, cin, h = 3, 1, 64
    sd = _layer_params(cin, h, h, seed=7)
    x = torch.randn(B, cin, tin, generator=torch.Generator().manual_seed(11))
    tout = (tin - 4) // 2 + 1
    dout = torch.randn(B, h, tout, generator=torch.Generator().manual_seed(12))
    ref = {k: (_rounded(v, dtype) if k.endswith("weight") else v.double()).requires_grad_(True)
           for k, v in sd.items() if k.startswith("encoder")}
    yr = R.encoder_layer(ref, 0, _rounded(x, dtype), store=_store(dtype))
    (yr * dout.double()).sum().backward()
    gi, gm, go = cs.Geo(B, tin, cin), cs.Geo(B, tout, h), cs.Geo(B, tout, h)
    keys = ["encoder.0.0.weight", "encoder.0.0.bias", "encoder.0.2.weight", "encoder.0.2.bias"]
    calls = []
    real_fwd, real_bwd = cs._enc0_fwd, cs._enc0_bwd
    monkeypatch.setattr(cs, "_enc0_fwd", lambda *a, **k: (calls.append("fwd"), real_fwd(*a, **k))[1])
    monkeypatch.setattr(cs, "_enc0_bwd", lambda *a, **k: (calls.append("bwd"), real_bwd(*a, **k))[1])

    def run(fused):
        monkeypatch.setattr(cs, "_ENC0_FUSED", fused)
        dev = {k: sd[k].to(cuda).requires_grad_(True) for k in keys}
        buf = cs.to_rows(x.to(cuda), gi, dtype)
        (ybuf,) = cs.EncoderStack.apply(buf, [(gi, gm, go)], True, *[dev[k] for k in keys])
        y = cs.from_rows(ybuf, go).float()
        rows = go.rows(ybuf)
        assert float(rows[:, go.T:].abs().max()) == 0 and float(ybuf[0].abs().max()) == 0     # framing rows stay zero
        (y * dout.to(cuda)).sum().backward()
        return y.detach(), {k: dev[k].grad.detach() for k in keys}
    y_f, g_f = run(True)
    assert calls == ["fwd", "bwd"]
    y_f2, g_f2 = run(True)
    assert torch.equal(y_f, y_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in keys)              # deterministic
    y_g, g_g = run(False)
    assert calls == ["fwd", "bwd"] * 2                                                           # generic route: not called
    tag = f"enc0_fused[{tin}-{dtype}]"
    assert record(tag + ".fwd", rel_l2(y_f, yr)) < tol
    assert record(tag + ".fwd_vs_generic", rel_l2(y_f, y_g)) < tol
    for k in keys:
        assert record(tag + ".d" + k, rel_l2(g_f[k], ref[k].grad)) < btol, k
        assert rel_l2(g_f[k], g_g[k]) < btol, k


@pytest.mark.parametrize("tin", [38, 270, 4100])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_width128_encoder_layer_forward(cuda, tin, dtype, monkeypatch):
    """csrc/ench.hip -- an encoder layer 64 -> 128 -> 128 (Conv1d k4 s2, ReLU, Conv1d 1x1, GLU: the second layer of E6 /
    E8, src/network/CleanUMamba.py:108-113) as one forward launch -- through EncoderStack: forward and all four parameter
    gradients (the unchanged backward on what the fused forward saved) against the f64 oracle, against the two-launch
    route on the same inputs (saved tensors included: hidden activation, sign nibbles, gate), framing rows zero,
    bit-reproducible, really taken; and the no-backward form (nothing but the output stored)."""
    from cleanumamba_amd.network import convstack as cs
    tol, btol = LAYER_TOL[dtype]
    B, cin, h = 3, 64, 128
    sd = _layer_params(cin, h, h, seed=17)
    x = torch.randn(B, cin, tin, generator=torch.Generator().manual_seed(21))
    tout = (tin - 4) // 2 + 1
    dout = torch.randn(B, h, tout, generator=torch.Generator().manual_seed(22))
    ref = {k: (_rounded(v, dtype) if k.endswith("weight") else v.double()).requires_grad_(True)
           for k, v in sd.items() if k.startswith("encoder")}
    yr = R.encoder_layer(ref, 0, _rounded(x, dtype), store=_store(dtype))
    (yr * dout.double()).sum().backward()
    gi, gm, go = cs.Geo(B, tin, cin), cs.Geo(B, tout, h), cs.Geo(B, tout, h)
    keys = ["encoder.0.0.weight", "encoder.0.0.bias", "encoder.0.2.weight", "encoder.0.2.bias"]
    calls, saved = [], {}
    real_fwd = cs._ench_fwd

    def spy(*a, **k):
        out = real_fwd(*a, **k)
        calls.append("fwd")
        saved["fused"] = out
        return out
    monkeypatch.setattr(cs, "_ench_fwd", spy)
    real_conv, real_glu = cs._conv_relu_fwd, cs._glu_fwd
    monkeypatch.setattr(cs, "_conv_relu_fwd", lambda *a, **k: saved.setdefault("conv", real_conv(*a, **k)))
    monkeypatch.setattr(cs, "_glu_fwd", lambda *a, **k: saved.setdefault("glu", real_glu(*a, **k)))

    def run(fused, save_z=True):
        monkeypatch.setattr(cs, "_ENCH_FUSED", fused)
        dev = {k: sd[k].to(cuda).requires_grad_(True) for k in keys}
        buf = cs.to_rows(x.to(cuda), gi, dtype)
        (ybuf,) = cs.EncoderStack.apply(buf, [(gi, gm, go)], save_z, *[dev[k] for k in keys])
        y = cs.from_rows(ybuf, go).float()
        rows = go.rows(ybuf)
        assert float(rows[:, go.T:].abs().max()) == 0 and float(ybuf[0].abs().max()) == 0     # framing rows stay zero
        assert float(ybuf[1 + go.M:].abs().max()) == 0
        if not save_z:
            return y.detach(), None
        (y * dout.to(cuda)).sum().backward()
        return y.detach(), {k: dev[k].grad.detach() for k in keys}
    y_f, g_f = run(True)
    assert calls == ["fwd"]
    y_f2, g_f2 = run(True)
    assert torch.equal(y_f, y_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in keys)              # deterministic
    y_g, g_g = run(False)
    assert calls == ["fwd"] * 2                                                                   # two-launch route: not called
    # what the fused forward saved for the backward is what the two launches save
    y1_f, bits_f, _, z_f = saved["fused"]
    (y1_g, bits_g), (_, z_g) = saved["conv"], saved["glu"]
    assert float(y1_f[0].abs().max()) == 0 and float(y1_f[1 + gm.M:].abs().max()) == 0
    assert rel_l2(y1_f.float(), y1_g.float()) < tol
    live = slice(gm.Cp // 4, (1 + gm.M) * gm.Cp // 4)
    same = (bits_f[live] == bits_g[live]).float().mean()
    assert float(same) > 0.999                        # (a sign can differ where the two summation orders round across 0)
    assert rel_l2(z_f.float(), z_g.float()) < tol
    tag = f"ench_fused[{tin}-{dtype}]"
    assert record(tag + ".fwd", rel_l2(y_f, yr)) < tol
    assert record(tag + ".fwd_vs_generic", rel_l2(y_f, y_g)) < tol
    for k in keys:
        assert record(tag + ".d" + k, rel_l2(g_f[k], ref[k].grad)) < btol, k
        assert rel_l2(g_f[k], g_g[k]) < btol, k
    y_n, _ = run(True, save_z=False)                  # inference form: hidden activation / nibbles / gate not stored
    assert saved["fused"][0] is None and saved["fused"][1] is None and saved["fused"][3] is None
    assert torch.equal(y_n, y_f)


@pytest.mark.parametrize("t", [15, 127, 1030])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_width128_decoder_layer_forward(cuda, t, dtype, monkeypatch):
    """csrc/dech.hip -- a decoder layer 128 -> 128 -> 64 (Conv1d 1x1, GLU, ConvTranspose1d k4 s2, ReLU, + skip: the
    second-to-last layer of E6 / E8, src/network/CleanUMamba.py:121-130, 313-316) as one forward launch -- as layer 0 of a
    two-layer DecoderStack (layer 1: the fused last layer): the layer's output against the f64 oracle and against the
    two-launch route, every tensor it saves for the backward (GLU output, gate, sign nibbles) against what the two
    launches save, every gradient of the stack (the unchanged backward) against the two-launch route and the f64 chain,
    framing rows zero, bit-reproducible, really taken; and the no-backward form."""
    from cleanumamba_amd.network import convstack as cs
    tol, btol = LAYER_TOL[dtype]
    B, h0, h1 = 3, 128, 64
    sd = {k: v for k, v in _layer_params(h0, h0, h1, seed=31).items() if k.startswith("decoder")}
    sd.update({k.replace("decoder.0", "decoder.1"): v for k, v in _layer_params(h1, h1, 1, seed=32).items() if k.startswith("decoder")})
    keys = [f"decoder.{j}.{i}.{n}" for j in range(2) for i in (0, 2) for n in ("weight", "bias")]
    t1, t2 = 2 * t + 2, 2 * (2 * t + 2) + 2
    gen = torch.Generator().manual_seed(33)
    x, skip, dout = torch.randn(B, h0, t, generator=gen), torch.randn(B, h1, t1, generator=gen), torch.randn(B, 1, t2, generator=gen)
    g0 = (cs.Geo(B, t, h0), cs.Geo(B, t, h0), cs.Geo(B, t1, h1))
    g1 = (cs.Geo(B, t1, h1), cs.Geo(B, t1, h1), cs.Geo(B, t2, 1))
    calls, saved = [], {}
    real_fwd, real_glu, real_convt = cs._dech_fwd, cs._glu_fwd, cs._convt_fwd

    def spy(*a, **k):
        out = real_fwd(*a, **k)
        calls.append("fwd")
        saved["fused"] = out
        return out
    monkeypatch.setattr(cs, "_dech_fwd", spy)
    monkeypatch.setattr(cs, "_glu_fwd", lambda *a, **k: saved.setdefault("glu", real_glu(*a, **k)))
    monkeypatch.setattr(cs, "_convt_fwd", lambda *a, **k: saved.setdefault("convt", real_convt(*a, **k)))

    def run(fused, save_z=True):
        monkeypatch.setattr(cs, "_DECH_FUSED", fused)
        dev = {k: sd[k].to(cuda).requires_grad_(True) for k in keys}
        xd, sk = x.to(cuda).requires_grad_(True), skip.to(cuda).requires_grad_(True)
        ybuf = cs.DecoderStack.apply(cs.to_rows(xd, g0[0], dtype), [g0, g1], save_z, 1, cs.to_rows(sk, g0[2], dtype),
                                     *[dev[k] for k in keys])
        y = cs.from_rows(ybuf, g1[2]).float()
        if not save_z:
            return y.detach(), None
        (y * dout.to(cuda)).sum().backward()
        return y.detach(), {**{k: dev[k].grad.detach() for k in keys}, "x": xd.grad.detach(), "skip": sk.grad.detach()}
    y_f, g_f = run(True)
    assert calls == ["fwd"]
    y_f2, g_f2 = run(True)
    assert torch.equal(y_f, y_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in g_f)              # deterministic
    y_g, g_g = run(False)
    assert calls == ["fwd"] * 2                                                                   # two-launch route: not called
    gg, go = g0[1], g0[2]
    gb_f, z_f, u1_f, act_f = saved["fused"]
    (gb_g, z_g), (u1_g, act_g) = saved["glu"], saved["convt"]
    # framing of the two row buffers the fused kernel writes
    assert float(gb_f[0].abs().max()) == 0 and float(gb_f[1 + gg.M:].abs().max()) == 0
    assert float(u1_f[0].abs().max()) == 0 and float(u1_f[1 + go.M:].abs().max()) == 0
    assert float(go.rows(u1_f)[:, go.T:].abs().max()) == 0 and float(gg.rows(gb_f)[:, gg.T:].abs().max()) == 0
    assert rel_l2(gb_f.float(), gb_g.float()) < tol and rel_l2(z_f.float(), z_g.float()) < tol
    assert rel_l2(u1_f.float(), u1_g.float()) < tol
    live = slice(go.Cp // 4, (1 + go.M) * go.Cp // 4)
    assert float((act_f[live] == act_g[live]).float().mean()) > 0.999     # (a sign can differ where summation orders round across 0)
    tag = f"dech_fused[{t}-{dtype}]"
    assert record(tag + ".fwd_vs_generic", rel_l2(y_f, y_g)) < tol
    for k in g_f:
        assert record(tag + ".vs_generic.d" + k, rel_l2(g_f[k], g_g[k])) < btol, k
    # f64 oracle: the layer's own output, and the chain of both layers for the gradients
    ref = {k: (_rounded(v, dtype) if k.endswith("weight") else v.double()).requires_grad_(True) for k, v in sd.items()}
    xr, sr = _rounded(x, dtype).requires_grad_(True), _rounded(skip, dtype).requires_grad_(True)
    u1r = R.decoder_layer(ref, 0, xr, last=False, store=_store(dtype)) + sr
    assert record(tag + ".fwd", rel_l2(cs.from_rows(u1_f, go).float(), u1r)) < tol
    yr = R.decoder_layer(ref, 1, u1r, last=True, store=_store(dtype))
    (yr * dout.double()).sum().backward()
    assert record(tag + ".chain_fwd", rel_l2(y_f, yr)) < 2 * tol
    assert record(tag + ".dx", rel_l2(g_f["x"], xr.grad)) < 2 * btol
    assert record(tag + ".dskip", rel_l2(g_f["skip"], sr.grad)) < 2 * btol
    for k in keys[:4]:
        assert record(tag + ".d" + k, rel_l2(g_f[k], ref[k].grad)) < 2 * btol, k
    y_n, _ = run(True, save_z=False)                  # inference form: GLU output / gate / nibbles not stored
    assert saved["fused"][0] is None and saved["fused"][1] is None and saved["fused"][3] is None
    assert torch.equal(y_n, y_f)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
def test_index_free_repack_equals_index_gather(cuda, dtype):
    """cum_pack2d (row / column offset tables, LDS transpose, contiguous-run fast path) against cum_gather over the same
    layouts, bit for bit: straight and transposed sources, padded rows and columns, ragged tile edges, runs of 8 on and off
    16-byte boundaries, several jobs in one launch."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.network import convstack as cs
    import numpy as np
    gen = torch.Generator().manual_seed(3)
    src = torch.randn(200_000, generator=gen).to(cuda)
    cases = []                                         # (g2 [R, C] int64 with -1 padding)
    # conv weight [Cout=70, Cin=33, K=4] at offset 1234, packed [72, 4 * 40] tap-major
    g = torch.full((72, 160), -1, dtype=torch.int64)
    o, t, i = torch.meshgrid(torch.arange(70), torch.arange(4), torch.arange(33), indexing="ij")
    g[o.reshape(-1), (t * 40 + i).reshape(-1)] = (1234 + (o * 33 + i) * 4 + t).reshape(-1)
    cases.append(g)
    cases.append(g.t().contiguous()[:, :72])            # its transpose (source fast along destination rows)
    # linear weight [N=130, K=96] at 50000 as-is and transposed with zero-padded tail columns
    w = 50000 + torch.arange(130 * 96).view(130, 96)
    cases.append(w.clone())
    wt = torch.full((96, 136), -1, dtype=torch.int64)
    wt[:, :130] = w.t()
    cases.append(wt)
    # the contiguous-run fast path: the same weight padded to [136, 104] (runs of 8 on 16-byte boundaries: runs8 = 2),
    # and at an odd offset (runs8 = 1: scalar loads of a run)
    wp = torch.full((136, 104), -1, dtype=torch.int64)
    wp[:130, :96] = w
    cases.append(wp)
    wo = torch.full((136, 104), -1, dtype=torch.int64)
    wo[:130, :96] = w + 70001
    cases.append(wo)
    jobs, tiles, tables, pos, off = [], [], [], 0, 0
    for g2 in cases:
        sep = cs._separable(g2)
        assert sep is not None
        ro, co, tr = sep
        R_, C_ = g2.shape
        jobs.append((off, R_, C_, pos, pos + R_, int(tr), 0 if tr else cs._runs8(ro, co)))
        tables += [ro, co]
        pos += R_ + C_
        tiles += [(len(jobs) - 1, a, b) for a in range((R_ + 63) // 64) for b in range((C_ + 63) // 64)]
        off += (R_ * C_ + 7) // 8 * 8
    assert [j[5] for j in jobs] == [0, 1, 0, 1, 0, 0]
    assert [j[6] for j in jobs] == [0, 0, 2, 0, 2, 1]      # tap-major conv weight: no runs; plain weights: runs of 8
    jb = np.array(jobs, dtype=[("off", "<i8"), ("rows", "<i4"), ("cols", "<i4"), ("rt", "<i4"), ("ct", "<i4"), ("tr", "<i4"),
                               ("runs8", "<i4")])
    jbt = torch.from_numpy(jb.view(np.uint8).copy()).to(cuda)
    tl = torch.tensor(tiles, dtype=torch.int32, device=cuda)
    tb = torch.cat(tables).to(cuda)
    out = torch.full((off,), 7.0, dtype=dtype, device=cuda)
    hip.check(hip.lib().cum_pack2d(hip.ptr(src), hip.ptr(jbt), hip.ptr(tl), len(tiles), hip.ptr(tb), hip.dtype_code(dtype),
                                   hip.ptr(out), hip.stream_ptr()))
    for g2, j in zip(cases, jobs):
        want = cs.gather(src, g2.reshape(-1).to(torch.int32).to(cuda), dtype)
        got = out[j[0]:j[0] + g2.numel()]
        assert torch.equal(want, got), (g2.shape, j)


@pytest.mark.parametrize("t", [15, 127, 1030])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_last_decoder_layer(cuda, t, dtype, monkeypatch):
    """csrc/dec7.hip -- the last decoder layer (64 -> 128 1x1, GLU, 64 -> 1 transposed conv) with its GLU output rebuilt
    from the layer input instead of stored (src/network/CleanUMamba.py:121-130 at channels_output 1, channels_H 64) --
    through a two-layer DecoderStack (the layer below supplies the ReLU sign bits the fused backward gates with):
    output, the last layer's four parameter gradients and the gradient of its input (= of the skip) against the f64
    oracle (the per-layer bounds); every gradient of the stack against the generic GEMM route on the same inputs;
    bit-reproducible; and really taken."""
    from cleanumamba_amd.network import convstack as cs
    tol, btol = LAYER_TOL[dtype]
    B, h0, h1 = 3, 128, 64
    sd = {k: v for k, v in _layer_params(h0, h0, h1, seed=21).items() if k.startswith("decoder")}
    sd.update({k.replace("decoder.0", "decoder.1"): v for k, v in _layer_params(h1, h1, 1, seed=22).items() if k.startswith("decoder")})
    keys = [f"decoder.{j}.{i}.{n}" for j in range(2) for i in (0, 2) for n in ("weight", "bias")]
    t1, t2 = 2 * t + 2, 2 * (2 * t + 2) + 2
    gen = torch.Generator().manual_seed(23)
    x, skip, dout = torch.randn(B, h0, t, generator=gen), torch.randn(B, h1, t1, generator=gen), torch.randn(B, 1, t2, generator=gen)
    g0 = (cs.Geo(B, t, h0), cs.Geo(B, t, h0), cs.Geo(B, t1, h1))
    g1 = (cs.Geo(B, t1, h1), cs.Geo(B, t1, h1), cs.Geo(B, t2, 1))
    calls = []
    real_fwd, real_bwd = cs._dec7_fwd, cs._dec7_bwd
    monkeypatch.setattr(cs, "_dec7_fwd", lambda *a, **k: (calls.append("fwd"), real_fwd(*a, **k))[1])
    monkeypatch.setattr(cs, "_dec7_bwd", lambda *a, **k: (calls.append("bwd"), real_bwd(*a, **k))[1])

    def run(fused):
        monkeypatch.setattr(cs, "_DEC7_FUSED", fused)
        dev = {k: sd[k].to(cuda).requires_grad_(True) for k in keys}
        xd, sk = x.to(cuda).requires_grad_(True), skip.to(cuda).requires_grad_(True)
        ybuf = cs.DecoderStack.apply(cs.to_rows(xd, g0[0], dtype), [g0, g1], True, 1, cs.to_rows(sk, g0[2], dtype),
                                     *[dev[k] for k in keys])
        go = g1[2]
        y = cs.from_rows(ybuf, go).float()
        rows = go.rows(ybuf)
        assert float(rows[:, go.T:].abs().max()) == 0 and float(ybuf[0].abs().max()) == 0      # framing rows stay zero
        assert float(rows[:, :, 1:].abs().max()) == 0                                             # padding channels too
        (y * dout.to(cuda)).sum().backward()
        return y.detach(), {**{k: dev[k].grad.detach() for k in keys}, "x": xd.grad.detach(), "skip": sk.grad.detach()}
    y_f, g_f = run(True)
    assert calls == ["fwd", "bwd"]
    y_f2, g_f2 = run(True)
    assert torch.equal(y_f, y_f2) and all(torch.equal(g_f[k], g_f2[k]) for k in g_f)              # deterministic
    y_g, g_g = run(False)
    assert calls == ["fwd", "bwd"] * 2                                                           # generic route: not called
    tag = f"dec7_fused[{t}-{dtype}]"
    assert record(tag + ".fwd_vs_generic", rel_l2(y_f, y_g)) < tol
    for k in g_f:
        assert record(tag + ".vs_generic.d" + k, rel_l2(g_f[k], g_g[k])) < btol, k
    # oracle for the fused layer alone, on the input the kernels saw: u = round(relu(layer 0) + skip) taken from the
    # generic route's arithmetic (rebuilt here in f64 from rounded operands)
    ref = {k: (_rounded(v, dtype) if k.endswith("weight") else v.double()).requires_grad_(True) for k, v in sd.items()}
    with torch.no_grad():
        u = R.decoder_layer(ref, 0, _rounded(x, dtype), last=False, store=_store(dtype)) + _rounded(skip, dtype)
    ur = _rounded(u.float(), dtype).requires_grad_(True)
    yr = R.decoder_layer(ref, 1, ur, last=True, store=_store(dtype))
    (yr * dout.double()).sum().backward()
    assert record(tag + ".fwd", rel_l2(y_f, yr)) < tol
    assert record(tag + ".dskip", rel_l2(g_f["skip"], ur.grad)) < btol
    for k in keys[4:]:
        assert record(tag + ".d" + k, rel_l2(g_f[k], ref[k].grad)) < btol, k
