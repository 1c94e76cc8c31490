"""BASELINE config 3 at its OWN batch: every distinct cum_gemm_nt / cum_gemm_tn call of one E8 train step with 16 clips
of 10 s (the benched configuration) is recorded from a real eager step, then re-issued through the C ABI on seeded
operands of exactly that descriptor (M, N, K, strides, pitch / valid rows, epilogue, its optional operands and flags) and
compared with an f64 product of the same rounded operands.  At B = 16 the launcher's tile map differs from the B <= 3
shapes of test_convstack_gpu.py: the deep layers (enc3..enc6, dec1..dec4) take the 256 x 256 ping-pong kernels
(gemm_nt9_kernel / gemm_tn9_kernel), which this file pins -- the library itself reports the tile (cum_gemm_nt_tile /
cum_gemm_tn_tile), so the statement "verified on the 256 x 256 kernel" is the launcher's, not a re-derivation.

Reference: the layers behind these GEMMs are src/network/CleanUMamba.py:108-113 (encoder), 121-130 (decoder), 139 / 194
(1x1 around the bottleneck) and the four projections of upstream Mamba.forward (called at :288-290)."""
import ctypes
import os

import pytest
import torch

from conftest import record, rel_l2

pytestmark = pytest.mark.gpu

E8 = dict(channels_input=1, channels_output=1, channels_H=64, max_H=768, encoder_n_layers=8, kernel_size=4,
          stride=2, tsfm_n_layers=3, tsfm_n_head=8, tsfm_d_model=512, tsfm_d_inner=2048)
B, CLIP = 16, 160000
# output rounding of one GEMM whose f32 accumulator is exact up to summation order (as test_convstack_gpu.py)
NT_TOL = {torch.float16: 8e-4, torch.bfloat16: 6e-3}
TN_TOL = 2e-6


def _record_step(cuda, dtype):
    """One eager E8 step at B = 16 with cs.gemm / cs.wgrad intercepted: -> (nt descriptors, tn descriptors), distinct,
    in first-seen order, each with its call count."""
    from cleanumamba_amd.network import Net
    from cleanumamba_amd.network import convstack as cs
    from cleanumamba_amd.training.train_step import TrainStep
    torch.manual_seed(0)
    net = Net("CleanUMamba", E8).to(cuda).train()
    step = TrainStep(net, autocast_dtype=dtype, use_graph=False)
    g = torch.Generator(device=cuda).manual_seed(1234)
    clean = 0.05 * torch.randn(B, 1, CLIP, generator=g, device=cuda)
    noisy = clean + 0.05 * torch.randn(B, 1, CLIP, generator=g, device=cuda)
    step(clean, noisy)                       # first step: un-batched packs, caches
    nt, tn = {}, {}
    real_gemm, real_wgrad = cs.gemm, cs.wgrad

    def gemm(A, a_off, lda, Wp, bias, out, o_off, ldc, M, pitch, valid, epilogue, n_store, res=None, r_off=0, ldr=0,
             aux=None, x_off=0, ldz=0, geo=None, aux2=None, y_off=0, ldy=0, gate_only=False, mask_bits=False,
             split_k=False):
        key = (epilogue, M, Wp.shape[0], Wp.shape[1], lda, ldc, min(pitch, 1 << 30), min(valid, 1 << 30), n_store,
               res is not None, ldr if res is not None else 0, aux is not None, ldz if aux is not None else 0,
               aux2 is not None, ldy if aux2 is not None else 0, bool(gate_only), bool(mask_bits), bool(split_k),
               bias is not None, (geo.head, geo.tail) if geo is not None else None, a_off, o_off)
        nt[key] = nt.get(key, 0) + 1
        return real_gemm(A, a_off, lda, Wp, bias, out, o_off, ldc, M, pitch, valid, epilogue, n_store, res=res,
                         r_off=r_off, ldr=ldr, aux=aux, x_off=x_off, ldz=ldz, geo=geo, aux2=aux2, y_off=y_off, ldy=ldy,
                         gate_only=gate_only, mask_bits=mask_bits, split_k=split_k)

    def wgrad(dZ, z_off, ldz, N, X, x_off, ldx, K, M, want_bias=True, **kw):
        key = (M, N, K, ldz, ldx, bool(want_bias))
        tn[key] = tn.get(key, 0) + 1
        return real_wgrad(dZ, z_off, ldz, N, X, x_off, ldx, K, M, want_bias=want_bias, **kw)

    cs.gemm, cs.wgrad = gemm, wgrad
    try:
        step(clean, noisy)
    finally:
        cs.gemm, cs.wgrad = real_gemm, real_wgrad
    torch.cuda.synchronize()
    del step, net
    torch.cuda.empty_cache()
    return nt, tn


def _rows(flat, off, M, ld, n):
    return torch.as_strided(flat, (M, n), (ld, 1), off)


def _nibbles(pos):
    """pos [M, n] bool -> one byte per four consecutive columns (low nibble), as the kernels store signs."""
    q = pos.view(pos.shape[0], -1, 4).to(torch.uint8)
    return q[..., 0] | q[..., 1] << 1 | q[..., 2] << 2 | q[..., 3] << 3


def _check_nt(cuda, dtype, key, tag):
    """Re-issue one recorded cum_gemm_nt descriptor on seeded operands; compare out (and aux) with f64."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.network import convstack as cs
    (epi, M, N, K, lda, ldc, pitch, valid, n_store, has_res, ldr, has_aux, ldz, has_aux2, ldy, gate_only, mask_bits,
     split_k, has_bias, frame, a_off, o_off) = key
    tol = NT_TOL[dtype]
    g = torch.Generator(device=cuda).manual_seed(hash((epi, M, N, K)) % (2 ** 31))
    rn = lambda *s: torch.randn(*s, generator=g, device=cuda)
    A = rn(a_off + (M - 1) * lda + K + 64).to(dtype)
    W = (rn(N, K) / K ** 0.5).to(dtype)
    bias = 0.1 * rn(N) if has_bias else None
    head, tail = frame if frame is not None else (0, 0)
    assert o_off >= head
    n_out = 2 * n_store if epi == hip.EPI_GLU_BWD else n_store         # columns cum_gemm_nt writes per row
    out = torch.full((o_off + M * ldc + tail + 64,), 7.0, device=cuda, dtype=dtype)
    m = torch.arange(M, device=cuda)
    real = ((m % pitch) < valid)[:, None]
    res = aux = aux2 = None
    r_off = x_off = y_off = 0
    geo = None
    if frame is not None:
        geo = type("G", (), {"head": head, "tail": tail})()
    sign = None
    if has_res:
        if epi == hip.EPI_MASK and mask_bits:
            sign = torch.rand(M, ldr, generator=g, device=cuda) > 0.5
            res = _nibbles(sign).reshape(-1).contiguous()
        else:
            res = rn(M * ldr + 64).to(dtype)
    if has_aux:
        if epi == hip.EPI_GLU_BWD:
            aux = rn(M * ldz + 64).to(dtype)                            # Z (a | b packed) or the gate b: an INPUT
        elif mask_bits and epi == hip.EPI_RELU:                         # (MASK: mask_bits describes res, aux is a full output)
            aux = torch.full((M * ldz // 4 + 64,), 255, dtype=torch.uint8, device=cuda)
        else:
            aux = torch.full((o_off + M * ldz + tail + 64,), 7.0, device=cuda, dtype=dtype)
            x_off = o_off if (epi != hip.EPI_GLU and frame is not None) else 0
    if has_aux2:
        aux2 = rn(M * ldy + 64).to(dtype)
    cs.gemm(A, a_off, lda, W, bias, out, o_off, ldc, M, pitch, valid, epi, n_store, res=res, r_off=r_off, ldr=ldr,
            aux=aux, x_off=x_off, ldz=ldz, geo=geo, aux2=aux2, y_off=y_off, ldy=ldy, gate_only=gate_only,
            mask_bits=mask_bits, split_k=split_k)
    torch.cuda.synchronize()

    # ---- f64 reference, row chunks
    got_out = _rows(out, o_off, M, ldc, n_out).double()
    want_out = torch.empty(M, n_out, dtype=torch.float64, device=cuda)
    want_aux = None
    Wd = W.double()
    bd = bias.double() if bias is not None else torch.zeros(N, dtype=torch.float64, device=cuda)
    CH = 32768
    n16 = N // 16
    for lo in range(0, M, CH):
        hi = min(M, lo + CH)
        acc = _rows(A, a_off + lo * lda, hi - lo, lda, K).double() @ Wd.t()
        rl = real[lo:hi]
        if epi in (hip.EPI_BIAS, hip.EPI_RELU):
            v = acc + bd
            if epi == hip.EPI_RELU:
                v = v.clamp_min(0)
            v = torch.where(rl, v, torch.zeros_like(v))
            if has_aux:
                if want_aux is None:
                    want_aux = torch.empty(M, n_store, dtype=torch.float64, device=cuda)
                want_aux[lo:hi] = v[:, :n_store]
            if has_res:
                v = torch.where(rl, v + _rows(res, r_off + lo * ldr, hi - lo, ldr, N).double(), torch.zeros_like(v))
            want_out[lo:hi] = v[:, :n_store]
        elif epi == hip.EPI_GLU:
            z = (acc + bd).view(hi - lo, N // 32, 2, 16)
            a, b = z[:, :, 0].reshape(hi - lo, N // 2), z[:, :, 1].reshape(hi - lo, N // 2)
            o = torch.where(rl, a * torch.sigmoid(b), torch.zeros_like(a))
            if has_aux:
                assert gate_only, "the train step saves the gate-only form"
                if want_aux is None:
                    want_aux = torch.empty(M, N // 2, dtype=torch.float64, device=cuda)
                want_aux[lo:hi] = b
            if has_res:
                o = torch.where(rl, o + _rows(res, r_off + lo * ldr, hi - lo, ldr, N // 2).double(), torch.zeros_like(o))
            want_out[lo:hi] = o[:, :n_store]
        elif epi == hip.EPI_MASK:
            v = torch.where(rl, acc + bd, torch.zeros_like(acc))
            if has_aux:
                if want_aux is None:
                    want_aux = torch.empty(M, n_store, dtype=torch.float64, device=cuda)
                want_aux[lo:hi] = v[:, :n_store]
            gate = sign[lo:hi, :N] if sign is not None else _rows(res, r_off + lo * ldr, hi - lo, ldr, N) > 0
            want_out[lo:hi] = torch.where(gate, v, torch.zeros_like(v))[:, :n_store]
        else:                                                            # GLU_BWD
            d = acc[:, :n_store]
            if has_res:
                d = d + _rows(res, r_off + lo * ldr, hi - lo, ldr, n_store).double()
            d = torch.where(rl, d, torch.zeros_like(d))
            if gate_only:
                bg = _rows(aux, lo * ldz, hi - lo, ldz, n_store).double()
                y = _rows(aux2, lo * ldy, hi - lo, ldy, n_store).double()
                sg = torch.sigmoid(bg)
                da, db = d * sg, d * y * (1 - sg)
            else:
                zz = _rows(aux, lo * ldz, hi - lo, ldz, 2 * n_store).double().view(hi - lo, n_store // 16, 2, 16)
                a, bg = zz[:, :, 0].reshape(hi - lo, n_store), zz[:, :, 1].reshape(hi - lo, n_store)
                sg = torch.sigmoid(bg)
                da, db = d * sg, d * a * sg * (1 - sg)
            w = want_out[lo:hi].view(hi - lo, n_store // 16, 2, 16)
            w[:, :, 0], w[:, :, 1] = da.view(hi - lo, -1, 16), db.view(hi - lo, -1, 16)
        del acc
    assert n16 > 0
    err = rel_l2(got_out, want_out)
    assert record(tag + ".out", err) < tol, (tag, key, err)
    # rows outside a clip are written as zeros, never left untouched (the next GEMM reads them as padding)
    if pitch < (1 << 30):
        dead = ~real[:, 0]
        assert float(got_out[dead].abs().max()) == 0.0, tag
    if frame is not None:
        assert float(out[o_off - head:o_off].float().abs().max()) == 0 and \
            float(out[o_off + M * ldc:o_off + M * ldc + tail].float().abs().max()) == 0, tag + ": framing rows"
    if has_aux and epi != hip.EPI_GLU_BWD:
        if mask_bits and epi == hip.EPI_RELU:
            got = aux[:M * ldz // 4].view(M, ldz // 4)[:, :n_store // 4]
            want = _nibbles(want_aux > 0)
            # a sign may differ only where the f32 sum is within rounding of zero
            bad = got != want
            if bool(bad.any()):
                mag = want_aux.abs().view(M, n_store // 4, 4).min(-1).values
                assert float(mag[bad].max()) < 1e-4 * float(want_aux.abs().mean()), tag + ": sign nibbles"
        else:
            live = real[:, 0]
            got_aux = _rows(aux, x_off, M, ldz, want_aux.shape[1]).double()
            assert record(tag + ".aux", rel_l2(got_aux[live], want_aux[live])) < tol, tag + ": aux"


def _check_tn(cuda, dtype, key, tag):
    from cleanumamba_amd.network import convstack as cs
    M, N, K, ldz, ldx, want_bias = key
    g = torch.Generator(device=cuda).manual_seed((M * 31 + N * 7 + K) % (2 ** 31))
    dz = torch.randn(M * ldz + 64, generator=g, device=cuda).to(dtype)
    x = torch.randn((M - 1) * ldx + K + 64, generator=g, device=cuda).to(dtype)
    dw, db = cs.wgrad(dz, 0, ldz, N, x, 0, ldx, K, M, want_bias=want_bias)
    torch.cuda.synchronize()
    want = torch.zeros(N, K, dtype=torch.float64, device=cuda)
    CH = 65536
    for lo in range(0, M, CH):
        hi = min(M, lo + CH)
        want += _rows(dz, lo * ldz, hi - lo, ldz, N).double().t() @ _rows(x, lo * ldx, hi - lo, ldx, K).double()
    assert record(tag + ".dW", rel_l2(dw, want)) < TN_TOL, (tag, key)
    if want_bias:
        assert record(tag + ".db", rel_l2(db, _rows(dz, 0, M, ldz, N).double().sum(0))) < TN_TOL, (tag, key)


@pytest.mark.parametrize("dtype", [torch.float16])
def test_every_gemm_of_the_b16_step_against_f64(cuda, dtype):
    from cleanumamba_amd import hip
    nt, tn = _record_step(cuda, dtype)
    lib = hip.lib()
    dc = hip.dtype_code(dtype)

    def nt_tile(key):
        d = hip.GemmDesc()
        d.dtype, d.M, d.N, d.K, d.allow_split_k = dc, key[1], key[2], key[3], 2 if key[17] else 0
        return lib.cum_gemm_nt_tile(ctypes.byref(d))
    assert sum(nt.values()) >= 80 and sum(tn.values()) >= 40, (sum(nt.values()), sum(tn.values()))
    # ---- the dispatch map of the benched configuration, as the library reports it
    tiles = {k: nt_tile(k) for k in nt}
    on9 = sorted({(k[0], k[1], k[2], k[3]) for k, t in tiles.items() if t == 512})
    rows = {16 * (t + 2): name for name, t in (("enc3/dec4", 10014), ("enc4/dec3", 5006), ("enc5/dec2", 2502),
                                               ("enc6/dec1", 1250))}
    for M, name in rows.items():
        epis = {e for (e, m, _, _) in on9 if m == M}
        # forward conv + ReLU, forward 1x1 + GLU, the gated and the GLU-backward data gradients of the deep layers
        assert {hip.EPI_RELU, hip.EPI_GLU, hip.EPI_MASK, hip.EPI_GLU_BWD} <= epis, (name, M, epis, on9)
    tn_tiles = {k: lib.cum_gemm_tn_tile(dc, k[0], k[1], k[2]) for k in tn}
    for M in rows:
        assert any(t == 256 for k, t in tn_tiles.items() if k[0] == M), M
    assert sum(t == 384 for t in tn_tiles.values()) >= 2, tn_tiles          # enc1 / dec6: the streaming kernel
    assert any(t == 128 for t in tiles.values()) and any(t == 64 for t in tiles.values())
    # the M ~ 10 000 launches with N = 512 / 768 and a long K axis: the 128 x 256 ring kernel (tile id 384), all epilogues they use
    ring = sorted({(k[0], k[2], k[3]) for k, t in tiles.items() if t == 384})
    assert {e for e, _, _ in ring} >= {hip.EPI_BIAS, hip.EPI_RELU, hip.EPI_GLU_BWD} and (0, 512, 4096) in ring, ring
    # ---- every distinct call against f64
    for i, key in enumerate(nt):
        _check_nt(cuda, dtype, key, f"b16.nt[{i}:epi{key[0]}:{key[1]}x{key[2]}x{key[3]}:tile{tiles[key]}]")
        torch.cuda.empty_cache()
    for i, key in enumerate(tn):
        _check_tn(cuda, dtype, key, f"b16.tn[{i}:{key[0]}x{key[1]}x{key[2]}:tile{tn_tiles[key]}]")
        torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(641024, 128, 256, 128, 128, True), (641024, 256, 128, 256, 128, True),
                                   (131072 + 64 * 3 + 17, 96, 200, 96, 200, True), (200003, 256, 72, 256, 72, False),
                                   (150001, 128, 256, 136, 264, False)])
def test_streaming_weight_gradient_kernel(cuda, dtype, shape):
    """gemm_tn_stream_kernel (the 128 x 256 / 256 x 128 results of the two widest layers): the E8 B = 16 shapes (the conv's
    overlapping rows: ldx = K / 2), ragged row counts (the last split ends inside a 64-row step), column counts that end
    inside the padded result, with and without the bias gradient -- against f64."""
    from cleanumamba_amd import hip
    M, N, K = shape[:3]
    assert hip.lib().cum_gemm_tn_tile(hip.dtype_code(dtype), M, N, K) == 384
    _check_tn(cuda, dtype, shape, f"tn_stream[{dtype}:{M}x{N}x{K}:ldx{shape[4]}]")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("kernel", ["nt9", "ring"])
def test_ping_pong_kernel_ragged_rows_every_epilogue(cuda, dtype, epi, kernel):
    """gemm_nt9_kernel (256 x 256 ping-pong; tile id 512) and gemm_nt_ring_kernel (128 x 256 through the three-stage ring; tile
    id 384) on a row count that ends inside a tile (and inside a 64-row wave slab), every epilogue with its optional operands:
    residual, pre-activation copies, sign nibbles both ways, gate-only GLU forms."""
    from cleanumamba_amd import hip
    M, N, K = (60000 + 77, 768, 1536) if kernel == "nt9" else (10000 + 77, 768, 1536)
    d = hip.GemmDesc()
    d.dtype, d.M, d.N, d.K = hip.dtype_code(dtype), M, N, K
    assert hip.lib().cum_gemm_nt_tile(ctypes.byref(d)) == (512 if kernel == "nt9" else 384)
    n_store = N // 2 if epi == hip.EPI_GLU else N
    cases = {
        0: [(True, N, False, 0, False, 0, False, False)],
        1: [(True, N, True, N, False, 0, False, False), (False, 0, True, N, False, 0, False, True)],
        2: [(False, 0, True, N // 2, False, 0, True, False), (True, N // 2, False, 0, False, 0, False, False)],
        3: [(True, N, True, N, False, 0, False, False), (True, N, False, 0, False, 0, False, True)],
        4: [(True, N, True, 2 * N, False, 0, False, False), (True, N, True, N, True, N, True, False)],
    }[epi]
    for ci, (has_res, ldr, has_aux, ldz, has_aux2, ldy, gate_only, mask_bits) in enumerate(cases):
        ldc = 2 * N if epi == hip.EPI_GLU_BWD else n_store
        key = (epi, M, N, K, K, ldc, 5003, 5001, n_store, has_res, ldr, has_aux, ldz, has_aux2, ldy, gate_only, mask_bits,
               False, epi != hip.EPI_GLU_BWD, None, 0, 0)
        _check_nt(cuda, dtype, key, f"{kernel}.ragged[{dtype}:epi{epi}:{ci}]")


def _e6_net(cuda):
    from conftest import golden_json, load_golden
    from oracle import synth
    from cleanumamba_amd.network import CleanUMamba
    g = load_golden("e2e_e6_synth")
    meta = golden_json(g["meta"])
    sd = synth.fill_state_dict(dict(zip(meta["keys"], meta["shapes"])), seed=meta["seed"])
    net = CleanUMamba(**meta["cfg"])
    net.load_state_dict(sd, strict=True)
    return net.to(cuda).eval(), sd


def test_every_gemm_of_the_e6_b32_forward(cuda):
    """BASELINE config 2 at its OWN batch: the E6 (27.2 M) no-grad forward on 32 clips of 10 s under f16 autocast (M =
    32 x 80 032 ... 32 x 2 501 rows, other tile choices than the B = 16 E8 step): every distinct cum_gemm_nt call is
    recorded, re-issued through the C ABI on seeded operands and compared with an f64 product; the library reports
    which launches run on the 256 x 256 ping-pong kernel."""
    from cleanumamba_amd import hip
    from cleanumamba_amd.network import convstack as cs
    dtype = torch.float16
    net, _ = _e6_net(cuda)
    g = torch.Generator(device=cuda).manual_seed(4321)
    noisy = 0.1 * torch.randn(32, 1, CLIP, generator=g, device=cuda)
    nt = {}
    real_gemm = cs.gemm

    def gemm(A, a_off, lda, Wp, bias, out, o_off, ldc, M, pitch, valid, epilogue, n_store, res=None, r_off=0, ldr=0,
             aux=None, x_off=0, ldz=0, geo=None, aux2=None, y_off=0, ldy=0, gate_only=False, mask_bits=False,
             split_k=False):
        key = (epilogue, M, Wp.shape[0], Wp.shape[1], lda, ldc, min(pitch, 1 << 30), min(valid, 1 << 30), n_store,
               res is not None, ldr if res is not None else 0, aux is not None, ldz if aux is not None else 0,
               aux2 is not None, ldy if aux2 is not None else 0, bool(gate_only), bool(mask_bits), bool(split_k),
               bias is not None, (geo.head, geo.tail) if geo is not None else None, a_off, o_off)
        nt[key] = nt.get(key, 0) + 1
        return real_gemm(A, a_off, lda, Wp, bias, out, o_off, ldc, M, pitch, valid, epilogue, n_store, res=res,
                         r_off=r_off, ldr=ldr, aux=aux, x_off=x_off, ldz=ldz, geo=geo, aux2=aux2, y_off=y_off, ldy=ldy,
                         gate_only=gate_only, mask_bits=mask_bits, split_k=split_k)

    with torch.no_grad(), torch.autocast("cuda", dtype=dtype):
        net(noisy)                                # first call: packs, caches
        cs.gemm = gemm
        try:
            y = net(noisy)
        finally:
            cs.gemm = real_gemm
    torch.cuda.synchronize()
    assert y.shape == (32, 1, CLIP) and bool(torch.isfinite(y).all())
    del net, y, noisy
    torch.cuda.empty_cache()
    lib, dc = hip.lib(), hip.dtype_code(dtype)

    def nt_tile(key):
        d = hip.GemmDesc()
        d.dtype, d.M, d.N, d.K, d.allow_split_k = dc, key[1], key[2], key[3], 2 if key[17] else 0
        return lib.cum_gemm_nt_tile(ctypes.byref(d))
    tiles = {k: nt_tile(k) for k in nt}
    # 4 unfused encoder layers x 2 + 5 unfused decoder layers x 2 + the 1x1 convs + 3 blocks x 4 projections
    assert sum(nt.values()) >= 30, sum(nt.values())
    on9 = {(k[0], k[1]) for k, t in tiles.items() if t == 512}
    for T in (10002, 5000, 2499):                 # enc3 / enc4 / enc5 (and their decoder mirrors): 768-wide layers
        assert any(m == 32 * (T + 2) for _, m in on9), (T, sorted(on9))
    for i, key in enumerate(nt):
        _check_nt(cuda, dtype, key, f"e6b32.nt[{i}:epi{key[0]}:{key[1]}x{key[2]}x{key[3]}:tile{tiles[key]}]")
        torch.cuda.empty_cache()


def test_e6_b32_f32_forward_against_the_oracle_on_two_clips(cuda):
    """The same configuration in f32 against oracle.cleanumamba_ref.forward_ref (CPU) on the first and the last clip of
    the batch: north_star's 1e-4 at BASELINE config 2's own batch and length."""
    from oracle import cleanumamba_ref as R
    net, sd = _e6_net(cuda)
    g = torch.Generator().manual_seed(77)
    noisy = 0.1 * torch.randn(32, 1, CLIP, generator=g)
    with torch.no_grad():
        y = net(noisy.to(cuda)).cpu()
        pick = [0, 31]
        ref = R.forward_ref(sd, noisy[pick])
    err = record("e6b32.f32_forward_vs_oracle", rel_l2(y[pick], ref))
    assert err < 1e-4
    assert rel_l2(y[31:32], ref[1:2]) < 1e-4


def test_e8_b16_step_against_the_oracle_on_two_clips(cuda):
    """BASELINE config 3 at its OWN batch and length, end to end against the CPU oracle (the call bench.py's
    cpu_baseline times, used here as the checker): E8, 16 clips of 10 s, f32.  Forward of the whole batch on the HIP
    path, clips 0 and 15 against oracle.cleanumamba_ref.forward_ref (north_star's 1e-4); the training loss (L1 +
    multi-resolution STFT, the reference's configs/config.json terms) of those two clips against oracle loss_ref; then the
    backward of that loss through the B = 16 network (the fourteen other clips carry a zero cotangent, every backward
    kernel runs at its benched size) against the oracle's backward on the two clips: the last decoder layer's weight
    gradient (behind no ReLU) to 1e-5, every parameter's gradient norm to 2 % (ReLU-gate flips between two f32
    implementations, tests/test_model_gpu.py::test_full_width_model_forward_and_gradients).
    Reference: src/network/CleanUMamba.py:252-324, src/util/util.py:215-327, src/training/train.py:278-285."""
    from oracle import cleanumamba_ref as R
    from oracle import synth
    from cleanumamba_amd.network import CleanUMamba
    from cleanumamba_amd.util.stft_loss import MultiResolutionSTFTLoss
    from cleanumamba_amd.util.util import loss_fn
    stft = {"sc_lambda": 0.5, "mag_lambda": 0.5, "band": "full", "hop_sizes": [50, 120, 240],
            "win_lengths": [240, 600, 1200], "fft_sizes": [512, 1024, 2048]}
    torch.manual_seed(0)
    net = CleanUMamba(**E8)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    net = net.to(cuda).train()
    clean, noisy = synth.waveform(B, CLIP, seed=1234)
    pick = [0, B - 1]
    y = net(noisy.to(cuda))
    assert y.shape == (B, 1, CLIP)
    mr = MultiResolutionSTFTLoss(**stft).to(cuda)
    ysel = y[pick]
    loss, _ = loss_fn(lambda x: ysel, (clean[pick].to(cuda), noisy[pick].to(cuda)), mrstftloss=mr)
    loss.backward()
    torch.cuda.synchronize()
    # the oracle on the same two clips (CPU; about 15 s on the box's cores)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    sdr = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    yr = R.forward_ref(sdr, noisy[pick])
    lr = R.loss_ref(yr, clean[pick], stft_config=stft)
    lr.backward()
    err = record("e8b16.f32_forward_vs_oracle", rel_l2(y[pick].detach(), yr.detach()))
    assert err < 1e-4
    assert rel_l2(y[B - 1:B].detach(), yr[1:2].detach()) < 1e-4
    lerr = record("e8b16.loss_vs_oracle", abs(loss.item() - lr.item()) / abs(lr.item()))
    assert lerr < 1e-4
    named = dict(net.named_parameters())
    last = f"decoder.{E8['encoder_n_layers'] - 1}.2.weight"
    gerr = record("e8b16.last_layer_wgrad_vs_oracle", rel_l2(named[last].grad, sdr[last].grad))
    assert gerr < 1e-5
    worst = 0.0
    for k, p in named.items():
        gr = sdr[k].grad
        assert p.grad is not None and gr is not None, k
        a, b = p.grad.double().norm().item(), gr.double().norm().item()
        worst = max(worst, abs(a - b) / max(b, 1e-30))
        assert abs(a - b) < 2e-2 * b, (k, a, b)
    record("e8b16.worst_gradnorm_dev_vs_oracle", worst)
    tot_a = sum((p.grad.double() ** 2).sum().item() for p in named.values()) ** 0.5
    tot_b = sum((sdr[k].grad.double() ** 2).sum().item() for k in named) ** 0.5
    assert abs(tot_a - tot_b) < 2e-2 * tot_b
