"""Host side of the one-launch streaming hop (csrc/hop.hip, ``cum_stream_hop``).

Builds, from a CleanUMamba module, what the kernel reads: the int32 plan (sizes, LDS regions, offsets), the f32 weight
blob (every matrix zero-padded and in MFMA fragment order) and the per-stream state blocks; converts the stream state
of the per-layer path (which runs the FIRST frame of a stream: whole windows, no history) into the kernel's layout and
back (for ``flush``'s drain).

Reference semantics: CleanUMamba.feed / _denoise_frame, src/network/CleanUMamba.py:370-490 (with the skip order and the
flush fixed as SURVEY fact 9 describes); Mamba.step, mamba-ssm 1.2.2.
"""
import struct

import numpy as np
import torch
import torch.nn as nn

from .. import hip
from . import convstack as cs

# op list of csrc/hop.hip (kind, then the fields in the order the kernel reads them; 24 ints per op)
_OP_END, _OP_STD, _OP_ENC0, _OP_GEMM, _OP_RING, _OP_LN, _OP_CONVSTEP, _OP_SSM, _OP_OVERLAP = range(9)
_ACT_NONE, _ACT_RELU, _ACT_SOFTPLUS = 0, 1, 2
_HDR_INTS, _OP_INTS, _MAX_OPS, _MAX_LAYERS, _MAX_BLOCKS = 16, 24, 160, 12, 8
_MAX_STAGES, _MAX_WAVE_STAGES = 8192, 64      # csrc/hop.hip::kHopMaxStages; a wave's list of one op is one 64-lane load
_MAGIC = 0x486f7033
_BIG = 1 << 30
_WAVES = 8           # waves of the kernel's workgroup (csrc/hop.hip::kHopWaves)
# every stream re-reads all weights from L2 once per hop: above this the per-layer path (streams batched as GEMM rows)
# is the right design
MAX_PARAMS = 3_000_000


def _rup(x, m):
    return (x + m - 1) // m * m


def _fbits(x):
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


def _frag(mat):
    """[N, K] -> MFMA A-operand fragment order [N/16][K/16][lane = 16 (k / 4 % 4) + n % 16][k % 4], zero-padded."""
    N, K = mat.shape
    Np, Kp = _rup(N, 16), _rup(K, 16)
    full = torch.zeros(Np, Kp, dtype=torch.float32, device=mat.device)
    full[:N, :K] = mat
    return full.view(Np // 16, 16, Kp // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous().reshape(-1)


def _pad1(v, n):
    out = torch.zeros(n, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v.reshape(-1)
    return out


def _pad2(m, rows, cols):
    out = torch.zeros(rows, cols, dtype=torch.float32, device=m.device)
    out[:m.shape[0], :m.shape[1]] = m
    return out


class _Blob:
    def __init__(self):
        self.parts, self.size = [], 0

    def add(self, t):
        off = self.size
        t = t.detach().float().reshape(-1)
        pad = _rup(t.numel(), 4) - t.numel()
        self.parts.append(t if pad == 0 else torch.cat([t, t.new_zeros(pad)]))
        self.size += t.numel() + pad
        return off


def _split(ntg, kcn, M, nacc, cap, kpr):
    """(mt, ks) for one matrix product of the hop: 16-row tiles per work item and k slices, by pricing the stage lists the
    candidates compile to (``_stages``) with constants fitted to the per-op stamps of tools/hop_phase_probe.py (shader
    cycles): an op costs ~3 k whatever it does; a wave has ONE stage in flight, so a stage takes the longer of the L2
    latency of its fragments (~1.3 k) and its MFMAs (32 cycles each; two waves share a SIMD's matrix pipe once more than
    four waves have work); closing an item ~0.4 k; a split product pays a barrier and the combine pass.  The product's
    time is its busiest wave's."""
    best = None
    mt_max = 1 if M <= 16 else 2 if M <= 32 else 4
    for mt in (1, 2, 4):
        if mt > mt_max:
            continue
        base = ntg * ((M + 16 * mt - 1) // (16 * mt))
        for ks in range(1, min(kcn, 8) + 1):
            if ks > 1 and ks * base * nacc * mt * 256 > cap:
                break
            kcs = (kcn + ks - 1) // ks
            if ks > 1 and (ks - 1) * kcs >= kcn:
                continue                                  # an empty last slice
            waves = _stages([_OP_GEMM, 0, 0, 0, ntg, kcn, 0, kpr, 0, M, cap, 0, -1, -1, -1, 0, 0, 0, 0, nacc, ks, kcs, mt])
            per_simd = 2 if sum(1 for lst in waves if lst) > 4 else 1
            cost = 3000 + max(sum(max(1300, per_simd * (t[2] & 7) * mt * nacc * 128 + 200) + 400 * (t[2] >> 4 & 1)
                                  for t in lst) for lst in waves)
            if ks > 1:
                cost += 1200 + base * mt * 64 / 512 * (60 + 15 * ks)
            if best is None or cost < best[0]:
                best = (cost, mt, ks)
    return best[1], best[2]


def _stages(op):
    """Per-wave stage lists of one matrix product, as csrc/hop.hip::hop_gemm walks them: [[(weight offset, operand offset,
    meta, out)] per wave].  Work items (tile group fastest, then group of ``mt`` 16-row tiles, then k slice) are dealt to
    the waves in contiguous runs; an item's k range is cut into stages of <= 4 chunks that stay inside one k segment
    (the 4 input rows of a strided conv are 4 segments of the same LDS image), walked in ascending k: the order of the
    fma chain -- and so every bit of the result -- is the k order, whatever the split.
    meta = chunks | first stage of the item << 3 | last << 4;  out of a last stage = first output channel | first row
    << 16 (unsplit product: the epilogue runs in place) or the LDS offset of the slice's partial sums (k split)."""
    (_, w, x, scratch, ntg, kcn, xs, kpr, seg, M, cap, dst, bias, bias2, add, pitch, row_off, act, nlimit, nacc, ks, kcs,
     mt) = op[:23]
    mgs = (M + 16 * mt - 1) // (16 * mt)
    base = ntg * mgs
    items = base * ks
    ipw = (items + _WAVES - 1) // _WAVES
    blk = nacc * mt * 256
    waves = []
    for wave in range(_WAVES):
        lst = []
        for it in range(wave * ipw, min(items, wave * ipw + ipw)):
            tg, q = it % ntg, it // ntg
            mg, sl = q % mgs, q // mgs
            k, k1, first = sl * kcs, min(kcn, sl * kcs + kcs), 1
            while k < k1:
                sg = k // kpr
                e = min(k + 4, k1, (sg + 1) * kpr)
                last = int(e == k1)
                out = tg * 16
                if last:
                    out = (tg * 16) | ((mg * 16 * mt) << 16) if ks == 1 else scratch + ((sl * mgs + mg) * ntg + tg) * blk
                lst.append((w + (tg * nacc * kcn + k) * 256, x + mg * 16 * mt * xs + sg * seg + (k - sg * kpr) * 16,
                            (e - k) | first << 3 | last << 4, out))
                k, first = e, 0
        waves.append(lst)
    return waves


def unsupported_reason(model):
    """None if the one-launch hop can run this model, else why not."""
    E = model.encoder_n_layers
    if model.kernel_size != 4 or model.stride != 2:
        return "kernel_size / stride other than 4 / 2"
    if E > _MAX_LAYERS or E < 1 or len(model.tsfm_Mamba_layers) > _MAX_BLOCKS:
        return "too many layers"
    if model.encoder[0][0].weight.shape[1] != 1 or model.decoder[E - 1][2].weight.shape[1] != 1:
        return "more than one input / output channel"
    if sum(p.numel() for p in model.parameters()) > MAX_PARAMS:
        return "weights too large to be re-read per stream"
    if any(p.dtype != torch.float32 for p in model.parameters()):
        return "parameters are not f32"
    for blk in model.tsfm_Mamba_layers:
        m = blk.mixer
        if type(m).__name__ != "Mamba" or not isinstance(blk.norm, nn.LayerNorm) or not blk.norm.elementwise_affine:
            return "bottleneck block is not LayerNorm + Mamba"
        if m.activation not in ("silu", "swish") or m.conv1d.weight.shape[-1] > 8:
            return "Mamba variant outside the kernel"
    if not isinstance(model.norm_f, nn.LayerNorm) or not model.norm_f.elementwise_affine:
        return "norm_f is not an affine LayerNorm"
    return None


class HopPlan:
    """Plan + weight blob of one model (at one value of its weights), shared by all its streams."""

    def __init__(self, model):
        why = unsupported_reason(model)
        if why is not None:
            raise ValueError(why)
        dev = model.tsfm_conv1.weight.device
        E = model.encoder_n_layers
        self.E = E
        hop = model.total_stride
        frame_len = model.valid_length(1)
        blob = _Blob()
        ints = np.zeros(_HDR_INTS + _MAX_OPS * _OP_INTS + _MAX_OPS * _WAVES + _MAX_STAGES * 4, dtype=np.int32)
        if ints.size != hip.lib().cum_stream_hop_plan_ints():
            raise RuntimeError("hopplan.py and csrc/hop.hip disagree about the plan's size")
        state_off = 4                     # [std, frames seen, phase, -]
        hdr = {"magic": _MAGIC, "E": E, "n_blocks": len(model.tsfm_Mamba_layers), "frame_len": frame_len, "hop_len": hop,
               "normalize": int(bool(model.normalize_input)), "std_off": 0, "phase_off": 2}
        # LDS row pitches are the k extent + 4 floats (bank spread); R2 doubles as the scratch of k-split products
        r0, r1, r2 = 4, 4, 8192
        encs, decs, blks = [], [], []

        # ---------------- encoder
        n, ld_in, c_in = hop, 0, 1
        for i, enc in enumerate(model.encoder):
            n //= 2
            w1, b1, w2, b2 = enc[0].weight.detach().float(), enc[0].bias.detach().float(), \
                enc[2].weight.detach().float(), enc[2].bias.detach().float()
            H, C = w1.shape[0], w2.shape[0] // 2
            assert w1.shape[1] == c_in and w2.shape[1] == H and w1.shape[2] == 4
            ld_h, ld_out = _rup(H, 16), _rup(C, 16)
            e = {"n": n, "ld_in": ld_in, "ld_h": ld_h, "ld_out": ld_out}
            if i == 0:
                e["w1"] = blob.add(_pad2(w1[:, 0, :], ld_h, 4))
                e["nt1"] = e["kc1"] = 0
                r0 = max(r0, 2 * n + 2)
            else:
                mat = _pad2(w1.permute(0, 2, 1).reshape(H * 4, c_in), H * 4, ld_in).reshape(H, 4 * ld_in)
                e["w1"] = blob.add(_frag(mat))
                e["nt1"], e["kc1"] = ld_h // 16, 4 * ld_in // 16
                r0 = max(r0, (2 * n + 2) * (ld_in + 4))
            e["b1"] = blob.add(_pad1(b1, ld_h))
            wa, wb = _pad2(w2[:C, :, 0], ld_out, H), _pad2(w2[C:, :, 0], ld_out, H)
            pair = torch.stack([wa.view(ld_out // 16, 16, H), wb.view(ld_out // 16, 16, H)], 1).reshape(2 * ld_out, H)
            e["w2"] = blob.add(_frag(pair))
            e["b2"] = blob.add(torch.cat([_pad1(b2[:C], ld_out), _pad1(b2[C:], ld_out)]))
            e["ntg2"], e["kc2"] = ld_out // 16, ld_h // 16
            e["ring"] = state_off
            state_off += _rup(3 * n * ld_out, 4)
            r0 = max(r0, (n + 2) * (ld_out + 4))
            r1 = max(r1, n * (ld_h + 4))
            e["C"] = C
            encs.append(e)
            ld_in, c_in = ld_out, C
        if n != 1:
            raise ValueError("the deepest encoder layer must emit one row per hop")
        hdr["ld_last"] = ld_in

        # ---------------- bottleneck
        dm = model.tsfm_conv1.weight.shape[0]
        dmp = _rup(dm, 16)
        hdr.update(dm=dm, dmp=dmp)
        t1, t2 = model.tsfm_conv1, model.tsfm_conv2
        assert t1.weight.shape[1] == c_in and t2.weight.shape[0] == c_in and t2.weight.shape[1] == dm
        hdr["t1_w"] = blob.add(_frag(_pad2(t1.weight.detach().float()[:, :, 0], dmp, ld_in)))
        hdr["t1_b"] = blob.add(_pad1(t1.bias.detach().float() if t1.bias is not None else torch.zeros(dm, device=dev), dmp))
        hdr["t1_nt"], hdr["t1_kc"] = dmp // 16, ld_in // 16
        hdr["t2_w"] = blob.add(_frag(_pad2(t2.weight.detach().float()[:, :, 0], ld_in, dmp)))
        hdr["t2_b"] = blob.add(_pad1(t2.bias.detach().float() if t2.bias is not None else torch.zeros(c_in, device=dev), ld_in))
        hdr["t2_nt"], hdr["t2_kc"] = ld_in // 16, dmp // 16
        hdr["nf_w"] = blob.add(_pad1(model.norm_f.weight.detach().float(), dmp))
        hdr["nf_b"] = blob.add(_pad1(model.norm_f.bias.detach().float(), dmp))
        hdr["nf_eps"] = _fbits(model.norm_f.eps)
        rv_blk = 0
        for blk in model.tsfm_Mamba_layers:
            m = blk.mixer
            di, R = m.in_proj.weight.shape[0] // 2, m.dt_proj.weight.shape[1]
            N = (m.x_proj.weight.shape[0] - R) // 2
            W = m.conv1d.weight.shape[-1]
            assert m.in_proj.weight.shape[1] == dm and m.out_proj.weight.shape == (dm, di)
            dip, xdbp = _rup(di, 16), _rup(R + 2 * N, 16)
            b = {"di": di, "dip": dip, "N": N, "R": R, "W": W, "xdbp": xdbp}
            b["ln_w"] = blob.add(_pad1(blk.norm.weight.detach().float(), dmp))
            b["ln_b"] = blob.add(_pad1(blk.norm.bias.detach().float(), dmp))
            b["ln_eps"] = _fbits(blk.norm.eps)
            b["in_w"] = blob.add(_frag(_pad2(m.in_proj.weight.detach().float(), _rup(2 * di, 16), dmp)))
            b["in_b"] = -1 if m.in_proj.bias is None else blob.add(_pad1(m.in_proj.bias.detach().float(), _rup(2 * di, 16)))
            b["nt_in"], b["kc_in"] = _rup(2 * di, 16) // 16, dmp // 16
            b["conv_w"] = blob.add(m.conv1d.weight.detach().float().reshape(di, W))
            b["conv_b"] = -1 if m.conv1d.bias is None else blob.add(m.conv1d.bias.detach().float())
            b["xp_w"] = blob.add(_frag(_pad2(m.x_proj.weight.detach().float(), xdbp, dip)))
            b["nt_xp"], b["kc_xp"] = xdbp // 16, dip // 16
            b["dt_w"] = blob.add(_frag(_pad2(m.dt_proj.weight.detach().float(), dip, _rup(R, 16))))
            b["dt_b"] = -1 if m.dt_proj.bias is None else blob.add(_pad1(m.dt_proj.bias.detach().float(), dip))
            b["nt_dt"], b["kc_dt"] = dip // 16, _rup(R, 16) // 16
            b["A"] = blob.add(-torch.exp(m.A_log.detach().float()))
            b["D"] = blob.add(m.D.detach().float())
            b["out_w"] = blob.add(_frag(_pad2(m.out_proj.weight.detach().float(), dmp, dip)))
            b["out_b"] = -1 if m.out_proj.bias is None else blob.add(_pad1(m.out_proj.bias.detach().float(), dmp))
            b["nt_out"], b["kc_out"] = dmp // 16, dip // 16
            b["conv_state"] = state_off
            state_off += _rup(di * W, 4)
            b["ssm_state"] = state_off
            state_off += _rup(di * N, 4)
            rv_blk = max(rv_blk, 2 * dip + 16 + 3 * dip + xdbp)
            blks.append(b)
        rv = 3 * dmp + rv_blk + 16

        # ---------------- decoder
        L = 1
        for j, dec in enumerate(model.decoder):
            last = j == E - 1
            w1, b1, wt, bt = dec[0].weight.detach().float(), dec[0].bias.detach().float(), \
                dec[2].weight.detach().float(), dec[2].bias.detach().float()
            G, cout = w1.shape[0] // 2, wt.shape[1]
            assert w1.shape[1] == c_in and wt.shape[0] == G and wt.shape[2] == 4
            ld_g, cq = _rup(G, 16), _rup(cout, 4)
            ld_out = 16 if last else _rup(cout, 16)
            d = {"L": L, "ld_in": ld_in, "ld_g": ld_g, "cq": cq, "ld_out": ld_out, "relu": int(not last), "cout": cout}
            wa, wb = _pad2(w1[:G, :, 0], ld_g, ld_in), _pad2(w1[G:, :, 0], ld_g, ld_in)
            pair = torch.stack([wa.view(ld_g // 16, 16, ld_in), wb.view(ld_g // 16, 16, ld_in)], 1).reshape(2 * ld_g, ld_in)
            d["w1"] = blob.add(_frag(pair))
            d["b1"] = blob.add(torch.cat([_pad1(b1[:G], ld_g), _pad1(b1[G:], ld_g)]))
            d["ntg1"], d["kc1"] = ld_g // 16, ld_in // 16
            taps = torch.zeros(4, cq, ld_g, dtype=torch.float32, device=dev)
            taps[:, :cout, :G] = wt.permute(2, 1, 0)
            d["w2"] = blob.add(_frag(taps.reshape(4 * cq, ld_g)))
            d["nt2"], d["kc2"] = _rup(4 * cq, 16) // 16, ld_g // 16
            d["b2"] = blob.add(_pad1(bt, cq))
            d["tail"] = state_off
            state_off += _rup(2 * cq, 4)
            if last:
                d["skip_ring"], d["skip_ld"], d["skip_n"] = -1, 0, 0
            else:
                e = encs[E - 2 - j]
                if e["C"] != cout or e["n"] != 2 * L:
                    raise ValueError("decoder / encoder skip shapes disagree")
                d["skip_ring"], d["skip_ld"], d["skip_n"] = e["ring"], e["ld_out"], e["n"]
            r0 = max(r0, L * (ld_in + 4), 2 * L * (ld_out + 4))
            r1 = max(r1, L * (ld_g + 4))
            r2 = max(r2, L * 4 * cq)
            decs.append(d)
            L, ld_in, c_in = 2 * L, ld_out, cout
        if 2 * (L // 2) != hop:
            raise ValueError("decoder does not emit one hop of samples")

        # ---------------- LDS layout (float offsets) and the op list
        pad = 64                                  # 16-row MFMA tiles read (discarded) rows past a region's last row
        R0 = 0
        R1 = _rup(r0 + pad, 4)
        R2 = R1 + _rup(r1 + pad, 4)
        RV = R2 + _rup(r2 + pad, 4)
        cap0, cap2 = R1 - R0, RV - R2
        v_hs, v_res, v_h, misc = RV, RV + dmp, RV + 2 * dmp, RV + 3 * dmp
        ops = []

        def gemm(w, x, scratch, ntg, kcn, xs, kpr, seg, M, cap, dst, bias=-1, bias2=-1, add=-1, pitch=0, row_off=0,
                 act=_ACT_NONE, nlimit=_BIG, nacc=1, ring=-1):
            mt, ks = _split(ntg, kcn, M, nacc, cap, kpr)
            assert kcn <= 4 * kpr
            ops.append([_OP_GEMM, w, x, scratch, ntg, kcn, xs, kpr, seg, M, cap, dst, bias, bias2, add, pitch, row_off,
                        act, nlimit, nacc, ks, (kcn + ks - 1) // ks, mt, ring])

        if hdr["normalize"]:
            ops.append([_OP_STD, 0, hdr["std_off"]])
        for i, e in enumerate(encs):
            pi, ph, po = e["ld_in"] + 4, e["ld_h"] + 4, e["ld_out"] + 4
            if i == 0:
                ops.append([_OP_ENC0, e["n"], e["ld_h"], e["w1"], e["b1"], R0, R1, ph])
            else:
                gemm(e["w1"], R0, R2, e["nt1"], e["kc1"], 2 * pi, e["ld_in"] // 16, pi, e["n"], cap2, R1, bias=e["b1"],
                     pitch=ph, act=_ACT_RELU)
            gemm(e["w2"], R1, R2, e["ntg2"], e["kc2"], ph, e["kc2"], 0, e["n"], cap2, R0, bias=e["b2"],
                 bias2=e["b2"] + e["ld_out"], pitch=po, row_off=2, nacc=2, ring=e["ring"])
            # (ring: the n new rows also go to the layer's ring, the two rows in front of them become rows 0, 1 of R0 --
            #  a separate op until the second half of round 5)
        x_enc = R0 + 2 * (hdr["ld_last"] + 4)
        gemm(hdr["t1_w"], x_enc, R2, hdr["t1_nt"], hdr["t1_kc"], 0, hdr["t1_kc"], 0, 1, cap2, v_hs, bias=hdr["t1_b"])
        for k, b in enumerate(blks):
            dip = b["dip"]
            s_xz = misc
            s_x = s_xz + 2 * dip + 16
            s_dt, s_y, s_xdb = s_x + dip, s_x + 2 * dip, s_x + 3 * dip
            ops.append([_OP_LN, v_hs, v_res, v_h, b["ln_w"], b["ln_b"], b["ln_eps"], dm, dmp, int(k > 0)])
            gemm(b["in_w"], v_h, R2, b["nt_in"], b["kc_in"], 0, b["kc_in"], 0, 1, cap2, s_xz, bias=b["in_b"])
            ops.append([_OP_CONVSTEP, b["di"], dip, b["W"], b["conv_state"], b["conv_w"], b["conv_b"], s_xz, s_x])
            gemm(b["xp_w"], s_x, R2, b["nt_xp"], b["kc_xp"], 0, b["kc_xp"], 0, 1, cap2, s_xdb)
            gemm(b["dt_w"], s_xdb, R2, b["nt_dt"], b["kc_dt"], 0, b["kc_dt"], 0, 1, cap2, s_dt, bias=b["dt_b"],
                 act=_ACT_SOFTPLUS)
            ops.append([_OP_SSM, b["di"], dip, b["N"], b["ssm_state"], b["A"], b["D"], s_dt, s_x, s_xdb + b["R"],
                        s_xdb + b["R"] + b["N"], s_xz + b["di"], s_y])
            gemm(b["out_w"], s_y, R2, b["nt_out"], b["kc_out"], 0, b["kc_out"], 0, 1, cap2, v_hs, bias=b["out_b"])
        ops.append([_OP_LN, v_hs, v_res, v_h, hdr["nf_w"], hdr["nf_b"], hdr["nf_eps"], dm, dmp, int(len(blks) > 0)])
        gemm(hdr["t2_w"], v_h, R2, hdr["t2_nt"], hdr["t2_kc"], 0, hdr["t2_kc"], 0, 1, cap2, R0, bias=hdr["t2_b"], add=x_enc)
        for j, d in enumerate(decs):
            pi, pg, po, ldy = d["ld_in"] + 4, d["ld_g"] + 4, d["ld_out"] + 4, 4 * d["cq"]
            gemm(d["w1"], R0, R2, d["ntg1"], d["kc1"], pi, d["kc1"], 0, d["L"], cap2, R1, bias=d["b1"],
                 bias2=d["b1"] + d["ld_g"], pitch=pg, nacc=2)
            gemm(d["w2"], R1, R0, d["nt2"], d["kc2"], pg, d["kc2"], 0, d["L"], cap0, R2, pitch=ldy, nlimit=ldy)
            ops.append([_OP_OVERLAP, d["L"], d["ld_out"], d["cq"], d["cout"], R2, ldy, d["b2"], d["tail"], d["skip_ring"],
                        d["skip_ld"], d["skip_n"], d["relu"], int(j == E - 1), R0, po])
        if len(ops) > _MAX_OPS:
            raise ValueError("too many ops for one hop")
        ops_lds = RV + _rup(rv + pad, 4)
        lds_floats = ops_lds + _rup(len(ops) * _OP_INTS, 4)
        self.lds_bytes = _rup(4 * lds_floats, 16)
        if self.lds_bytes > hip.lib().cum_stream_hop_max_lds_bytes():
            raise ValueError(f"a hop of this model needs {self.lds_bytes} bytes of LDS")
        for k, op in enumerate(ops):
            ints[_HDR_INTS + k * _OP_INTS:_HDR_INTS + k * _OP_INTS + len(op)] = op
        # the products' stage lists: wtab[op][wave] = first stage | count << 16, then the stages themselves
        wtab0 = _HDR_INTS + _MAX_OPS * _OP_INTS
        stg0 = wtab0 + _MAX_OPS * _WAVES
        n_stages = 0
        self.stages = {}
        for k, op in enumerate(ops):
            if op[0] != _OP_GEMM:
                continue
            self.stages[k] = _stages(op)
            for wave, lst in enumerate(self.stages[k]):
                if len(lst) > _MAX_WAVE_STAGES or n_stages + len(lst) > min(_MAX_STAGES, 65535):
                    raise ValueError("a product of this model has more stages than the hop kernel's tables hold")
                ints[wtab0 + k * _WAVES + wave] = n_stages | len(lst) << 16
                for t in lst:
                    ints[stg0 + 4 * n_stages:stg0 + 4 * n_stages + 4] = t
                    n_stages += 1
        ints[:8] = [_MAGIC, len(ops), frame_len, hop, lds_floats, hdr["phase_off"], ops_lds, n_stages]
        self.ops = ops
        self.hdr, self.encs, self.decs, self.blks = hdr, encs, decs, blks
        self.plan = torch.from_numpy(ints).to(dev)
        # 32 KiB of zeros behind the last matrix: the kernel's stage loads run past short tiles without a bounds check
        self.weights = torch.cat(blob.parts + [torch.zeros(8192, dtype=torch.float32, device=dev)])
        self.state_stride = _rup(state_off, 4)
        self.hop, self.frame_len, self.device = hop, frame_len, dev
        # model flops per hop and stream (2 x multiply-adds of the layers' real sizes): the `roofline` of bench.py's C5 row
        self.flops_per_hop = 2 * sum(
            e["n"] * (enc[0].weight.numel() + enc[2].weight.numel()) for e, enc in zip(encs, model.encoder))
        self.flops_per_hop += 2 * sum(
            d["L"] * (dec[0].weight.numel() + dec[2].weight.numel()) for d, dec in zip(decs, model.decoder))
        self.flops_per_hop += 2 * (model.tsfm_conv1.weight.numel() + model.tsfm_conv2.weight.numel()) + 2 * sum(
            b.mixer.in_proj.weight.numel() + b.mixer.x_proj.weight.numel() + b.mixer.dt_proj.weight.numel()
            + b.mixer.out_proj.weight.numel() + 3 * b.mixer.A_log.numel() for b in model.tsfm_Mamba_layers)

    # ------------------------------------------------------------------ state
    def import_state(self, model, S):
        """State blocks [S, state_stride] from the per-layer path's state right after the first frame of the streams."""
        st = torch.zeros(S, self.state_stride, dtype=torch.float32, device=self.device)
        old = model.encoder_decoder_state
        if old["enc0"].dim() != 2:
            raise RuntimeError("import_state: the first frame must have run on the fused per-layer path")
        T = self.frame_len
        for i, e in enumerate(self.encs):
            T = (T - 4) // 2 + 1
            C, n = e["C"], e["n"]
            if T != 3 * n - 2:
                raise RuntimeError("import_state: unexpected window length")
            rows = cs.Geo(S, T, C).rows(old[f"enc{i}"])[:, :T, :C].float()
            ring = st[:, e["ring"]:e["ring"] + 3 * n * e["ld_out"]].view(S, 3 * n, e["ld_out"])
            ring[:, :T, :C] = rows
        for j, d in enumerate(self.decs):
            tail = st[:, d["tail"]:d["tail"] + 2 * d["cq"]].view(S, 2, d["cq"])
            tail[:, :, :d["cout"]] = old[f"dec{j}"][:, :, :d["cout"]].float()
        kv = model.inference_params.key_value_memory_dict
        for k, b in enumerate(self.blks):
            conv_state, ssm_state = kv[k]
            st[:, b["conv_state"]:b["conv_state"] + b["di"] * b["W"]] = conv_state.reshape(S, -1).float()
            st[:, b["ssm_state"]:b["ssm_state"] + b["di"] * b["N"]] = ssm_state.reshape(S, -1).float()
        if self.hdr["normalize"]:
            st[:, 0] = model.input_std.reshape(S).float()
            st[:, 1] = float(model._std_frames)
        return st

    def export_state(self, model, st):
        """The per-layer path's view of the state (the layout ``_drain`` reads: unconsumed encoder rows (S, C, 2 n - 2),
        decoder tails (S, C, 2)), plus the Mamba states and the running std."""
        S = st.shape[0]
        phase = int(st[0, self.hdr["phase_off"]].item())
        out = {}
        for i, e in enumerate(self.encs):
            C, n = e["C"], e["n"]
            ring = st[:, e["ring"]:e["ring"] + 3 * n * e["ld_out"]].view(S, 3 * n, e["ld_out"])
            idx = (phase * n + torch.arange(n, 3 * n - 2, device=st.device)) % (3 * n)
            out[f"enc{i}"] = ring[:, idx, :C].transpose(1, 2).contiguous()
        for j, d in enumerate(self.decs):
            tail = st[:, d["tail"]:d["tail"] + 2 * d["cq"]].view(S, 2, d["cq"])
            out[f"dec{j}"] = tail[:, :, :d["cout"]].transpose(1, 2).contiguous()
        kv = model.inference_params.key_value_memory_dict
        for k, b in enumerate(self.blks):
            conv_state, ssm_state = kv[k]
            conv_state.copy_(st[:, b["conv_state"]:b["conv_state"] + b["di"] * b["W"]].view_as(conv_state))
            ssm_state.copy_(st[:, b["ssm_state"]:b["ssm_state"] + b["di"] * b["N"]].view_as(ssm_state))
        return out

    def run(self, st, frames, out, n_hops):
        """frames: (S, >= (n_hops - 1) hop + frame_len) raw samples, unit stride along time; out: (S, n_hops * hop)."""
        S = st.shape[0]
        assert frames.stride(1) == 1 and out.stride(1) == 1 and frames.dtype == out.dtype == torch.float32
        assert frames.shape[1] >= (n_hops - 1) * self.hop + self.frame_len and out.shape[1] >= n_hops * self.hop
        with torch.cuda.device(self.device):
            hip.check(hip.lib().cum_stream_hop(hip.ptr(self.plan), hip.ptr(self.weights), hip.ptr(st), self.state_stride,
                                               S, hip.ptr(frames), frames.stride(0), hip.ptr(out), out.stride(0), n_hops,
                                               self.lds_bytes, hip.stream_ptr()))
