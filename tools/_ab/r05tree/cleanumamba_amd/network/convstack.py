"""Encoder / decoder layers on the fused GEMM kernels (csrc/gemm.hip).

Host side of the "convolution as GEMM" design (DESIGN.md "conv stack").  It reads the
SAME parameters as the reference's modules -- encoder[i][0] Conv1d(k4,s2), encoder[i][2]
Conv1d(1x1), decoder[j][0] Conv1d(1x1), decoder[j][2] ConvTranspose1d(k4,s2)
(src/network/CleanUMamba.py:108-113, 121-130) -- re-packs them into GEMM operands and runs
every layer forward and backward-data through ``cum_gemm_nt`` and every weight/bias gradient
through ``cum_gemm_tn``.

Activation layout ("rows"): a 2-D tensor [1 + B*(T+2) + slack, Cp]; Cp = channels rounded
up to 8; row 0 is a zero row, then per clip T real rows followed by 2 zero rows; the slack
rows are zero.  For inputs of ``valid_length`` the pitch T+2 halves exactly with every
encoder layer (T_in + 2 == 2 * (T_out + 2)), which is what makes the strided conv and the
transposed conv single GEMMs over the flat buffer:
  conv k4 s2 : A row m = 4*Cp contiguous elements at element offset (1 + 2m) * Cp
  convT k4 s2: A row m = 2*Cp contiguous elements at element offset m * Cp  (rows m-1, m)
"""
import ctypes
import os

import torch

from .. import hip


def rup(x, m):
    return (x + m - 1) // m * m


class Geo:
    """Geometry of one activation buffer."""

    def __init__(self, B, T, C):
        self.B, self.T, self.C = B, T, C
        self.P = T + 2
        self.Cp = rup(C, 8)
        self.M = B * self.P                      # flat rows that carry clips
        self.slack = 2 + (128 + self.Cp - 1) // self.Cp
        self.R = 1 + self.M + self.slack

    def new(self, dtype, device, zero=False):
        """Uninitialised buffer.  The kernel that fills it also clears the leading row and the slack rows
        (``head`` / ``tail`` elements, passed as zero_head / zero_tail)."""
        if zero:
            return torch.zeros(self.R, self.Cp, dtype=dtype, device=device)
        return torch.empty(self.R, self.Cp, dtype=dtype, device=device)

    @property
    def head(self):
        return self.Cp

    @property
    def tail(self):
        return self.slack * self.Cp

    def rows(self, buf):
        """[B, P, Cp] view of the clip rows."""
        return buf[1:1 + self.M].view(self.B, self.P, self.Cp)


def bk_of(dtype):
    return 64 if dtype in hip.HALF_TYPES else 32


# ------------------------------------------------------------------ GEMM launcher
def gemm(A, a_off, lda, Wp, bias, out, o_off, ldc, M, pitch, valid, epilogue, n_store, res=None, r_off=0, ldr=0,
         aux=None, x_off=0, ldz=0, geo=None, aux2=None, y_off=0, ldy=0, gate_only=False, mask_bits=False, split_k=False):
    """A, out, res, aux, aux2: flat tensors; *_off element offsets of row 0; Wp [N, K] packed weights."""
    hip.require_gpu(A, Wp, out, res, aux, aux2, any_dtype=True)
    dt = A.dtype
    if Wp.dtype != dt or out.dtype != dt:
        raise RuntimeError("gemm: A, W and out must share one dtype")
    esz = A.element_size()
    d = hip.GemmDesc()
    d.dtype, d.epilogue = hip.dtype_code(dt), epilogue
    d.M, d.N, d.K = M, Wp.shape[0], Wp.shape[1]
    d.lda, d.ldw, d.ldc, d.ldr, d.ldz = lda, Wp.stride(0), ldc, ldr if res is not None else 4, ldz if aux is not None else 4
    d.pitch, d.valid, d.n_store = pitch, valid, n_store
    d.gate_only, d.ldy = int(gate_only), ldy if aux2 is not None else 4
    d.mask_bits = int(mask_bits)
    # small-M split-K kernel: streaming hops (see small_m_gemms); split_k: asked for by the caller (narrow projections)
    d.allow_split_k = 2 if split_k else int(_SPLIT_K_OK)
    if geo is not None:      # out (and an activation-type aux) is a row buffer of this geometry: frame it with zeros
        d.zero_head, d.zero_tail = geo.head, geo.tail
    # offsets are in elements of the GEMM dtype; bit arrays (mask_bits) are passed as views that start at the right word
    P = lambda t, off: None if t is None else ctypes.c_void_p(t.data_ptr() + off * (esz if t.dtype == dt else 0))
    with torch.cuda.device(A.device):
        hip.check(hip.lib().cum_gemm_nt(ctypes.byref(d), P(A, a_off), P(Wp, 0), hip.ptr(bias), P(res, r_off),
                                        P(out, o_off), P(aux, x_off), P(aux2, y_off), hip.stream_ptr()))


_SPLIT_K_OK = False


class small_m_gemms:
    """Context: cum_gemm_nt launches inside may use the small-M kernel (64x64 tiles, K split over the waves), which
    differs from the standard kernels in summation order.  The streaming hop and the no-grad parallel forward opt in;
    training keeps one summation order for every shape (grad mode cannot be the switch: it is off inside backward())."""

    def __enter__(self):
        global _SPLIT_K_OK
        self.prev, _SPLIT_K_OK = _SPLIT_K_OK, True

    def __exit__(self, *exc):
        global _SPLIT_K_OK
        _SPLIT_K_OK = self.prev


def wgrad(dZ, z_off, ldz, N, X, x_off, ldx, K, M, want_bias=True, out=None, out_w=None, out_b=None):
    """dW [N, K] f32 = dZ^T X (X rows may overlap), db [N] f32 = column sums of dZ.  csrc/gemm_tn.hip.
    ``out = (flat f32 buffer, offset)`` places dW then db at that offset (N*K + N elements) instead of allocating;
    ``out_w`` / ``out_b``: contiguous f32 tensors of N*K / N elements that receive dW / db (gradient views of the
    flat buffer, see grad_sink)."""
    lib = hip.lib()
    dc = hip.dtype_code(dZ.dtype)
    dev = dZ.device
    if out_w is not None:
        dW = out_w.view(N, K)
        db = (out_b if out_b is not None else torch.empty(N, dtype=torch.float32, device=dev)) if want_bias else None
    elif out is not None:
        arena, off = out
        dW = arena[off:off + N * K].view(N, K)
        db = arena[off + N * K:off + N * K + N] if want_bias else None
    else:
        dW = torch.empty(N, K, dtype=torch.float32, device=dev)
        db = torch.empty(N, dtype=torch.float32, device=dev) if want_bias else None
    ws = torch.empty(max(lib.cum_gemm_tn_workspace_elems(dc, M, N, K), 1), dtype=torch.float32, device=dev)
    esz = dZ.element_size()
    with torch.cuda.device(dev):
        hip.check(lib.cum_gemm_tn(dc, M, N, K, ctypes.c_void_p(dZ.data_ptr() + z_off * esz), ldz,
                                  ctypes.c_void_p(X.data_ptr() + x_off * esz), ldx, hip.ptr(dW), K, hip.ptr(db),
                                  hip.ptr(ws), hip.stream_ptr()))
    return dW, db


def colsum(X, x_off, ld, M, n):
    lib = hip.lib()
    out = torch.empty(n, dtype=torch.float32, device=X.device)
    ws = torch.empty(max(lib.cum_colsum_workspace_elems(M, n), 1), dtype=torch.float32, device=X.device)
    with torch.cuda.device(X.device):
        hip.check(lib.cum_colsum(hip.dtype_code(X.dtype), M, n, ctypes.c_void_p(X.data_ptr() + x_off * X.element_size()),
                                 ld, hip.ptr(out), hip.ptr(ws), hip.stream_ptr()))
    return out


# ------------------------------------------------------------------ weight packing
# Every packed operand is "gather from [flattened parameter, 0]".  The gather index depends only on shapes,
# so it is built once on the host (by running the layout code below on element ids) and cached; at run time
# a pack is one pad/cast kernel + one index_select, with no host synchronisation.
_INDEX_CACHE = {}


def gather(src_flat, idx32, dtype, out=None):
    """[src_flat[idx] or 0 where idx < 0] as ``dtype``; csrc/pack.hip (32-bit index, conversion in the same pass).
    ``out``: write into this buffer (same length and dtype) instead of allocating."""
    if out is None or out.numel() != idx32.numel() or out.dtype != dtype or out.device != src_flat.device:
        out = torch.empty(idx32.numel(), dtype=dtype, device=src_flat.device)
    if not src_flat.is_contiguous():
        src_flat = src_flat.contiguous()
    with torch.cuda.device(src_flat.device):
        hip.check(hip.lib().cum_gather(hip.dtype_code(src_flat.dtype), hip.ptr(src_flat), hip.ptr(idx32), idx32.numel(),
                                       hip.dtype_code(dtype), hip.ptr(out), hip.stream_ptr()))
    return out


def _ids(shape):
    n = 1
    for d in shape:
        n *= d
    return torch.arange(1, n + 1, dtype=torch.int64).view(shape)      # 0 is reserved for "zero padding"


PACK_PAD = -2 ** 31


def _separable(g2):
    """g2: [R, C] int64 global source index of a packed operand (-1 = zero padding).  Returns (rowoff, coloff, transpose)
    with g2[r][c] == rowoff[r] + coloff[c] wherever both are valid (PACK_PAD marks padding), or None if the layout does not
    separate.  transpose: neighbouring destination ROWS are the near neighbours in the source."""
    valid = g2 >= 0
    rv, cv = valid.any(1), valid.any(0)
    if not bool(rv.any()) or not torch.equal(valid, rv[:, None] & cv[None, :]):
        return None
    r0, c0 = int(rv.nonzero()[0]), int(cv.nonzero()[0])
    coloff = g2[r0] - g2[r0, c0]
    rowoff = g2[:, c0].clone()
    if not torch.equal((rowoff[:, None] + coloff[None, :])[valid], g2[valid]):
        return None
    if int(rowoff[rv].max()) + int(coloff[cv].abs().max()) >= 2 ** 31 - 1:
        return None
    dr = (rowoff[rv][1:] - rowoff[rv][:-1]).abs().float().median() if int(rv.sum()) > 1 else torch.tensor(float("inf"))
    dc = (coloff[cv][1:] - coloff[cv][:-1]).abs().float().median() if int(cv.sum()) > 1 else torch.tensor(float("inf"))
    rowoff = torch.where(rv, rowoff, torch.full_like(rowoff, PACK_PAD))
    coloff = torch.where(cv, coloff, torch.full_like(coloff, PACK_PAD))
    return rowoff.to(torch.int32), coloff.to(torch.int32), bool(dr < dc)


def _runs8(ro, co, base=0):
    """cum_pack2d's fast path for a separable layout: 1 if every aligned group of 8 destination columns is either all
    padding or 8 consecutive source elements, 2 if in addition every run starts on a 16-byte boundary of the f32 source
    (``base``: element offset of the source inside its 16-byte aligned buffer), else 0."""
    c = co.to(torch.int64).view(-1, 8)
    pad = c == PACK_PAD
    if bool((pad.any(1) & ~pad.all(1)).any()):
        return 0
    live = ~pad.all(1)
    if not bool(live.any()):
        return 0
    cl = c[live]
    if not torch.equal(cl, cl[:, :1] + torch.arange(8, dtype=torch.int64)):
        return 0
    rl = ro.to(torch.int64)
    rl = rl[rl != PACK_PAD]
    aligned = bool(((cl[:, 0] % 4) == 0).all()) and bool((((rl + base) % 4) == 0).all())
    return 2 if aligned else 1


class PackPlan:
    """Batches every weight / bias pack of one model into two gathers per forward.

    The first step runs un-batched and records each (parameter, layout) request made through ``take``; from the
    second forward on, ``refresh()`` concatenates the parameters once, casts once, and gathers all packed operands
    (forward and backward-data ones) with one ``index_select`` per dtype; ``take`` then returns views."""

    def __init__(self, params):
        self.params = [p for p in params]
        # Parameters that live in ONE flat f32 buffer (training/flat_optim.py: p.data are views of it) are gathered
        # straight from that buffer, cast in the same pass: no concatenation, no cast pass.  Otherwise they are
        # concatenated first.
        self.source = self._shared_buffer(self.params)
        self.offset, off = {}, 0
        for p in self.params:
            if self.source is not None:
                self.offset[p.data_ptr()] = (p.data_ptr() - self.source.data_ptr()) // p.element_size()
            else:
                self.offset[p.data_ptr()] = off
            off += p.numel()
        self.total = off
        self.reqs = {}                        # (key, ptr, dtype) -> (global index (cpu), shape)
        self.dirty = False
        self.gidx = {}                        # (dtype, late) -> (device index tensor, [(reqkey, start, numel, shape)])
        self.current = {}
        self.big = {}                         # (dtype, late) -> the flat buffer all packed operands of that group are views of
        self.late_event = None                # side-stream gather of the backward-only operands still to be joined

    @staticmethod
    def _shared_buffer(params):
        """A flat view over the storage all ``params`` are contiguous f32 slices of, or None."""
        if not params or any(p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda for p in params):
            return None
        st = params[0].untyped_storage()
        if any(p.untyped_storage().data_ptr() != st.data_ptr() for p in params):
            return None
        n = st.nbytes() // 4
        if n >= 2 ** 31:
            return None
        return torch.empty(0, dtype=torch.float32, device=params[0].device).set_(st, 0, (n,), (1,))

    def record(self, key, src, idx, shape, dtype):
        rk = (key, src.data_ptr(), dtype)
        if rk in self.reqs:
            return
        local = idx.cpu().to(torch.int64)
        g = local + self.offset[src.data_ptr()]
        g[local < 0] = -1                     # zero padding
        self.reqs[rk] = (g.to(torch.int32), shape)
        self.dirty = True

    # layouts only the backward reads (data-gradient operands): skipped without autograd
    LATE = ("conv_dgrad", "convt_dgrad", "glu_dgrad", "plain_dgrad", "proj_t")
    # "1": gather them on a side stream beside the forward's GEMMs.  Measured on the E8 train step (same box, two runs
    # each): 21.44 / 21.48 ms with the side stream against 21.23 / 21.25 ms without -- the gather's scattered reads slow
    # the concurrent GEMMs by more than the 0.3 ms it takes when run alone (as the round-2 attempt to run weight-gradient
    # GEMMs on a second stream: every launch already fills the chip).  Off.
    SIDE_STREAM = os.environ.get("CUM_PACK_SIDE_STREAM", "0") == "1"

    def refresh(self):
        """Recompute every recorded pack from the current parameter values.  While the set of recorded packs is
        unchanged the packed operands are rewritten IN PLACE: a captured streaming hop (CleanUMamba._hop) replays
        kernels that hold their addresses.

        The operands of the data-gradient GEMMs (``LATE`` keys, about half of the packed bytes) are not needed before the
        backward: with autograd off they are not gathered at all (``SIDE_STREAM``: see above)."""
        grad_on = torch.is_grad_enabled()
        if not self.reqs:
            self.current = {}
            return
        if self.dirty:
            self.current, self.big = {}, {}
        dev = self.params[0].device
        if self.dirty:
            self.gidx = {}
            groups = {}
            for rk, (g, shape) in self.reqs.items():
                groups.setdefault((rk[2], rk[0][0] in self.LATE), []).append((rk, g, shape))
            for gk, items in groups.items():
                # separable 2-D layouts first (index-free cum_pack2d: two small tables per operand), the rest (bias
                # vectors, anything irregular) behind them through the per-element index of cum_gather
                metas, start = [], 0
                jobs, tiles, tables, tab_pos = [], [], [], 0
                rest = []
                for rk, g, shape in items:
                    sep = None
                    if _PACK2D and len(shape) == 2 and shape[1] % 8 == 0 and self.source is not None:
                        sep = _separable(g.view(shape).to(torch.int64))
                    if sep is None:
                        rest.append((rk, g, shape))
                        continue
                    ro, co, tr = sep
                    jobs.append((start, shape[0], shape[1], tab_pos, tab_pos + shape[0], int(tr), 0 if tr else _runs8(ro, co)))
                    tables += [ro, co]
                    tab_pos += shape[0] + shape[1]
                    for tr_ in range((shape[0] + 63) // 64):
                        for tc_ in range((shape[1] + 63) // 64):
                            tiles.append((len(jobs) - 1, tr_, tc_))
                    metas.append((rk, start, g.numel(), shape))
                    start += (g.numel() + 7) // 8 * 8
                rest_start, parts = start, []
                for rk, g, shape in rest:
                    metas.append((rk, start, g.numel(), shape))
                    parts.append(g)
                    start += g.numel()
                pack2d = None
                if jobs:
                    import numpy as np
                    jb = np.zeros(len(jobs), dtype=[("off", "<i8"), ("rows", "<i4"), ("cols", "<i4"), ("rt", "<i4"), ("ct", "<i4"),
                                                    ("tr", "<i4"), ("runs8", "<i4")])
                    for i, (off, r, c, rt, ct, tr, r8) in enumerate(jobs):
                        jb[i] = (off, r, c, rt, ct, tr, r8)
                    pack2d = (torch.from_numpy(jb.view(np.uint8)).to(dev), torch.tensor(tiles, dtype=torch.int32).to(dev),
                              torch.cat(tables).to(dev), len(tiles))
                gi = torch.cat(parts).to(dev) if parts else None
                self.gidx[gk] = (gi, metas, pack2d, rest_start, start)
            self.dirty = False
        self.late_event = None
        with torch.no_grad():
            flat = self.source if self.source is not None else torch.cat([p.detach().reshape(-1) for p in self.params])
            cast, side = {}, None
            for gk, (gi, metas, pack2d, rest_start, total) in self.gidx.items():
                dt, late = gk
                if late and not grad_on:
                    for rk, _, _, _ in metas:         # not needed without a backward: a stale copy must not be served
                        self.current.pop(rk, None)
                    self.big.pop(gk, None)
                    continue
                if self.source is None and flat.dtype != dt:
                    # cast first: the gather then reads 2-byte elements of a source that stays cache-resident
                    if dt not in cast:
                        cast[dt] = flat.to(dt)
                    flat_dt = cast[dt]
                else:
                    flat_dt = flat                    # flat parameter buffer: converted inside the gather
                big = self.big.get(gk)
                fresh = big is None
                if fresh:
                    big = torch.zeros(max(total, 8), dtype=dt, device=dev)

                def run():
                    if pack2d is not None:
                        jb, tl, tb, ntiles = pack2d
                        with torch.cuda.device(dev):
                            hip.check(hip.lib().cum_pack2d(hip.ptr(flat), hip.ptr(jb), hip.ptr(tl), ntiles, hip.ptr(tb),
                                                           hip.dtype_code(dt), hip.ptr(big), hip.stream_ptr()))
                    if gi is not None:
                        gather(flat_dt, gi, dt, out=big[rest_start:rest_start + gi.numel()])
                if late and self.SIDE_STREAM:
                    if side is None:
                        side = self.__dict__.get("_side")
                        if side is None:
                            side = self.__dict__["_side"] = torch.cuda.Stream(device=dev)
                        side.wait_stream(torch.cuda.current_stream(dev))
                    with torch.cuda.stream(side):
                        run()
                else:
                    run()
                if fresh:
                    self.big[gk] = big
                    for rk, start, n, shape in metas:
                        self.current[rk] = big[start:start + n].view(shape)
            if side is not None:
                self.late_event = side.record_event()

    def join(self):
        """First backward-side use of a late operand: the current stream waits for the side-stream gather."""
        ev = self.late_event
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
            self.late_event = None


_PACK2D = os.environ.get("CUM_PACK2D", "1") != "0"      # "0": every packed operand through the per-element index (A/B)
_ACTIVE_PLAN = None


def set_active_plan(plan):
    global _ACTIVE_PLAN
    _ACTIVE_PLAN = plan


def take(src, key, build, dtype=None):
    """out = [src.flatten(), 0][index].view(shape); ``build()`` returns the id layout (0 = padding)."""
    plan = _ACTIVE_PLAN
    if plan is not None and src.data_ptr() in plan.offset:
        hit = plan.current.get((key, src.data_ptr(), dtype if dtype is not None else src.dtype))
        if hit is not None:
            if plan.late_event is not None and key[0] in plan.LATE:
                plan.join()
            return hit
    ent = _INDEX_CACHE.get((key, src.device))
    if ent is None:
        ids = build()
        idx = (ids.reshape(-1) - 1).to(torch.int32)          # -1 = zero padding
        ent = (idx.to(src.device), tuple(ids.shape))
        _INDEX_CACHE[(key, src.device)] = ent
    idx, shape = ent
    if plan is not None and src.data_ptr() in plan.offset:
        plan.record(key, src, idx, shape, dtype if dtype is not None else src.dtype)
    return gather(src.detach().reshape(-1), idx, dtype if dtype is not None else src.dtype).view(shape)


def _invert(packed_ids, param_shape):
    """packed_ids: id layout of a packed tensor (values = 1-based parameter element ids, 0 = padding).
    Returns the id layout that gathers the parameter back out of the packed tensor."""
    flat = packed_ids.reshape(-1)
    n = 1
    for d in param_shape:
        n *= d
    out = torch.zeros(n, dtype=torch.int64)
    pos = torch.arange(1, flat.numel() + 1, dtype=torch.int64)
    m = flat > 0
    out[flat[m] - 1] = pos[m]
    return out.view(param_shape)


def _zeros(rows, cols):
    return torch.zeros(rows, cols, dtype=torch.int64)


# Weight gradients of a whole stack: every cum_gemm_tn of the stack's backward writes (dW, db) in GEMM layout into one
# flat f32 arena, and ONE gather (index built once per stack signature) brings all of them into the parameters'
# layouts -- instead of two or three gathers per layer (~50 launches per step).
_ARENA_BATCH = os.environ.get("CUM_WGRAD_ARENA", "1") != "0"      # "0": one gather per parameter (A/B timing)
_ARENA_INDEX = {}


def grad_sink(params):
    """(FlatParams, [index], [offset]) if every parameter's gradient may be WRITTEN into the flat gradient buffer of
    training/flat_optim.py right now (all flat-managed by one buffer, nothing accumulated since zero_grad), else None.
    Producers then skip autograd's AccumulateGrad: one launch instead of one read-read-write add per tensor."""
    from ..training.flat_optim import sink_of
    flat = sink_of(params[0])
    if flat is None or not _GRAD_SINK:
        return None
    idx, offs = [], []
    for p in params:
        if sink_of(p) is not flat:
            return None
        sl = flat.slot(p)
        if sl is None:
            return None
        idx.append(sl[0])
        offs.append(sl[1])
    return flat, idx, offs


_GRAD_SINK = os.environ.get("CUM_GRAD_SINK", "1") != "0"        # "0": every gradient goes through AccumulateGrad (A/B)


class _WgradArena:
    def __init__(self, sizes, dev):
        """sizes: [(N, K)] per weight-gradient GEMM, in slot order."""
        self.sizes, self.offs, off = sizes, [], 0
        for N, K in sizes:
            self.offs.append(off)
            off += (N * K + N + 3) // 4 * 4           # 16-byte aligned slots (the GEMM stores float4)
        self.buf = torch.empty(max(off, 4), dtype=torch.float32, device=dev)

    def out(self, slot):
        return (self.buf, self.offs[slot])

    def dw_index(self, slot, inv_ids):
        """inv_ids: _invert(...) layout of a parameter (1-based position in the slot's dW, 0 = no source)."""
        ids = inv_ids.reshape(-1)
        return torch.where(ids > 0, ids - 1 + self.offs[slot], torch.full_like(ids, -1))

    def db_index(self, slot, inv_ids):
        N, K = self.sizes[slot]
        ids = inv_ids.reshape(-1)
        return torch.where(ids > 0, ids - 1 + self.offs[slot] + N * K, torch.full_like(ids, -1))

    def unpack_into(self, key, build_parts, shapes, dst, offs):
        """The arena's weight / bias gradients into the flat gradient buffer ``dst``: parameter k lands at element offset
        offs[k]; the alignment padding between parameters is not written (FlatParams.zero_grad cleared it).  One
        index-free launch (cum_pack2d) for everything whose layout separates."""
        lo = min(offs)
        hi = max(o + _numel(sh) for o, sh in zip(offs, shapes))
        if sum((_numel(sh) + 3) // 4 * 4 for sh in shapes) < hi - lo:
            raise RuntimeError("unpack_into: the parameters are not one contiguous run of the flat buffer")
        # the cached jobs hold offsets RELATIVE to lo (the same stack in another flat layout -- other bottleneck, other
        # frozen set -- must not reuse absolute positions); lo's 16-byte phase decides which parameters pack2d may take
        ck = (key, "into", tuple(o - lo for o in offs), lo % 4, self.buf.device)
        ent = _ARENA_INDEX.get(ck)
        if ent is None:
            # Every GEMM-layout -> parameter-layout map separates (source = rowoff[r] + coloff[c] over the parameter seen
            # as a matrix: weights (a, b * c), vectors (1, n)), like the forward packs: the index-free cum_pack2d brings
            # them over with two small tables per parameter instead of a 4-byte index per element.  What does not
            # separate (or is narrower than 8 columns) keeps the per-element gather.
            parts = build_parts()
            jobs, tiles, tables, tab_pos, rest = [], [], [], 0, []
            for k, (part, o, sh) in enumerate(zip(parts, offs, shapes)):
                rows, cols = (sh[0], _numel(sh) // sh[0]) if len(sh) > 1 else (1, _numel(sh))
                sep = _separable(part.reshape(rows, cols).to(torch.int64)) if (_PACK2D and cols % 8 == 0 and o % 4 == 0) else None
                if sep is None or bool((part.reshape(-1) < 0).any()):
                    rest.append(k)
                    continue
                ro, co, tr = sep
                jobs.append((o - lo, rows, cols, tab_pos, tab_pos + rows, int(tr), 0 if tr else _runs8(ro, co)))
                tables += [ro, co]
                tab_pos += rows + cols
                for tr_ in range((rows + 63) // 64):
                    for tc_ in range((cols + 63) // 64):
                        tiles.append((len(jobs) - 1, tr_, tc_))
            pack2d = None
            if jobs:
                import numpy as np
                jb = np.zeros(len(jobs), dtype=[("off", "<i8"), ("rows", "<i4"), ("cols", "<i4"), ("rt", "<i4"), ("ct", "<i4"),
                                                ("tr", "<i4"), ("runs8", "<i4")])
                for i, (off, r, c, rt, ct, tr, r8) in enumerate(jobs):
                    jb[i] = (off, r, c, rt, ct, tr, r8)
                dev = self.buf.device
                pack2d = (torch.from_numpy(jb.view(np.uint8)).to(dev), torch.tensor(tiles, dtype=torch.int32).to(dev),
                          torch.cat(tables).to(dev), len(tiles))
            gidx = None
            if rest:
                # (the leftovers need not be contiguous: one small gather each)
                gidx = [(offs[k] - lo, parts[k].reshape(-1).to(torch.int32).to(self.buf.device)) for k in rest]
            ent = (pack2d, gidx)
            _ARENA_INDEX[ck] = ent
        pack2d, gidx = ent
        dst = dst[lo:hi]
        if pack2d is not None:
            jb, tl, tb, ntiles = pack2d
            with torch.cuda.device(self.buf.device):
                hip.check(hip.lib().cum_pack2d(hip.ptr(self.buf), hip.ptr(jb), hip.ptr(tl), ntiles, hip.ptr(tb),
                                               hip.dtype_code(torch.float32), hip.ptr(dst), hip.stream_ptr()))
        for o, gi in (gidx or ()):
            gather(self.buf, gi, torch.float32, out=dst[o:o + gi.numel()])

    def unpack(self, key, build_parts, shapes):
        """One gather for all parameters; build_parts() -> [index tensor per parameter]; returns views per shape."""
        ent = _ARENA_INDEX.get((key, self.buf.device))
        if ent is None:
            ent = torch.cat([p.reshape(-1) for p in build_parts()]).to(torch.int32).to(self.buf.device)
            _ARENA_INDEX[(key, self.buf.device)] = ent
        flat = gather(self.buf, ent, torch.float32)
        outs, off = [], 0
        for sh in shapes:
            n = 1
            for d in sh:
                n *= d
            outs.append(flat[off:off + n].view(sh))
            off += n
        return outs


def _contiguous_run(offs, shapes):
    """True if the parameters (16-byte aligned, as FlatParams lays them out) fill [min offset, max end) without strangers."""
    lo = min(offs)
    hi = max(o + _numel(sh) for o, sh in zip(offs, shapes))
    return sum((_numel(sh) + 3) // 4 * 4 for sh in shapes) >= hi - lo


def _numel(shape):
    n = 1
    for d in shape:
        n *= d
    return n


def lay_conv_fwd(wshape, cp_in, rows, cols):
    """Conv1d(k4,s2) weight (H, Cin, 4) -> [rows][cols], element (h, kk*cp_in + c)."""
    H, Cin, _ = wshape
    out = _zeros(rows, cols)
    out[:H, :4 * cp_in].view(H, 4, cp_in)[:, :, :Cin] = _ids(wshape).permute(0, 2, 1)
    return out


def lay_conv_dgrad(wshape, cp_in, cp_out, rows, cols):
    """Conv1d(k4,s2) weight -> transposed-conv form [(j, c)][(half, h)]: half 0 <-> tap j+2, half 1 <-> tap j."""
    H, Cin, _ = wshape
    out = _zeros(rows, cols)
    wv = out[:2 * cp_in, :2 * cp_out].view(2, cp_in, 2, cp_out)
    wt = _ids(wshape).permute(2, 1, 0)                                   # [kk][c][h]
    wv[0, :Cin, 0, :H], wv[0, :Cin, 1, :H] = wt[2], wt[0]
    wv[1, :Cin, 0, :H], wv[1, :Cin, 1, :H] = wt[3], wt[1]
    return out


def lay_convt_fwd(wshape, cp_in, cp_out, rows, cols):
    """ConvTranspose1d(k4,s2) weight (Cin, Cout, 4) -> [(j, co)][(half, c)]."""
    Cin, Cout, _ = wshape
    out = _zeros(rows, cols)
    wv = out[:2 * cp_out, :2 * cp_in].view(2, cp_out, 2, cp_in)
    wt = _ids(wshape).permute(2, 1, 0)                                   # [kk][co][c]
    wv[0, :Cout, 0, :Cin], wv[0, :Cout, 1, :Cin] = wt[2], wt[0]
    wv[1, :Cout, 0, :Cin], wv[1, :Cout, 1, :Cin] = wt[3], wt[1]
    return out


def lay_convt_dgrad(wshape, cp_out, rows, cols):
    """ConvTranspose1d weight -> strided-conv form [c][kk*cp_out + co]."""
    Cin, Cout, _ = wshape
    out = _zeros(rows, cols)
    out[:Cin, :4 * cp_out].view(Cin, 4, cp_out)[:, :, :Cout] = _ids(wshape).permute(0, 2, 1)
    return out


def glu_rows(H):
    """Row order of a GLU-packed matrix: per 32 rows, 16 a-rows (channels 16g..16g+15) then their 16 b-rows.
    Returns (row -> source row in the (2H, .) weight or -1, number of groups)."""
    G = (H + 15) // 16
    idx = torch.full((G * 32,), -1, dtype=torch.long)
    ch = torch.arange(H)
    g, c = ch // 16, ch % 16
    idx[g * 32 + c] = ch
    idx[g * 32 + 16 + c] = H + ch
    return idx, G


def lay_glu_fwd(wshape, rows, cols):
    """1x1 GLU weight (2H, Cin, 1) -> [G*32 packed rows][cols]."""
    H2, Cin, _ = wshape
    idx, _ = glu_rows(H2 // 2)
    out = _zeros(rows, cols)
    ok = idx >= 0
    out[:idx.numel()][ok, :Cin] = _ids((H2, Cin))[idx[ok]]
    return out


def lay_glu_vec(H2):
    idx, G = glu_rows(H2 // 2)
    out = torch.zeros(G * 32, dtype=torch.int64)
    ok = idx >= 0
    out[ok] = _ids((H2,))[idx[ok]]
    return out


def lay_plain(wshape, rows, cols, transpose=False):
    Cout, Cin, _ = wshape
    out = _zeros(rows, cols)
    ids = _ids((Cout, Cin))
    if transpose:
        out[:Cin, :Cout] = ids.t()
    else:
        out[:Cout, :Cin] = ids
    return out


def lay_vec(n, rows):
    out = torch.zeros(rows, dtype=torch.int64)
    out[:n] = _ids((n,))
    return out


# ===================================================================== fused layers
class ConvK4S2ReLU(torch.autograd.Function):
    """y = relu(conv1d(x, w, b, stride=2)) on row buffers.  w: (H, Cin, 4)."""

    @staticmethod
    def forward(ctx, xbuf, w, b, gi, go):
        dt, dev = xbuf.dtype, xbuf.device
        H, Cin, Kw = w.shape
        assert Kw == 4 and gi.P == 2 * go.P and gi.C == Cin and go.C == H
        Np, Kp = rup(H, 16), rup(4 * gi.Cp, bk_of(dt))
        sh = tuple(w.shape)
        wp = take(w, ("conv_fwd", sh, gi.Cp, Np, Kp), lambda: lay_conv_fwd(sh, gi.Cp, Np, Kp), dt)
        bp = take(b, ("vec", H, Np), lambda: lay_vec(H, Np), torch.float32)
        ybuf = go.new(dt, dev)
        gemm(xbuf, gi.Cp, 2 * gi.Cp, wp, bp, ybuf, go.Cp, go.Cp, go.M, go.P, go.T, hip.EPI_RELU, go.Cp, geo=go)
        ctx.gi, ctx.go = gi, go
        ctx.save_for_backward(xbuf, w, ybuf)
        return ybuf

    @staticmethod
    def backward(ctx, dy):
        xbuf, w, ybuf = ctx.saved_tensors
        gi, go = ctx.gi, ctx.go
        dt, dev = xbuf.dtype, xbuf.device
        H, Cin, _ = w.shape
        dy = dy.contiguous()
        # dz = dy * (y > 0), same row layout (closing rows stay zero because dy's are zero there)
        dz = go.new(dt, dev)
        with torch.cuda.device(dev):
            hip.check(hip.lib().cum_relu_bwd(hip.dtype_code(dt), go.M, go.Cp, hip.ptr(ybuf[1:]), go.Cp,
                                             hip.ptr(dy[1:]), go.Cp, hip.ptr(dz[1:]), go.Cp, go.head, go.tail, hip.stream_ptr()))
        # weight + bias gradient in one launch: X row m = the 4*Cp contiguous inputs of output row m
        dwp, dbp = wgrad(dz, go.Cp, go.Cp, go.Cp, xbuf, gi.Cp, 2 * gi.Cp, 4 * gi.Cp, go.M)
        db = dbp[:H]
        sh = tuple(w.shape)
        dw = take(dwp, ("conv_unpack", sh, gi.Cp, go.Cp),
                  lambda: _invert(lay_conv_fwd(sh, gi.Cp, go.Cp, 4 * gi.Cp), sh))
        dx = None
        if ctx.needs_input_grad[0]:
            # data gradient = transposed conv: pair row t' of dx reads dz rows t'-1, t'
            Nd, Kd = rup(2 * gi.Cp, 16), rup(2 * go.Cp, bk_of(dt))
            wd = take(w, ("conv_dgrad", sh, gi.Cp, go.Cp, Nd, Kd), lambda: lay_conv_dgrad(sh, gi.Cp, go.Cp, Nd, Kd), dt)
            dx = gi.new(dt, dev)
            gemm(dz, 0, go.Cp, wd, None, dx, gi.Cp, 2 * gi.Cp, go.M, go.P, go.T + 1, hip.EPI_BIAS, 2 * gi.Cp, geo=gi)
        return dx, dw.to(w.dtype), db.to(w.dtype), None, None


class PointwiseGLU(torch.autograd.Function):
    """y = glu(conv1x1(x, w, b)) (+ nothing); w: (2H, Cin, 1).  Rows in, rows out (same geometry T)."""

    @staticmethod
    def forward(ctx, xbuf, w, b, gi, go, save_z):
        dt, dev = xbuf.dtype, xbuf.device
        H2, Cin, _ = w.shape
        H = H2 // 2
        assert gi.T == go.T and go.C == H and gi.C == Cin
        G = (H + 15) // 16
        Kp = rup(gi.Cp, bk_of(dt))
        sh = tuple(w.shape)
        wp = take(w, ("glu_fwd", sh, G * 32, Kp), lambda: lay_glu_fwd(sh, G * 32, Kp), dt)
        bp = take(b, ("glu_vec", H2), lambda: lay_glu_vec(H2), torch.float32)
        ybuf = go.new(dt, dev)
        z = torch.empty(go.M, G * 32, dtype=dt, device=dev) if save_z else None
        gemm(xbuf, gi.Cp, gi.Cp, wp, bp, ybuf, go.Cp, go.Cp, go.M, go.P, go.T, hip.EPI_GLU, go.Cp,
             aux=z, x_off=0, ldz=G * 32, geo=go)
        ctx.gi, ctx.go, ctx.G = gi, go, G
        ctx.save_for_backward(xbuf, w, z)
        return ybuf

    @staticmethod
    def backward(ctx, dy):
        xbuf, w, z = ctx.saved_tensors
        if z is None:
            raise RuntimeError("PointwiseGLU was run without save_z; backward is unavailable")
        gi, go, G = ctx.gi, ctx.go, ctx.G
        dt, dev = xbuf.dtype, xbuf.device
        H2, Cin, _ = w.shape
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        with torch.cuda.device(dev):
            hip.check(hip.lib().cum_glu_bwd(hip.dtype_code(dt), go.M, G, go.Cp, hip.ptr(z), G * 32, hip.ptr(dy[1:]),
                                            go.Cp, hip.ptr(dz), hip.stream_ptr()))
        dwp, dbp = wgrad(dz, 0, G * 32, G * 32, xbuf, gi.Cp, gi.Cp, gi.Cp, go.M)
        sh = tuple(w.shape)
        db = take(dbp, ("glu_vec_unpack", H2), lambda: _invert(lay_glu_vec(H2), (H2,)))
        dw = take(dwp, ("glu_unpack", sh, G * 32, gi.Cp), lambda: _invert(lay_glu_fwd(sh, G * 32, gi.Cp), sh))
        dx = None
        if ctx.needs_input_grad[0]:
            Nd, Kd = rup(gi.Cp, 16), rup(G * 32, bk_of(dt))
            wt = take(w, ("glu_dgrad", sh, Nd, Kd),
                      lambda: torch.nn.functional.pad(lay_glu_fwd(sh, G * 32, Nd).t(), (0, Kd - G * 32)), dt)
            dx = gi.new(dt, dev)
            gemm(dz, 0, G * 32, wt, None, dx, gi.Cp, gi.Cp, gi.M, gi.P, gi.T, hip.EPI_BIAS, gi.Cp, geo=gi)
        return dx, dw.to(w.dtype), db.to(w.dtype), None, None, None


class ConvT4S2(torch.autograd.Function):
    """y = [relu](conv_transpose1d(x, w, b, stride=2)) [+ skip]; w: (Cin, Cout, 4).  go.P == 2 * gi.P."""

    @staticmethod
    def forward(ctx, xbuf, w, b, skip, gi, go, relu):
        dt, dev = xbuf.dtype, xbuf.device
        Cin, Cout, Kw = w.shape
        assert Kw == 4 and go.P == 2 * gi.P and gi.C == Cin and go.C == Cout
        N = 2 * go.Cp
        Np, Kp = rup(N, 16), rup(2 * gi.Cp, bk_of(dt))
        sh = tuple(w.shape)
        wp = take(w, ("convt_fwd", sh, gi.Cp, go.Cp, Np, Kp), lambda: lay_convt_fwd(sh, gi.Cp, go.Cp, Np, Kp), dt)

        def bias_layout():
            out = torch.zeros(Np, dtype=torch.int64)
            out[:N].view(2, go.Cp)[:, :Cout] = _ids((Cout,))
            return out
        bp = take(b, ("convt_vec", Cout, go.Cp, Np), bias_layout, torch.float32)
        ybuf = go.new(dt, dev)
        keep = relu and skip is not None            # the ReLU mask is not recoverable from y + skip
        act = go.new(dt, dev) if keep else None
        gemm(xbuf, 0, gi.Cp, wp, bp, ybuf, go.Cp, N, gi.M, gi.P, gi.T + 1, hip.EPI_RELU if relu else hip.EPI_BIAS, N,
             res=skip, r_off=go.Cp, ldr=N, aux=act, x_off=go.Cp, ldz=N, geo=go)
        ctx.gi, ctx.go, ctx.relu, ctx.has_skip = gi, go, relu, skip is not None
        ctx.save_for_backward(xbuf, w, act if keep else (ybuf if relu else None))
        return ybuf

    @staticmethod
    def backward(ctx, dy):
        xbuf, w, act = ctx.saved_tensors
        gi, go = ctx.gi, ctx.go
        dt, dev = xbuf.dtype, xbuf.device
        Cin, Cout, _ = w.shape
        dy = dy.contiguous()
        dskip = dy if ctx.has_skip else None
        if ctx.relu:
            dz = go.new(dt, dev)
            with torch.cuda.device(dev):
                hip.check(hip.lib().cum_relu_bwd(hip.dtype_code(dt), go.M, go.Cp, hip.ptr(act[1:]), go.Cp,
                                                 hip.ptr(dy[1:]), go.Cp, hip.ptr(dz[1:]), go.Cp, go.head, go.tail, hip.stream_ptr()))
        else:
            dz = dy
        # weight + bias gradient: pair rows of dz against the 2*Cp contiguous inputs (rows m-1, m) of x
        dwp, dbp = wgrad(dz, go.Cp, 2 * go.Cp, 2 * go.Cp, xbuf, 0, gi.Cp, 2 * gi.Cp, gi.M)
        db = (dbp[:go.Cp] + dbp[go.Cp:])[:Cout]
        sh = tuple(w.shape)
        dw = take(dwp, ("convt_unpack", sh, gi.Cp, go.Cp),
                  lambda: _invert(lay_convt_fwd(sh, gi.Cp, go.Cp, 2 * go.Cp, 2 * gi.Cp), sh))
        dx = None
        if ctx.needs_input_grad[0]:
            # data gradient = strided conv of dz: row t reads dz rows 2t..2t+3
            Nd, Kd = rup(gi.Cp, 16), rup(4 * go.Cp, bk_of(dt))
            wc = take(w, ("convt_dgrad", sh, go.Cp, Nd, Kd), lambda: lay_convt_dgrad(sh, go.Cp, Nd, Kd), dt)
            dx = gi.new(dt, dev)
            gemm(dz, go.Cp, 2 * go.Cp, wc, None, dx, gi.Cp, gi.Cp, gi.M, gi.P, gi.T, hip.EPI_BIAS, gi.Cp, geo=gi)
        return dx, dw.to(w.dtype), db.to(w.dtype), dskip, None, None, None


class Pointwise(torch.autograd.Function):
    """y = conv1x1(x, w, b) [+ skip]; w: (Cout, Cin, 1)."""

    @staticmethod
    def forward(ctx, xbuf, w, b, skip, gi, go):
        dt, dev = xbuf.dtype, xbuf.device
        Cout, Cin, _ = w.shape
        assert gi.T == go.T and gi.C == Cin and go.C == Cout
        Np, Kp = rup(Cout, 16), rup(gi.Cp, bk_of(dt))
        sh = tuple(w.shape)
        wp = take(w, ("plain_fwd", sh, Np, Kp), lambda: lay_plain(sh, Np, Kp), dt)
        bp = take(b, ("vec", Cout, Np), lambda: lay_vec(Cout, Np), torch.float32)
        ybuf = go.new(dt, dev)
        gemm(xbuf, gi.Cp, gi.Cp, wp, bp, ybuf, go.Cp, go.Cp, go.M, go.P, go.T, hip.EPI_BIAS, go.Cp,
             res=skip, r_off=go.Cp, ldr=go.Cp, geo=go)
        ctx.gi, ctx.go, ctx.has_skip = gi, go, skip is not None
        ctx.bias = b                         # the parameter object: only its identity is used (gradient sink lookup)
        ctx.save_for_backward(xbuf, w)
        return ybuf

    @staticmethod
    def backward(ctx, dy):
        xbuf, w = ctx.saved_tensors
        gi, go = ctx.gi, ctx.go
        dt, dev = xbuf.dtype, xbuf.device
        Cout, Cin, _ = w.shape
        dy = dy.contiguous()
        b = ctx.bias
        sink = grad_sink([w, b]) if (b is not None and Cout == go.Cp and Cin == gi.Cp) else None
        if sink is not None:               # dW / db written straight into the flat gradient buffer
            flat, idx, offs = sink
            wgrad(dy, go.Cp, go.Cp, go.Cp, xbuf, gi.Cp, gi.Cp, gi.Cp, go.M,
                  out_w=flat.grad[offs[0]:offs[0] + Cout * Cin], out_b=flat.grad[offs[1]:offs[1] + Cout])
            flat.wrote(idx)
            dw = db = None
        else:
            dwp, dbp = wgrad(dy, go.Cp, go.Cp, go.Cp, xbuf, gi.Cp, gi.Cp, gi.Cp, go.M)
            db = dbp[:Cout].to(w.dtype)
            dw = dwp[:Cout, :Cin].unsqueeze(-1).to(w.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            Nd, Kd = rup(gi.Cp, 16), rup(go.Cp, bk_of(dt))
            sh = tuple(w.shape)
            wt = take(w, ("plain_dgrad", sh, Nd, Kd), lambda: lay_plain(sh, Nd, Kd, transpose=True), dt)
            dx = gi.new(dt, dev)
            gemm(dy, go.Cp, go.Cp, wt, None, dx, gi.Cp, gi.Cp, gi.M, gi.P, gi.T, hip.EPI_BIAS, gi.Cp, geo=gi)
        return dx, dw, db, (dy if ctx.has_skip else None), None, None


# ===================================================================== row <-> tensor glue
def to_rows(x, geo, dtype):
    """(B, C, T) tensor -> row buffer (differentiable)."""
    buf = torch.zeros(geo.R, geo.Cp, dtype=dtype, device=x.device)
    geo.rows(buf)[:, :geo.T, :geo.C] = x.transpose(1, 2).to(dtype)
    return buf


def from_rows(buf, geo):
    """row buffer -> (B, C, T) view (channel stride 1)."""
    return geo.rows(buf)[:, :geo.T, :geo.C].transpose(1, 2)


# ------------------------------------------------------------------ plain projections (Mamba in / x / dt / out_proj)
def lay_proj(shape, Np, Kp):
    """nn.Linear weight [N, K] -> [Np, Kp] (zero padded): the NT GEMM's weight operand of y = x W^T."""
    N, K = shape
    ids = _zeros(Np, Kp)
    ids[:N, :K] = _ids((N, K))
    return ids


def lay_proj_t(shape, Np, Kp):
    """nn.Linear weight [N, K] -> its transpose [K -> Np rows, N -> Kp columns] (zero padded): the weight operand of the
    data gradient dx = dy W, which is the NT GEMM dy (W^T)^T."""
    N, K = shape
    ids = _zeros(Np, Kp)
    ids[:K, :N] = _ids((N, K)).t()
    return ids


_ZERO_BIAS = {}


def _zero_bias(n, dev):
    z = _ZERO_BIAS.get((n, dev))
    if z is None:
        z = _ZERO_BIAS[(n, dev)] = torch.zeros(n, dtype=torch.float32, device=dev)
    return z


def _gemm_rows(x2, Kp, tail_ok=False):
    """x2 as an A operand whose rows can be read Kp elements wide: 16-byte aligned rows, unit column stride, and either a
    row stride that covers Kp (the over-read stays inside the next columns of the same row, which zero weight columns
    ignore), or -- ``tail_ok``: the caller's buffer continues Kp - K readable elements past its last row, so an over-read
    runs into the next row (finite values times zero weights) -- or a zero-padded copy."""
    K = x2.shape[1]
    ok = x2.stride(1) == 1 and x2.stride(0) % 8 == 0 and x2.data_ptr() % 16 == 0
    # "the over-read stays inside the same row": the view's first column may sit at an offset c0 inside the parent's row
    # (a column slice), so the test is c0 + Kp <= row length, not row stride >= Kp -- else the last Kp - K elements come
    # from the next row (and from past the storage on the last row).  Non-finite caveat: inf / NaN times a zero weight is
    # NaN, so the over-read columns must hold finite values (x_dbl's B | C columns next to dt do; a step that has already
    # overflowed in f16 is skipped by the loss scaler whatever this GEMM returns).
    c0 = x2.storage_offset() % x2.stride(0) if x2.stride(0) > 0 else 0
    inside = x2.stride(0) >= Kp and c0 + Kp <= x2.stride(0)
    if inside:
        # c0 is the column offset only if the parent starts at a multiple of the row stride; whatever the parent, the
        # over-read of the LAST row must stay inside the storage
        last_end = x2.storage_offset() + (x2.shape[0] - 1) * x2.stride(0) + Kp
        inside = last_end * x2.element_size() <= x2.untyped_storage().nbytes()
    if ok and (K == Kp or inside or tail_ok):
        return x2
    if K == Kp:
        return x2.contiguous()
    return torch.nn.functional.pad(x2, (0, Kp - K))


def proj_fwd(x2, w, dt, out=None):
    """y [M, N] = x2 [M, K] @ w[N, K]^T on cum_gemm_nt (csrc/gemm.hip): the forward GEMM of a bias-free nn.Linear
    (upstream Mamba.forward's in_proj / x_proj / dt_proj / out_proj, reached from src/network/CleanUMamba.py:172-189).
    ``out``: a [M, >= N] view with unit column stride that receives the result (row stride = its own)."""
    N, K = w.shape
    Np, Kp = rup(N, 32), rup(K, bk_of(dt))
    wp = take(w, ("proj", (N, K), Np, Kp), lambda: lay_proj((N, K), Np, Kp), dt)
    a = _gemm_rows(x2, Kp)
    M = a.shape[0]
    if out is None:
        out = torch.empty(M, rup(N, 8), dtype=dt, device=a.device)[:, :N]
    # odd widths (pruned checkpoints): whole 16-byte groups are stored, the columns past N are the packed operand's
    # zero rows and land in the padding of the row
    ns = N if N % 8 == 0 else rup(N, 8)
    assert out.stride(0) >= ns
    gemm(a, 0, a.stride(0), wp, _zero_bias(Np, a.device), out, 0, out.stride(0), M, 1 << 30, 1 << 30, hip.EPI_BIAS, ns,
         split_k=N <= 256)
    return out


def proj_dgrad(dy2, w, dt, out=None, res=None, tail_ok=False):
    """dx [M, K] = dy2 [M, N] @ w[N, K] (+ res): the data gradient of the same layer, as the NT GEMM against w^T.
    ``out``: a [M, >= K] view that receives it; ``res`` [M, K] (row stride = its own) is added in the epilogue."""
    N, K = w.shape
    Np, Kp = rup(K, 32), rup(N, bk_of(dt))
    wp = take(w, ("proj_t", (N, K), Np, Kp), lambda: lay_proj_t((N, K), Np, Kp), dt)
    a = _gemm_rows(dy2, Kp, tail_ok)
    M = a.shape[0]
    if out is None:
        out = torch.empty(M, rup(K, 8), dtype=dt, device=a.device)[:, :K]
    ks = K if K % 8 == 0 else rup(K, 8)
    assert out.stride(0) >= ks and (res is None or K % 8 == 0)
    gemm(a, 0, a.stride(0), wp, _zero_bias(Np, a.device), out, 0, out.stride(0), M, 1 << 30, 1 << 30, hip.EPI_BIAS, ks,
         res=res, r_off=0, ldr=res.stride(0) if res is not None else 0, split_k=K <= 256 and res is None)
    return out


def clip_std(x, eps):
    """(B, 1, L) f32 -> (B, 1, 1): unbiased std of every clip + eps (csrc/loss.hip cum_clip_std; the reference's
    `noisy_audio.std(dim=2, keepdim=True) + 1e-3`, src/network/CleanUMamba.py:260-262)."""
    hip.require_gpu(x)
    B, _, L = x.shape
    x2 = x.reshape(B, L)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    lib = hip.lib()
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    part = torch.empty(3 * B * lib.cum_clip_std_parts(L), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        hip.check(lib.cum_clip_std(hip.ptr(x2), B, L, x2.stride(0), float(eps), hip.ptr(part), hip.ptr(out),
                                   hip.stream_ptr()))
    return out.view(B, 1, 1)


def frame_input(x, std, T, dtype):
    """(B, 1, L) f32 signal -> row buffer of Geo(B, T, 1): x / std, zero padding to T, zero rows and columns, in one
    launch (cum_frame_rows).  Not differentiable: the caller checks that the signal needs no gradient."""
    hip.require_gpu(x)
    B, _, L = x.shape
    geo = Geo(B, T, 1)
    x2 = x.reshape(B, L)
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    buf = geo.new(dtype, x.device)
    with torch.cuda.device(x.device):
        hip.check(hip.lib().cum_frame_rows(hip.dtype_code(dtype), hip.ptr(x2), B, L, x2.stride(0), T, geo.R,
                                           hip.ptr(std), 1, hip.ptr(buf), hip.stream_ptr()))
    return buf


class Unframe(torch.autograd.Function):
    """Row buffer of the 1-channel output -> (B, 1, L) f32, times the clip's std (`x[:, :, :L] * std`,
    src/network/CleanUMamba.py:319).  Backward: the gradient framed back into a row buffer (times std)."""

    @staticmethod
    def forward(ctx, buf, std, geo, L):
        y = torch.empty(geo.B, 1, L, dtype=torch.float32, device=buf.device)
        with torch.cuda.device(buf.device):
            hip.check(hip.lib().cum_unframe_rows(hip.dtype_code(buf.dtype), hip.ptr(buf), geo.B, L, geo.T, hip.ptr(std),
                                                 hip.ptr(y), hip.stream_ptr()))
        ctx.geo, ctx.dtype, ctx.std = geo, buf.dtype, std
        return y

    @staticmethod
    def backward(ctx, dy):
        geo = ctx.geo
        dy = dy.contiguous().float()
        B, _, L = dy.shape
        d = geo.new(ctx.dtype, dy.device)
        with torch.cuda.device(dy.device):
            hip.check(hip.lib().cum_frame_rows(hip.dtype_code(ctx.dtype), hip.ptr(dy), B, L, L, geo.T, geo.R,
                                               hip.ptr(ctx.std), 0, hip.ptr(d), hip.stream_ptr()))
        return d, None, None, None


class LpLoss(torch.autograd.Function):
    """mean |y - c|^p over all elements, p in {1, 2}: F.l1_loss / F.mse_loss of loss_fn (src/util/util.py:262-268) as
    two launches with a fixed summation order (csrc/loss.hip).  Gradient wrt y only (c is the target)."""

    @staticmethod
    @torch.amp.custom_fwd(device_type="cuda", cast_inputs=torch.float32)
    def forward(ctx, y, c, p):
        hip.require_gpu(y, c)
        if y.shape != c.shape:
            raise RuntimeError("lp loss: denoised and clean audio must have the same shape")
        y, c = y.contiguous(), c.contiguous()
        lib, n = hip.lib(), y.numel()
        out = torch.empty((), dtype=torch.float32, device=y.device)
        part = torch.empty(lib.cum_lp_loss_parts(n), dtype=torch.float32, device=y.device)
        with torch.cuda.device(y.device):
            hip.check(lib.cum_lp_loss_fwd(p, hip.ptr(y), hip.ptr(c), n, hip.ptr(part), hip.ptr(out), hip.stream_ptr()))
        ctx.save_for_backward(y, c)
        ctx.p = p
        return out

    @staticmethod
    @torch.amp.custom_bwd(device_type="cuda")
    def backward(ctx, g):
        if ctx.needs_input_grad[1]:
            raise NotImplementedError("lp loss: no gradient wrt the target signal")
        y, c = ctx.saved_tensors
        dy = torch.empty_like(y)
        g = g.float().contiguous()
        with torch.cuda.device(y.device):
            hip.check(hip.lib().cum_lp_loss_bwd(ctx.p, hip.ptr(y), hip.ptr(c), y.numel(), hip.ptr(g), hip.ptr(dy),
                                                hip.stream_ptr()))
        return dy, None, None


def supported(model):
    """The fused path covers the shipped configuration: kernel 4, stride 2, ungrouped convs, sigmoid GLU,
    no bypass channels."""
    if model.kernel_size != 4 or model.stride != 2:
        return False
    for enc, dec in zip(model.encoder, model.decoder):
        if enc[0].groups != 1 or enc[3].bypass_channels != 0 or dec[1].bypass_channels != 0:
            return False
        if not isinstance(enc[3].activation, torch.nn.Sigmoid) or not isinstance(dec[1].activation, torch.nn.Sigmoid):
            return False
    return True


# ===================================================================== whole-stack Functions
# The per-layer Functions above leave three elementwise passes per layer to the backward (ReLU gate, GLU
# backward, the add that merges a skip gradient with the gradient from the next layer).  Chaining the layers
# inside ONE backward lets each of them ride in the epilogue of the GEMM that produces its operand
# (csrc/gemm.hip EPI_MASK / EPI_GLU_BWD): per layer the backward is then 2 data-gradient GEMMs + 2 weight-gradient
# GEMMs and nothing else.  Forward kernels are the same as above.  Layers whose channel count is not a multiple
# of 16 after padding (pruned checkpoints) keep the unfused elementwise kernels.
_SIGN_MASK = os.environ.get("CUM_SIGN_MASK", "1") != "0"


def _glu_fwd(xbuf, w, b, gi, go, save_z):
    """1x1 conv + GLU.  Saved for the backward: only the gate pre-activation b ([go.M, G*16], output-column order);
    together with the output itself (kept anyway: it is the next layer's input) it determines the GLU backward,
    da = d*sig(b), db = d*y*(1 - sig(b)) -- half the bytes of the (a | b) pre-activation."""
    dt, dev = xbuf.dtype, xbuf.device
    H2, Cin, _ = w.shape
    G = (H2 // 2 + 15) // 16
    Kp = rup(gi.Cp, bk_of(dt))
    sh = tuple(w.shape)
    wp = take(w, ("glu_fwd", sh, G * 32, Kp), lambda: lay_glu_fwd(sh, G * 32, Kp), dt)
    bp = take(b, ("glu_vec", H2), lambda: lay_glu_vec(H2), torch.float32)
    ybuf = go.new(dt, dev)
    z = torch.empty(go.M, G * 16, dtype=dt, device=dev) if save_z else None
    gemm(xbuf, gi.Cp, gi.Cp, wp, bp, ybuf, go.Cp, go.Cp, go.M, go.P, go.T, hip.EPI_GLU, go.Cp,
         aux=z, x_off=0, ldz=G * 16, geo=go, gate_only=True)
    return ybuf, z


def _conv_relu_fwd(xbuf, w, b, gi, go, want_bits=False):
    """Conv1d(k4, s2) + ReLU.  ``want_bits``: also -> the SIGN of every output element, four channels per byte (row buffer
    geometry): the backward's ReLU gate then reads 1/8 of the bytes of the activation itself (EPI_MASK with mask_bits;
    the activation is still kept -- it is the 1x1's weight-gradient operand)."""
    dt, dev = xbuf.dtype, xbuf.device
    H = w.shape[0]
    Np, Kp = rup(H, 16), rup(4 * gi.Cp, bk_of(dt))
    sh = tuple(w.shape)
    wp = take(w, ("conv_fwd", sh, gi.Cp, Np, Kp), lambda: lay_conv_fwd(sh, gi.Cp, Np, Kp), dt)
    bp = take(b, ("vec", H, Np), lambda: lay_vec(H, Np), torch.float32)
    ybuf = go.new(dt, dev)
    if want_bits and _SIGN_MASK and dt in hip.HALF_TYPES and go.Cp % 16 == 0:
        bits = torch.empty(go.R * go.Cp // 4, dtype=torch.uint8, device=dev)
        gemm(xbuf, gi.Cp, 2 * gi.Cp, wp, bp, ybuf, go.Cp, go.Cp, go.M, go.P, go.T, hip.EPI_RELU, go.Cp, geo=go,
             aux=bits[go.Cp // 4:], x_off=0, ldz=go.Cp, mask_bits=True)
        return ybuf, bits
    gemm(xbuf, gi.Cp, 2 * gi.Cp, wp, bp, ybuf, go.Cp, go.Cp, go.M, go.P, go.T, hip.EPI_RELU, go.Cp, geo=go)
    return (ybuf, None) if want_bits else ybuf


def _convt_fwd(xbuf, w, b, skip, gi, go, relu):
    dt, dev = xbuf.dtype, xbuf.device
    Cout = w.shape[1]
    N = 2 * go.Cp
    Np, Kp = rup(N, 16), rup(2 * gi.Cp, bk_of(dt))
    sh = tuple(w.shape)
    wp = take(w, ("convt_fwd", sh, gi.Cp, go.Cp, Np, Kp), lambda: lay_convt_fwd(sh, gi.Cp, go.Cp, Np, Kp), dt)

    def bias_layout():
        out = torch.zeros(Np, dtype=torch.int64)
        out[:N].view(2, go.Cp)[:, :Cout] = _ids((Cout,))
        return out
    bp = take(b, ("convt_vec", Cout, go.Cp, Np), bias_layout, torch.float32)
    ybuf = go.new(dt, dev)
    keep = relu and skip is not None            # the ReLU mask is not recoverable from y + skip
    if keep and _SIGN_MASK:
        # only the sign of the activation is kept (row buffer geometry, 4 channels per byte): 1/8 of a bf16 copy
        act = torch.empty(go.R * go.Cp // 4, dtype=torch.uint8, device=dev)
        gemm(xbuf, 0, gi.Cp, wp, bp, ybuf, go.Cp, N, gi.M, gi.P, gi.T + 1, hip.EPI_RELU, N,
             res=skip, r_off=go.Cp, ldr=N, aux=act[go.Cp // 4:], x_off=0, ldz=N, geo=go, mask_bits=True)
        return ybuf, act
    act = go.new(dt, dev) if keep else None       # CUM_SIGN_MASK=0: a full copy of the activation (A/B timing)
    gemm(xbuf, 0, gi.Cp, wp, bp, ybuf, go.Cp, N, gi.M, gi.P, gi.T + 1, hip.EPI_RELU if relu else hip.EPI_BIAS, N,
         res=skip, r_off=go.Cp, ldr=N, aux=act, x_off=go.Cp, ldz=N, geo=go)
    return ybuf, (act if keep else (ybuf if relu else None))


def _glu_bwd(z, ybuf, dy, go):
    """Standalone GLU backward from the gate pre-activation z [go.M, G*16] and the layer's output row buffer ybuf;
    dy is a row buffer of geometry go; the result is [go.M, G*32] in the packed (16 a | 16 b) layout."""
    G = z.shape[1] // 16
    dz = torch.empty(go.M, G * 32, dtype=z.dtype, device=z.device)
    with torch.cuda.device(z.device):
        hip.check(hip.lib().cum_glu_bwd_gate(hip.dtype_code(z.dtype), go.M, G, go.Cp, hip.ptr(z), G * 16,
                                             hip.ptr(ybuf[1:]), go.Cp, hip.ptr(dy[1:]), go.Cp, hip.ptr(dz), G * 32,
                                             hip.stream_ptr()))
    return dz


def _glu_wgrad(dz, xbuf, w, gi, M, out=None):
    """Weight / bias gradient of a 1x1+GLU layer from dZ [M, G*32] and its input row buffer.  With ``out`` (an arena
    slot) the GEMM-layout results stay there for the stack's batched un-pack and nothing is returned."""
    G32 = dz.shape[1]
    H2 = w.shape[0]
    sh = tuple(w.shape)
    dwp, dbp = wgrad(dz, 0, G32, G32, xbuf, gi.Cp, gi.Cp, gi.Cp, M, out=out)
    if out is not None:
        return None, None
    db = take(dbp, ("glu_vec_unpack", H2), lambda: _invert(lay_glu_vec(H2), (H2,)))
    dw = take(dwp, ("glu_unpack", sh, G32, gi.Cp), lambda: _invert(lay_glu_fwd(sh, G32, gi.Cp), sh))
    return dw.to(w.dtype), db.to(w.dtype)


def _glu_dgrad_weights(w, gi, G32, dt):
    sh = tuple(w.shape)
    Nd, Kd = rup(gi.Cp, 16), rup(G32, bk_of(dt))
    return take(w, ("glu_dgrad", sh, Nd, Kd),
                lambda: torch.nn.functional.pad(lay_glu_fwd(sh, G32, Nd).t(), (0, Kd - G32)), dt)


_ENC_BITS = os.environ.get("CUM_ENC_BITS", "1") != "0"           # "0": the encoder's ReLU gates read the activation (A/B)
_ENC0_FUSED = os.environ.get("CUM_ENC0_FUSED", "1") != "0"     # "0": first encoder layer on the generic GEMM path (A/B)


def _enc0_ok(w1, w2, gi, gm, go, dt):
    """The fused first-layer kernels (csrc/enc0.hip): one input channel, 64 conv channels, 16-bit activations."""
    return (_ENC0_FUSED and dt in hip.HALF_TYPES and tuple(w1.shape) == (64, 1, 4) and tuple(w2.shape) == (128, 64, 1)
            and gi.C == 1 and gi.Cp == 8 and gm.C == 64 and go.C == 64 and w1.dtype == torch.float32)


def _enc0_w2p(w2, gm, dt):
    sh = tuple(w2.shape)
    G = (sh[0] // 2 + 15) // 16
    Kp = rup(gm.Cp, bk_of(dt))
    wp = take(w2, ("glu_fwd", sh, G * 32, Kp), lambda: lay_glu_fwd(sh, G * 32, Kp), dt)
    return wp


def _enc0_fwd(xbuf, w1, b1, w2, b2, gm, go, save_z):
    dt, dev = xbuf.dtype, xbuf.device
    wp = _enc0_w2p(w2, gm, dt)
    bp = take(b2, ("glu_vec", w2.shape[0]), lambda: lay_glu_vec(w2.shape[0]), torch.float32)
    ybuf = go.new(dt, dev)
    z = torch.empty(go.M, 64, dtype=dt, device=dev) if save_z else None
    with torch.cuda.device(dev):
        hip.check(hip.lib().cum_enc0_fwd(hip.dtype_code(dt), go.M, go.P, go.T, hip.ptr(xbuf), hip.ptr(w1.detach()),
                                         hip.ptr(b1.detach()), hip.ptr(wp), hip.ptr(bp), hip.ptr(ybuf), go.tail, hip.ptr(z),
                                         hip.stream_ptr()))
    return ybuf, z


_ENCH_FUSED = os.environ.get("CUM_ENCH_FUSED", "1") != "0"      # "0": the two GEMM launches (A/B timing)


def _ench_ok(w1, w2, gi, gm, go, dt):
    """The width-128 encoder layer (64 -> 128 -> 128: the second layer of E6 / E8) takes the fused forward kernel."""
    return (_ENCH_FUSED and dt in hip.HALF_TYPES and tuple(w1.shape) == (128, 64, 4) and tuple(w2.shape) == (256, 128, 1)
            and gi.Cp == 64 and gm.Cp == 128 and go.Cp == 128 and gm.P >= 3)


def _ench_fwd(xbuf, w1, b1, w2, b2, gi, gm, go, save_z):
    """-> (y1 buffer, sign nibbles, output buffer, gate): what _conv_relu_fwd(want_bits) + _glu_fwd produce, in one launch
    (csrc/ench.hip).  Without a backward to come neither the hidden activation nor the gate is stored."""
    dt, dev = xbuf.dtype, xbuf.device
    sh1, sh2 = tuple(w1.shape), tuple(w2.shape)
    w1p = take(w1, ("conv_fwd", sh1, gi.Cp, 128, 256), lambda: lay_conv_fwd(sh1, gi.Cp, 128, 256), dt)
    b1p = take(b1, ("vec", 128, 128), lambda: lay_vec(128, 128), torch.float32)
    w2p = take(w2, ("glu_fwd", sh2, 256, 128), lambda: lay_glu_fwd(sh2, 256, 128), dt)
    b2p = take(b2, ("glu_vec", 256), lambda: lay_glu_vec(256), torch.float32)
    ybuf = go.new(dt, dev)
    y1 = gm.new(dt, dev) if save_z else None
    bits = torch.empty(gm.R * gm.Cp // 4, dtype=torch.uint8, device=dev) if (save_z and _ENC_BITS and _SIGN_MASK) else None
    z = torch.empty(go.M, 128, dtype=dt, device=dev) if save_z else None
    with torch.cuda.device(dev):
        hip.check(hip.lib().cum_ench_fwd(
            hip.dtype_code(dt), go.M, go.P, go.T, hip.ptr(xbuf[1:]), gi.R - 1, hip.ptr(w1p), hip.ptr(b1p), hip.ptr(w2p),
            hip.ptr(b2p), hip.ptr(y1[1:]) if y1 is not None else None, gm.tail,
            hip.ptr(bits[gm.Cp // 4:]) if bits is not None else None, hip.ptr(ybuf[1:]), go.tail, hip.ptr(z),
            hip.stream_ptr()))
    return y1, bits, ybuf, z


_DECH_FUSED = os.environ.get("CUM_DECH_FUSED", "1") != "0"      # "0": the two GEMM launches (A/B timing)


def _dech_ok(w1, wt, skip, relu, gi, gg, go, dt):
    """The width-128 decoder layer (128 -> 128 -> 64, ReLU, skip: the second-to-last layer of E6 / E8) takes the fused
    forward kernel."""
    return (_DECH_FUSED and _SIGN_MASK and relu and skip is not None and dt in hip.HALF_TYPES
            and tuple(w1.shape) == (256, 128, 1) and tuple(wt.shape) == (128, 64, 4)
            and gi.Cp == 128 and gg.Cp == 128 and go.Cp == 64 and gg.P >= 3)


def _dech_fwd(ubuf, w1, b1, wt, bt, skip, gi, gg, go, save_z):
    """-> (g buffer, gate, output buffer, sign nibbles): what _glu_fwd + _convt_fwd produce, in one launch
    (csrc/dech.hip).  Without a backward to come neither g nor the gate nor the nibbles are stored."""
    dt, dev = ubuf.dtype, ubuf.device
    sh1, sht = tuple(w1.shape), tuple(wt.shape)
    w1p = take(w1, ("glu_fwd", sh1, 256, 128), lambda: lay_glu_fwd(sh1, 256, 128), dt)
    b1p = take(b1, ("glu_vec", 256), lambda: lay_glu_vec(256), torch.float32)
    wtp = take(wt, ("convt_fwd", sht, gg.Cp, go.Cp, 128, 256), lambda: lay_convt_fwd(sht, gg.Cp, go.Cp, 128, 256), dt)

    def bias_layout():
        out = torch.zeros(128, dtype=torch.int64)
        out.view(2, 64)[:, :64] = _ids((64,))
        return out
    btp = take(bt, ("convt_vec", 64, go.Cp, 128), bias_layout, torch.float32)
    ybuf = go.new(dt, dev)
    g = gg.new(dt, dev) if save_z else None
    z = torch.empty(gg.M, 128, dtype=dt, device=dev) if save_z else None
    act = torch.empty(go.R * go.Cp // 4, dtype=torch.uint8, device=dev) if save_z else None
    with torch.cuda.device(dev):
        hip.check(hip.lib().cum_dech_fwd(
            hip.dtype_code(dt), gg.M, gg.P, gg.T, hip.ptr(ubuf), gi.R, hip.ptr(w1p), hip.ptr(b1p), hip.ptr(wtp),
            hip.ptr(btp), hip.ptr(skip[1:]), hip.ptr(g), gg.tail, hip.ptr(z), hip.ptr(ybuf[1:]), go.tail,
            hip.ptr(act[go.Cp // 4:]) if act is not None else None, hip.stream_ptr()))
    return g, z, ybuf, act


def _enc0_bwd(dz, xbuf, w1, b1, w2, gm, go, slot_w1, slot_w2):
    """slot_* = (arena buffer, offset): weight / bias gradients of the conv and of the 1x1 in their arena layouts."""
    dt, dev = dz.dtype, dz.device
    wp = _enc0_w2p(w2, gm, dt)
    lib = hip.lib()
    ws = torch.empty(lib.cum_enc0_bwd_workspace_elems(go.M), dtype=torch.float32, device=dev)
    (a1, o1), (a2, o2) = slot_w1, slot_w2
    with torch.cuda.device(dev):
        hip.check(lib.cum_enc0_bwd(hip.dtype_code(dt), go.M, go.P, go.T, hip.ptr(dz), hip.ptr(xbuf), hip.ptr(w1.detach()),
                                   hip.ptr(b1.detach()), hip.ptr(wp), ctypes.c_void_p(a2.data_ptr() + 4 * o2),
                                   ctypes.c_void_p(a1.data_ptr() + 4 * o1), hip.ptr(ws), hip.stream_ptr()))


_DEC7_FUSED = os.environ.get("CUM_DEC7_FUSED", "1") != "0"     # "0": last decoder layer on the generic GEMM path (A/B)


def _dec7_ok(w1, wt, gi, gg, go, dt):
    """The fused last-layer kernels (csrc/dec7.hip): 64 -> 128 1x1 + GLU, then 64 -> 1 transposed conv, 16-bit activations."""
    return (_DEC7_FUSED and dt in hip.HALF_TYPES and tuple(w1.shape) == (128, 64, 1) and tuple(wt.shape) == (64, 1, 4)
            and gi.C == 64 and gg.C == 64 and go.C == 1 and go.Cp == 8 and gi.P >= 32 and w1.dtype == torch.float32
            and wt.dtype == torch.float32)


def _dec7_fwd(ubuf, w1, b1, wt, bt, gi, go):
    dt, dev = ubuf.dtype, ubuf.device
    wp = _enc0_w2p(w1, gi, dt)
    bp = take(b1, ("glu_vec", w1.shape[0]), lambda: lay_glu_vec(w1.shape[0]), torch.float32)
    ybuf = go.new(dt, dev)
    with torch.cuda.device(dev):
        hip.check(hip.lib().cum_dec7_fwd(hip.dtype_code(dt), gi.M, gi.P, gi.T, hip.ptr(ubuf), hip.ptr(wp), hip.ptr(bp),
                                         hip.ptr(wt.detach()), hip.ptr(bt.detach()), hip.ptr(ybuf), go.tail, hip.stream_ptr()))
    return ybuf


def _dec7_bwd(dy, ubuf, mask_bits, w1, b1, wt, gi, slot_w1, slot_wt):
    """-> (dU, dU gated by the ReLU below) as row buffers of geometry gi; the weight / bias gradients of the 1x1 and of
    the transposed conv go to their arena slots ((buffer, offset)) in the layouts cum_gemm_tn writes."""
    dt, dev = ubuf.dtype, ubuf.device
    wp = _enc0_w2p(w1, gi, dt)
    bp = take(b1, ("glu_vec", w1.shape[0]), lambda: lay_glu_vec(w1.shape[0]), torch.float32)
    lib = hip.lib()
    du, dpre = gi.new(dt, dev), gi.new(dt, dev)
    ws = torch.empty(lib.cum_dec7_bwd_workspace_elems(gi.M), dtype=torch.float32, device=dev)
    (a1, o1), (a2, o2) = slot_w1, slot_wt
    with torch.cuda.device(dev):
        hip.check(lib.cum_dec7_bwd(hip.dtype_code(dt), gi.M, gi.P, gi.T, hip.ptr(dy), hip.ptr(ubuf), hip.ptr(mask_bits),
                                   hip.ptr(wp), hip.ptr(bp), hip.ptr(wt.detach()), hip.ptr(du), hip.ptr(dpre), gi.tail,
                                   ctypes.c_void_p(a1.data_ptr() + 4 * o1), ctypes.c_void_p(a2.data_ptr() + 4 * o2),
                                   hip.ptr(ws), hip.stream_ptr()))
    return du, dpre


class EncoderStack(torch.autograd.Function):
    """x -> (x_1, ..., x_E): every encoder layer [Conv1d k4 s2, ReLU, Conv1d 1x1, GLU]
    (src/network/CleanUMamba.py:108-113) on row buffers.  geos[i] = (g_in, g_mid, g_out); params = w1, b1, w2, b2
    per layer."""

    @staticmethod
    def forward(ctx, xbuf, geos, save_z, *params):
        bufs, y1s, zs, bits = [xbuf], [], [], []
        for i, (gi, gm, go) in enumerate(geos):
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            assert gi.P == 2 * gm.P and gm.T == go.T and gi.C == w1.shape[1] and gm.C == w1.shape[0] == w2.shape[1]
            sb = None
            if i == 0 and _enc0_ok(w1, w2, gi, gm, go, xbuf.dtype):
                y1 = None                      # rebuilt from the input where the backward needs it (csrc/enc0.hip)
                y, z = _enc0_fwd(xbuf, w1, b1, w2, b2, gm, go, save_z)
            elif _ench_ok(w1, w2, gi, gm, go, xbuf.dtype):
                y1, sb, y, z = _ench_fwd(bufs[-1], w1, b1, w2, b2, gi, gm, go, save_z)    # one launch (csrc/ench.hip)
            else:
                # with a backward to come, the ReLU's sign bits ride along: its gate then reads 1/8 of y1's bytes
                y1, sb = _conv_relu_fwd(bufs[-1], w1, b1, gi, gm, want_bits=True) if (save_z and _ENC_BITS) else \
                    (_conv_relu_fwd(bufs[-1], w1, b1, gi, gm), None)
                y, z = _glu_fwd(y1, w2, b2, gm, go, save_z)
            bufs.append(y)
            y1s.append(y1)
            zs.append(z)
            bits.append(sb)
        ctx.geos, ctx.E, ctx.saved_z = geos, len(geos), save_z
        ctx.has_bits = [sb is not None for sb in bits]
        ctx.save_for_backward(*bufs, *y1s, *[z for z in zs if z is not None], *params, *[sb for sb in bits if sb is not None])
        return tuple(bufs[1:])

    @staticmethod
    def backward(ctx, *dys):
        if not ctx.saved_z:
            raise RuntimeError("EncoderStack was run without save_z; backward is unavailable")
        E, geos = ctx.E, ctx.geos
        t = ctx.saved_tensors
        bufs, y1s, zs, params = t[:E + 1], t[E + 1:2 * E + 1], t[2 * E + 1:3 * E + 1], t[3 * E + 1:7 * E + 1]
        kept = list(t[7 * E + 1:])
        bits = [kept.pop(0) if h else None for h in ctx.has_bits]
        dt, dev = bufs[0].dtype, bufs[0].device
        grads = [None] * (4 * E)
        dz, dx0 = None, None
        # arena slots: 2 i = conv of layer i (N = Cp_mid, K = 4 Cp_in), 2 i + 1 = its 1x1+GLU (N = G*32, K = Cp_mid)
        arena = None
        if _ARENA_BATCH:
            arena = _WgradArena([nk for i in range(E) for nk in ((geos[i][1].Cp, 4 * geos[i][0].Cp),
                                                                 (2 * zs[i].shape[1], geos[i][1].Cp))], dev)
        shapes = [tuple(p.shape) for p in params]

        def parts(lo, hi):                     # arena -> parameter index of layers [lo, hi)
            out = []
            for i in range(lo, hi):
                gi, gm, _ = geos[i]
                sh1, sh2 = shapes[4 * i], shapes[4 * i + 2]
                G32 = 2 * zs[i].shape[1]
                out.append(arena.dw_index(2 * i, _invert(lay_conv_fwd(sh1, gi.Cp, gm.Cp, 4 * gi.Cp), sh1)))
                out.append(arena.db_index(2 * i, torch.arange(1, sh1[0] + 1, dtype=torch.int64)))
                out.append(arena.dw_index(2 * i + 1, _invert(lay_glu_fwd(sh2, G32, gm.Cp), sh2)))
                out.append(arena.db_index(2 * i + 1, _invert(lay_glu_vec(sh2[0]), (sh2[0],))))
            return out
        key = ("enc", tuple(shapes), tuple((g[0].Cp, g[1].Cp) for g in geos), tuple(z.shape[1] for z in zs))
        sink = grad_sink(params) if arena is not None else None
        if sink is not None and not _contiguous_run(sink[2], shapes):
            sink = None
        # Data-parallel runs: the three deepest layers hold ~85 % of the stack's parameters and finish first, while the
        # (slow, HBM-bound) outer layers are still to come -- their gradients are unpacked and announced to the
        # exchange as soon as their GEMMs are enqueued instead of at the end of the stack.
        cut = E - 3 if E > 3 else 0
        early = (sink is not None and cut > 0 and getattr(sink[0], "early_announce", False)
                 and _contiguous_run(sink[2][4 * cut:], shapes[4 * cut:]) and _contiguous_run(sink[2][:4 * cut], shapes[:4 * cut]))

        def flush(lo, hi):
            flat, idx, offs = sink
            arena.unpack_into((key, lo, hi), lambda: parts(lo, hi), shapes[4 * lo:4 * hi], flat.grad, offs[4 * lo:4 * hi])
            flat.wrote(idx[4 * lo:4 * hi])
        for i in reversed(range(E)):
            gi, gm, go = geos[i]
            w1, b1, w2, b2 = params[4 * i:4 * i + 4]
            if dz is None:                     # top layer: its output gradient arrives from outside only
                if dys[i] is None:
                    raise RuntimeError("EncoderStack: the deepest output must be used")
                dz = _glu_bwd(zs[i], bufs[i + 1], dys[i].contiguous(), go)
            G32 = dz.shape[1]
            y1 = y1s[i]
            if y1 is None:                     # fused first layer: y1 was never stored
                if arena is not None and not ctx.needs_input_grad[0]:
                    _enc0_bwd(dz, bufs[0], w1, b1, w2, gm, go, arena.out(0), arena.out(1))
                    break
                y1 = _conv_relu_fwd(bufs[0], w1, b1, gi, gm)      # generic route (input gradient wanted): rebuild it
            grads[4 * i + 2], grads[4 * i + 3] = _glu_wgrad(dz, y1, w2, gm, go.M,
                                                            out=arena.out(2 * i + 1) if arena else None)
            # 1x1 data gradient, gated by the ReLU below it in the epilogue
            wt = _glu_dgrad_weights(w2, gm, G32, dt)
            dzc = gm.new(dt, dev)
            if bits[i] is not None:
                gemm(dz, 0, G32, wt, None, dzc, gm.Cp, gm.Cp, gm.M, gm.P, gm.T, hip.EPI_MASK, gm.Cp,
                     res=bits[i][gm.Cp // 4:], r_off=0, ldr=gm.Cp, geo=gm, mask_bits=True)
            else:
                gemm(dz, 0, G32, wt, None, dzc, gm.Cp, gm.Cp, gm.M, gm.P, gm.T, hip.EPI_MASK, gm.Cp,
                     res=y1, r_off=gm.Cp, ldr=gm.Cp, geo=gm)
            # conv weight gradient: X row m = the 4*Cp contiguous inputs of output row m
            sh = tuple(w1.shape)
            dwp, dbp = wgrad(dzc, gm.Cp, gm.Cp, gm.Cp, bufs[i], gi.Cp, 2 * gi.Cp, 4 * gi.Cp, gm.M,
                             out=arena.out(2 * i) if arena else None)
            if arena is None:
                grads[4 * i] = take(dwp, ("conv_unpack", sh, gi.Cp, gm.Cp),
                                    lambda: _invert(lay_conv_fwd(sh, gi.Cp, gm.Cp, 4 * gi.Cp), sh)).to(w1.dtype)
                grads[4 * i + 1] = dbp[:sh[0]].to(w1.dtype)
            dz = None
            if early and i == cut:
                flush(cut, E)
            if i == 0 and not ctx.needs_input_grad[0]:
                break
            # conv data gradient = transposed conv: pair row t' reads dzc rows t'-1, t'
            Nd, Kd = rup(2 * gi.Cp, 16), rup(2 * gm.Cp, bk_of(dt))
            wd = take(w1, ("conv_dgrad", sh, gi.Cp, gm.Cp, Nd, Kd), lambda: lay_conv_dgrad(sh, gi.Cp, gm.Cp, Nd, Kd), dt)
            ext = None if i == 0 or dys[i - 1] is None else dys[i - 1].contiguous()
            if i > 0 and gi.Cp % 16 == 0:
                # ... + the skip gradient, pushed through the GLU of layer i-1 in the epilogue: dZ_{i-1} directly
                dz = torch.empty(gi.M, 2 * gi.Cp, dtype=dt, device=dev)
                gemm(dzc, 0, gm.Cp, wd, None, dz, 0, 4 * gi.Cp, gm.M, gm.P, gm.T + 1, hip.EPI_GLU_BWD, 2 * gi.Cp,
                     res=ext, r_off=gi.Cp, ldr=2 * gi.Cp, aux=zs[i - 1], x_off=0, ldz=2 * gi.Cp,
                     aux2=bufs[i], y_off=gi.Cp, ldy=2 * gi.Cp, gate_only=True)
                continue
            dx = gi.new(dt, dev)
            gemm(dzc, 0, gm.Cp, wd, None, dx, gi.Cp, 2 * gi.Cp, gm.M, gm.P, gm.T + 1, hip.EPI_BIAS, 2 * gi.Cp, geo=gi)
            if i == 0:
                dx0 = dx
            else:
                dz = _glu_bwd(zs[i - 1], bufs[i], dx if ext is None else dx + ext, gi)
        if arena is not None:
            if sink is not None:           # straight into the flat gradient buffer: no AccumulateGrad adds
                flush(0, cut if early else E)
                grads = [None] * len(params)
            else:
                grads = [g.to(p.dtype) for g, p in zip(arena.unpack(key, lambda: parts(0, E), shapes), params)]
        return (dx0, None, None, *grads)


def _convt_bias_grad(bt, dbp, Cp, C, dtype):
    """Transposed-conv bias gradient = sum of the two halves of the paired-row bias gradient; one add straight into the
    flat gradient buffer when it takes it (returns None then), else the tensor for autograd."""
    sink = grad_sink([bt]) if (bt.is_leaf and bt.dtype == torch.float32) else None
    if sink is not None:
        flat, idx, offs = sink
        torch.add(dbp[:C], dbp[Cp:Cp + C], out=flat.grad[offs[0]:offs[0] + C])
        flat.wrote(idx)
        return None
    return (dbp[:Cp] + dbp[Cp:2 * Cp])[:C].to(dtype)


class DecoderStack(torch.autograd.Function):
    """u_0 -> u_E: every decoder layer [Conv1d 1x1, GLU, ConvTranspose1d k4 s2, (ReLU)] with the encoder skip added
    to its output (src/network/CleanUMamba.py:121-130, 313-316).  geos[j] = (g_in, g_glu, g_out); skips[j] is added
    to the output of layer j (None for the last); params = w1, b1, wt, bt per layer; ReLU on all but the last layer."""

    @staticmethod
    def forward(ctx, ubuf, geos, save_z, n_skips, *rest):
        E = len(geos)
        skips, params = list(rest[:n_skips]) + [None] * (E - n_skips), rest[n_skips:]
        us, gs, zs, acts = [ubuf], [], [], []
        fused_last = False
        for j, (gi, gg, go) in enumerate(geos):
            w1, b1, wt, bt = params[4 * j:4 * j + 4]
            assert gi.T == gg.T and go.P == 2 * gg.P and gg.C == wt.shape[0] and go.C == wt.shape[1]
            relu = j < E - 1
            # last layer (64 -> 1): g and the gate are rebuilt from u where the backward needs them (csrc/dec7.hip); the
            # fused backward takes the ReLU of the layer below as sign bits and writes into the stack's arena
            if (j == E - 1 and skips[j] is None and _dec7_ok(w1, wt, gi, gg, go, ubuf.dtype)
                    and (not save_z or (E >= 2 and skips[E - 2] is not None and _SIGN_MASK and _ARENA_BATCH))):
                us.append(_dec7_fwd(us[-1], w1, b1, wt, bt, gi, go))
                gs.append(None)
                zs.append(None)
                acts.append(None)
                fused_last = True
                continue
            if _dech_ok(w1, wt, skips[j], relu, gi, gg, go, ubuf.dtype):
                g, z, y, act = _dech_fwd(us[-1], w1, b1, wt, bt, skips[j], gi, gg, go, save_z)   # one launch (csrc/dech.hip)
            else:
                g, z = _glu_fwd(us[-1], w1, b1, gi, gg, save_z)
                y, act = _convt_fwd(g, wt, bt, skips[j], gg, go, relu)
            us.append(y)
            gs.append(g)
            zs.append(z)
            acts.append(act)
        ctx.geos, ctx.E, ctx.saved_z, ctx.n_skips = geos, E, save_z, n_skips
        ctx.has_act = [a is not None for a in acts]
        ctx.fused_last = fused_last
        ctx.save_for_backward(*us[:E], *gs, *zs, *[a for a in acts if a is not None], *params)
        return us[E]

    @staticmethod
    def backward(ctx, dy):
        if not ctx.saved_z:
            raise RuntimeError("DecoderStack was run without save_z; backward is unavailable")
        E, geos = ctx.E, ctx.geos
        t = ctx.saved_tensors
        us, gs, zs = t[:E], t[E:2 * E], t[2 * E:3 * E]
        n_act = sum(ctx.has_act)
        kept, params = list(t[3 * E:3 * E + n_act]), t[3 * E + n_act:]
        acts = [kept.pop(0) if h else None for h in ctx.has_act]
        dt, dev = us[0].dtype, us[0].device
        grads = [None] * (4 * E)
        dskips = [None] * ctx.n_skips
        gi, gg, go = geos[E - 1]
        dpre = dy.contiguous()
        if acts[E - 1] is not None:            # a ReLU on the last layer (not the reference's configuration)
            gated = go.new(dt, dev)
            with torch.cuda.device(dev):
                hip.check(hip.lib().cum_relu_bwd(hip.dtype_code(dt), go.M, go.Cp, hip.ptr(acts[E - 1][1:]), go.Cp,
                                                 hip.ptr(dpre[1:]), go.Cp, hip.ptr(gated[1:]), go.Cp, go.head, go.tail,
                                                 hip.stream_ptr()))
            dpre = gated
        du = None
        g32s = [32 * ((params[4 * j].shape[0] // 2 + 15) // 16) for j in range(E)]       # packed 1x1 rows (16 a | 16 b per 32)
        # arena slots: 2 j = 1x1+GLU of layer j (N = G*32, K = Cp_in), 2 j + 1 = its transposed conv (N = 2 Cp_out,
        # K = 2 Cp_glu)
        arena = None
        if _ARENA_BATCH:
            arena = _WgradArena([nk for j in range(E) for nk in ((g32s[j], geos[j][0].Cp),
                                                                 (2 * geos[j][2].Cp, 2 * geos[j][1].Cp))], dev)
        for j in reversed(range(E)):
            gi, gg, go = geos[j]
            w1, b1, wt, bt = params[4 * j:4 * j + 4]
            sht = tuple(wt.shape)
            if j == E - 1 and ctx.fused_last:
                (ab, ao), nk = arena.out(2 * j + 1), 2 * go.Cp * 2 * gg.Cp
                du, dpre = _dec7_bwd(dpre, us[j], acts[j - 1][gi.Cp // 4:], w1, b1, wt, gi, arena.out(2 * j), (ab, ao))
                grads[4 * j + 3] = _convt_bias_grad(bt, ab[ao + nk:ao + nk + 2 * go.Cp], go.Cp, sht[1], wt.dtype)
                if j - 1 < ctx.n_skips:
                    dskips[j - 1] = du
                continue
            # transposed-conv weight gradient: pair rows of dpre against the 2*Cp contiguous inputs (rows m-1, m)
            dwp, dbp = wgrad(dpre, go.Cp, 2 * go.Cp, 2 * go.Cp, gs[j], 0, gg.Cp, 2 * gg.Cp, gg.M,
                             out=arena.out(2 * j + 1) if arena else None)
            grads[4 * j + 3] = _convt_bias_grad(bt, dbp, go.Cp, sht[1], wt.dtype)
            if arena is None:
                grads[4 * j + 2] = take(dwp, ("convt_unpack", sht, gg.Cp, go.Cp),
                                        lambda: _invert(lay_convt_fwd(sht, gg.Cp, go.Cp, 2 * go.Cp, 2 * gg.Cp), sht)).to(wt.dtype)
            # its data gradient = strided conv of dpre (row t reads rows 2t..2t+3), through the GLU in the epilogue
            Nd, Kd = rup(gg.Cp, 16), rup(4 * go.Cp, bk_of(dt))
            wc = take(wt, ("convt_dgrad", sht, go.Cp, Nd, Kd), lambda: lay_convt_dgrad(sht, go.Cp, Nd, Kd), dt)
            z = zs[j]
            G32 = 2 * z.shape[1]
            if gg.Cp % 16 == 0:
                dz = torch.empty(gg.M, G32, dtype=dt, device=dev)
                gemm(dpre, go.Cp, 2 * go.Cp, wc, None, dz, 0, G32, gg.M, gg.P, gg.T, hip.EPI_GLU_BWD, gg.Cp,
                     aux=z, x_off=0, ldz=gg.Cp, aux2=gs[j], y_off=gg.Cp, ldy=gg.Cp, gate_only=True)
            else:
                dg = gg.new(dt, dev)
                gemm(dpre, go.Cp, 2 * go.Cp, wc, None, dg, gg.Cp, gg.Cp, gg.M, gg.P, gg.T, hip.EPI_BIAS, gg.Cp, geo=gg)
                dz = _glu_bwd(z, gs[j], dg, gg)
            grads[4 * j], grads[4 * j + 1] = _glu_wgrad(dz, us[j], w1, gi, gg.M, out=arena.out(2 * j) if arena else None)
            # 1x1 data gradient: ungated it is the gradient of u_j (and of the skip added into it); gated by the ReLU
            # of layer j-1 it is that layer's dpre -- both written by one epilogue
            w1t = _glu_dgrad_weights(w1, gi, G32, dt)
            du = gi.new(dt, dev)
            if j > 0 and acts[j - 1] is not None:
                dpre = gi.new(dt, dev)
                bits = acts[j - 1].dtype == torch.uint8          # sign array written by the forward
                gemm(dz, 0, G32, w1t, None, dpre, gi.Cp, gi.Cp, gi.M, gi.P, gi.T, hip.EPI_MASK, gi.Cp,
                     res=acts[j - 1][gi.Cp // 4:] if bits else acts[j - 1], r_off=0 if bits else gi.Cp, ldr=gi.Cp,
                     aux=du, x_off=gi.Cp, ldz=gi.Cp, geo=gi, mask_bits=bits)
            else:
                gemm(dz, 0, G32, w1t, None, du, gi.Cp, gi.Cp, gi.M, gi.P, gi.T, hip.EPI_BIAS, gi.Cp, geo=gi)
                dpre = du
            if j > 0 and j - 1 < ctx.n_skips:
                dskips[j - 1] = du
        if arena is not None:
            # w1, b1, wt per layer from one gather (bt is the sum of two halves of its slab: set above)
            shapes = [tuple(params[4 * j + k].shape) for j in range(E) for k in range(3)]

            def parts():
                out = []
                for j in range(E):
                    gi, gg, go = geos[j]
                    sh1, sht = shapes[3 * j], shapes[3 * j + 2]
                    G32 = g32s[j]
                    out.append(arena.dw_index(2 * j, _invert(lay_glu_fwd(sh1, G32, gi.Cp), sh1)))
                    out.append(arena.db_index(2 * j, _invert(lay_glu_vec(sh1[0]), (sh1[0],))))
                    out.append(arena.dw_index(2 * j + 1, _invert(lay_convt_fwd(sht, gg.Cp, go.Cp, 2 * go.Cp, 2 * gg.Cp), sht)))
                return out
            key = ("dec", tuple(shapes), tuple((g[0].Cp, g[1].Cp, g[2].Cp) for g in geos), tuple(g32s))
            three = [params[4 * j + k] for j in range(E) for k in range(3)]
            sink = grad_sink(three)
            if sink is not None:           # w1, b1, wt straight into the flat gradient buffer (bt keeps the autograd path)
                flat, idx, offs = sink
                # the transposed-conv biases sit between the gathered parameters: they must not be overwritten with
                # zeros by the gap fill, so the gather runs per contiguous run of gathered parameters
                order = sorted(range(len(three)), key=lambda q: offs[q])
                runs, cur = [], [order[0]]
                for q_prev, q in zip(order, order[1:]):
                    end_prev = offs[q_prev] + (_numel(shapes[q_prev]) + 3) // 4 * 4
                    if offs[q] == end_prev:
                        cur.append(q)
                    else:
                        runs.append(cur)
                        cur = [q]
                runs.append(cur)
                all_parts = None
                for r, run in enumerate(runs):
                    def run_parts(run=run):
                        nonlocal all_parts
                        if all_parts is None:
                            all_parts = parts()
                        return [all_parts[q] for q in run]
                    arena.unpack_into((key, r), run_parts, [shapes[q] for q in run], flat.grad, [offs[q] for q in run])
                flat.wrote(idx)
            else:
                un = arena.unpack(key, parts, shapes)
                for j in range(E):
                    for k in range(3):
                        grads[4 * j + k] = un[3 * j + k].to(params[4 * j + k].dtype)
        return (du, None, None, None, *dskips, *grads)
