"""ctypes binding of libcleanumamba_hip.so (C ABI in include/cleanumamba_hip.h).

This is the only place Python touches the native library.  There is NO fallback:
if the library is missing, or a tensor is not on a GPU, the call raises.  The
signatures carry plain pointers and sizes; torch is used only to own device
memory and to name the current HIP stream.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CUM_LIB") or os.path.join(_HERE, "libcleanumamba_hip.so")   # CUM_LIB: another build (A/B)

c_i32, c_i64, c_f32p, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p


class ScanShape(ctypes.Structure):
    _fields_ = [("batch", c_i32), ("dim", c_i32), ("dstate", c_i32), ("len", c_i32),
                ("u_sb", c_i64), ("u_sd", c_i64), ("u_sl", c_i64),
                ("dt_sb", c_i64), ("dt_sd", c_i64), ("dt_sl", c_i64),
                ("z_sb", c_i64), ("z_sd", c_i64), ("z_sl", c_i64),
                ("o_sb", c_i64), ("o_sd", c_i64), ("o_sl", c_i64),
                ("B_sb", c_i64), ("B_sn", c_i64), ("B_sl", c_i64),
                ("C_sb", c_i64), ("C_sn", c_i64), ("C_sl", c_i64),
                ("delta_softplus", c_i32), ("io_dtype", c_i32)]


class ScanGradStrides(ctypes.Structure):
    _fields_ = [("du_sb", c_i64), ("du_sd", c_i64), ("du_sl", c_i64),
                ("dd_sb", c_i64), ("dd_sd", c_i64), ("dd_sl", c_i64),
                ("dz_sb", c_i64), ("dz_sd", c_i64), ("dz_sl", c_i64)]


class ConvShape(ctypes.Structure):
    _fields_ = [("batch", c_i32), ("dim", c_i32), ("len", c_i32), ("width", c_i32),
                ("x_sb", c_i64), ("x_sd", c_i64), ("x_sl", c_i64),
                ("y_sb", c_i64), ("y_sd", c_i64), ("y_sl", c_i64),
                ("silu", c_i32), ("io_dtype", c_i32)]


class GemmDesc(ctypes.Structure):
    _fields_ = [("dtype", c_i32), ("epilogue", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("lda", c_i64), ("ldw", c_i64), ("ldc", c_i64), ("ldr", c_i64), ("ldz", c_i64),
                ("pitch", c_i32), ("valid", c_i32), ("n_store", c_i32), ("zero_head", c_i64), ("zero_tail", c_i64),
                ("gate_only", c_i32), ("ldy", c_i64), ("mask_bits", c_i32), ("allow_split_k", c_i32)]


CUM_F32, CUM_BF16, CUM_F16 = 0, 1, 2
HALF_TYPES = (torch.bfloat16, torch.float16)          # 16-bit element types the kernels read / write directly
IO_TYPES = (torch.float32,) + HALF_TYPES
EPI_BIAS, EPI_RELU, EPI_GLU, EPI_MASK, EPI_GLU_BWD = 0, 1, 2, 3, 4

# name -> (restype, argtypes); mirrors include/cleanumamba_hip.h one to one.
_P = ctypes.c_void_p
SIGNATURES = {
    "cum_abi_version": (c_i32, []),
    "cum_last_error": (ctypes.c_char_p, []),
    "cum_scan_chunk": (c_i32, []),
    "cum_scan_ckpt_elems": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "cum_selective_scan_fwd": (c_i32, [ctypes.POINTER(ScanShape)] + [_P] * 12),
    "cum_scan_fwd_workspace_elems": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "cum_selective_scan_fwd_ws": (c_i32, [ctypes.POINTER(ScanShape)] + [_P] * 14),
    "cum_scan_fwd_keeps_y": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32]),
    "cum_scan_bwd_workspace_elems": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "cum_selective_scan_bwd": (c_i32, [ctypes.POINTER(ScanShape), ctypes.POINTER(ScanGradStrides)] + [_P] * 21),
    "cum_scan_bwd_tp_workspace_elems": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "cum_selective_scan_bwd_tp": (c_i32, [ctypes.POINTER(ScanShape), ctypes.POINTER(ScanGradStrides)] + [_P] * 21),
    "cum_selective_state_update": (c_i32, [c_i32, c_i32, c_i32, _P, _P, _P, _P, _P, c_i64, _P, c_i64,
                                           _P, _P, _P, c_i32, _P, _P]),
    "cum_causal_conv1d_fwd": (c_i32, [ctypes.POINTER(ConvShape)] + [_P] * 5),
    "cum_conv_bwd_workspace_elems": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "cum_causal_conv1d_bwd": (c_i32, [ctypes.POINTER(ConvShape)] + [_P] * 5 + [c_i64] * 3 + [_P] * 4),
    "cum_causal_conv1d_update": (c_i32, [c_i32, c_i32, c_i32, _P, _P, _P, _P, c_i32, _P, _P]),
    "cum_gemm_nt": (c_i32, [ctypes.POINTER(GemmDesc)] + [_P] * 8),
    "cum_gemm_nt_tile": (c_i32, [ctypes.POINTER(GemmDesc)]),
    "cum_gemm_tn_tile": (c_i32, [c_i32, c_i64, c_i32, c_i32]),
    "cum_glu_bwd_gate": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, c_i64, _P, c_i64, _P, c_i64, _P, c_i64, _P]),
    "cum_glu_bwd": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, c_i64, _P, c_i64, _P, _P]),
    "cum_relu_bwd": (c_i32, [c_i32, c_i64, c_i32, _P, c_i64, _P, c_i64, _P, c_i64, c_i64, c_i64, _P]),
    "cum_colsum_workspace_elems": (c_i64, [c_i64, c_i32]),
    "cum_colsum": (c_i32, [c_i32, c_i64, c_i32, _P, c_i64, _P, _P, _P]),
    "cum_gemm_tn_workspace_elems": (c_i64, [c_i32, c_i64, c_i32, c_i32]),
    "cum_gemm_tn": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, c_i64, _P, c_i64, _P, c_i64, _P, _P, _P]),
    "cum_stft_frames": (c_i32, [_P, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, _P, _P, c_i64, _P]),
    "cum_stft_loss_workspace_elems": (c_i64, [c_i64, c_i64]),
    "cum_stft_loss_fwd": (c_i32, [_P, _P, c_i64, c_i64, c_i32, c_i64, _P, _P, _P]),
    "cum_stft_loss_bwd": (c_i32, [_P, _P, c_i64, c_i64, c_i32, c_i64, _P, _P, _P, _P, _P]),
    "cum_stft_fold": (c_i32, [_P, c_i64, c_i64, c_i32, c_i32, c_i32, _P, c_i64, _P, c_i64, c_i32, _P]),
    "cum_gather": (c_i32, [c_i32, _P, _P, c_i64, c_i32, _P, _P]),
    "cum_pack2d": (c_i32, [_P, _P, _P, c_i32, _P, c_i32, _P, _P]),
    "cum_add_layernorm_fwd": (c_i32, [c_i32, c_i32, c_i64, c_i32, c_i32, _P, c_i64, c_i64, _P, _P, _P, ctypes.c_float,
                                      _P, _P, _P, _P, _P]),
    "cum_add_layernorm_bwd_workspace_elems": (c_i64, [c_i32]),
    "cum_add_layernorm_bwd": (c_i32, [c_i32, c_i32, c_i64, c_i32] + [_P] * 12),
    "cum_small_linear": (c_i32, [c_i32, c_i32, c_i32, _P, c_i64, _P, _P, _P, c_i64, _P]),
    "cum_mamba_step_supported": (c_i32, [c_i32] * 5),
    "cum_mamba_step": (c_i32, [c_i32] * 6 + [ctypes.c_float] + [_P] * 20),
    "cum_stream_window_update": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, _P, _P, c_i64, c_i64, c_i32, _P, _P, c_i64, _P]),
    "cum_stream_tail_rows": (c_i32, [c_i32, c_i32, c_i32, c_i32, _P, c_i64, c_i32, _P, c_i64, _P]),
    "cum_stream_overlap_add": (c_i32, [c_i32, c_i32, c_i32, c_i32, c_i32, _P, c_i64, _P, _P, _P, c_i64, _P, c_i64, c_i32, _P]),
    "cum_stream_hop_plan_ints": (c_i32, []),
    "cum_stream_hop_max_lds_bytes": (c_i32, []),
    "cum_stream_hop": (c_i32, [_P, _P, _P, c_i64, c_i32, _P, c_i64, _P, c_i64, c_i32, c_i32, _P]),
    "cum_fft_plan_create": (c_i32, [c_i32, c_i32, c_i64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(c_i64)]),
    "cum_fft_plan_destroy": (c_i32, [_P]),
    "cum_fft_exec": (c_i32, [_P, _P, _P, c_i32, _P, _P]),
    "cum_stft_fused_supported": (c_i32, [c_i32]),
    "cum_stft_fused_workspace_elems": (c_i64, [c_i64, c_i64]),
    "cum_stft_fused_fwd": (c_i32, [_P, _P, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, _P, _P, c_i64, c_i64, _P, _P, _P]),
    "cum_stft_fused_bwd": (c_i32, [_P, _P, c_i64, c_i64, c_i64, c_i64, c_i32, c_i32, c_i32, _P, _P, c_i64, c_i64, _P, _P, _P,
                                   _P, _P]),
    "cum_stft_loss_fwd_packed": (c_i32, [_P, _P, c_i64, c_i64, c_i32, c_i64, _P, _P, _P, _P]),
    "cum_stft_loss_bwd_packed": (c_i32, [_P, _P, c_i64, c_i64, c_i32, c_i64, _P, _P, _P, _P, _P, _P]),
    "cum_enc0_fwd": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, _P, _P, _P, _P, _P, c_i64, _P, _P]),
    "cum_ench_fwd": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, c_i64, _P, _P, _P, _P, _P, c_i64, _P, _P, c_i64, _P, _P]),
    "cum_dech_fwd": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, c_i64, _P, _P, _P, _P, _P, _P, c_i64, _P, _P, c_i64, _P, _P]),
    "cum_enc0_bwd_workgroups": (c_i32, [c_i64]),
    "cum_enc0_bwd_workspace_elems": (c_i64, [c_i64]),
    "cum_enc0_bwd": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "cum_dec7_fwd": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, _P, _P, _P, _P, _P, c_i64, _P]),
    "cum_dec7_bwd_workgroups": (c_i32, [c_i64]),
    "cum_dec7_bwd_workspace_elems": (c_i64, [c_i64]),
    "cum_dec7_bwd": (c_i32, [c_i32, c_i64, c_i32, c_i32, _P, _P, _P, _P, _P, _P, _P, _P, c_i64, _P, _P, _P, _P]),
    "cum_lp_loss_parts": (c_i32, [c_i64]),
    "cum_lp_loss_fwd": (c_i32, [c_i32, _P, _P, c_i64, _P, _P, _P]),
    "cum_lp_loss_bwd": (c_i32, [c_i32, _P, _P, c_i64, _P, _P, _P]),
    "cum_clip_std_parts": (c_i32, [c_i64]),
    "cum_clip_std": (c_i32, [_P, c_i32, c_i64, c_i64, ctypes.c_float, _P, _P, _P]),
    "cum_frame_rows": (c_i32, [c_i32, _P, c_i32, c_i64, c_i64, c_i64, c_i64, _P, c_i32, _P, _P]),
    "cum_unframe_rows": (c_i32, [c_i32, _P, c_i32, c_i64, c_i64, _P, _P, _P]),
    "cum_optim_state_elems": (c_i32, []),
    "cum_optim_sumsq_parts": (c_i32, [c_i64]),
    "cum_optim_sumsq": (c_i32, [_P, c_i64, _P, _P]),
    "cum_optim_prepare": (c_i32, [_P, _P, c_i32, ctypes.c_float, ctypes.c_double, ctypes.c_double, c_i32, ctypes.c_float,
                                  ctypes.c_float, c_i32, _P]),
    "cum_optim_adam": (c_i32, [_P, _P, _P, _P, c_i64, _P, ctypes.c_double, ctypes.c_double, ctypes.c_float, ctypes.c_float,
                               _P]),
}

_lib = None


def lib():
    """Load (once) and return the native library; raise if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C cleanumamba_amd/csrc`). There is no CPU or PyTorch fallback for the hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.cum_abi_version() != 14:
            raise RuntimeError("libcleanumamba_hip.so ABI version mismatch")
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise RuntimeError(f"cleanumamba_hip error {rc}: {lib().cum_last_error().decode()}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def dtype_code(dtype):
    if dtype == torch.float32:
        return CUM_F32
    if dtype == torch.bfloat16:
        return CUM_BF16
    if dtype == torch.float16:
        return CUM_F16
    raise RuntimeError(f"cleanumamba_amd kernels take float32, bfloat16 or float16 (got {dtype})")


def require_gpu(*tensors, any_dtype=False):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("cleanumamba_amd: the hot path runs only on a ROCm GPU (got a %s tensor); "
                               "there is no CPU fallback" % t.device)
        if not any_dtype and t.dtype != torch.float32:
            raise RuntimeError("cleanumamba_amd kernels take float32 tensors (got %s)" % t.dtype)
        dev = t.device if dev is None else dev
        if t.device != dev:
            raise RuntimeError("cleanumamba_amd: tensors on different devices")
    return dev


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def scan_chunk():
    return lib().cum_scan_chunk()


# ---- FFT plans: the library holds no plan cache (include/cleanumamba_hip.h); the caller -- this module -- owns the plan
# objects and their work areas.  One plan per (device, kind, n, batch); a plan's work area is a tensor kept beside it, so
# its address is stable across hipGraph replays.
FFT_R2C, FFT_C2R, FFT_C2C = 0, 1, 2
_fft_plans = {}
_FFT_PLAN_LIMIT = 32          # distinct (device, kind, n, batch) plans kept; each holds a hipFFT work area


class _FftPlan:
    def __init__(self, kind, n, batch, device):
        self.handle, wb = ctypes.c_void_p(), c_i64()
        with torch.cuda.device(device):
            check(lib().cum_fft_plan_create(kind, n, batch, ctypes.byref(self.handle), ctypes.byref(wb)))
            self.work = torch.empty(max(int(wb.value), 16), dtype=torch.uint8, device=device) if wb.value > 0 else None

    def __del__(self):
        try:
            if self.handle:
                lib().cum_fft_plan_destroy(self.handle)
        except Exception:          # noqa: BLE001 - interpreter shutdown
            pass


def fft(kind, n, batch, src, dst, inverse=False):
    """One batched transform through a cached, caller-owned plan (src, dst: float32 tensors, see cum_fft_exec)."""
    dev = src.device
    if int(batch) == 0:                 # empty batch: nothing to transform (cum_fft_plan_create wants batch >= 1)
        return
    key = (dev.index, kind, int(n), int(batch))
    plan = _fft_plans.get(key)
    if plan is None:
        if torch.cuda.is_current_stream_capturing():
            # a plan's work area must not come out of a capture's private pool (it outlives the graph): create plans in
            # the warm-up steps
            raise RuntimeError("cleanumamba_amd.hip.fft: first use of an FFT plan inside hipGraph capture; run one "
                               "un-captured step first")
        if len(_fft_plans) >= _FFT_PLAN_LIMIT:      # bounded: drop the oldest plan (and its work area)
            _fft_plans.pop(next(iter(_fft_plans)))
        plan = _fft_plans[key] = _FftPlan(kind, int(n), int(batch), dev)
    with torch.cuda.device(dev):
        check(lib().cum_fft_exec(plan.handle, ptr(src), ptr(dst), int(bool(inverse)), ptr(plan.work), stream_ptr()))
