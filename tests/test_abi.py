"""The C-ABI library loads and exports every symbol include/cleanumamba_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cleanumamba_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cum_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from cleanumamba_amd import hip
    lib = hip.lib()
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
        assert s in hip.SIGNATURES, f"{s} has no ctypes signature"
    assert set(hip.SIGNATURES) <= set(syms)
    assert lib.cum_abi_version() == 15
    assert lib.cum_scan_chunk() == 16


def test_size_queries_and_argument_errors():
    from cleanumamba_amd import hip
    lib = hip.lib()
    # two saved states (one per 8-step half) per 16-step chunk, states padded to whole 8-state wave slices
    assert lib.cum_scan_ckpt_elems(2, 8, 8, 33) == 2 * 2 * 3 * 8 * 8
    assert lib.cum_scan_ckpt_elems(1, 5, 13, 16) == 2 * 1 * 1 * 2 * 5 * 8
    assert lib.cum_scan_ckpt_elems(1, 70, 64, 16) == 2 * 1 * 1 * 8 * 128 * 8      # d_state > 16: whole 64-channel groups
    assert lib.cum_scan_bwd_workspace_elems(2, 100, 13, 7) == 2 * 100 * 13 + 2 * 2 * 100 + 2 * 2 * 2 * 7 * 13
    assert lib.cum_conv_bwd_workspace_elems(2, 10, 33, 4) == 2 * 3 * 5 * 10
    # time-parallel forward scan: a workspace is asked for only where the sequential grid leaves the chip mostly idle
    assert lib.cum_scan_fwd_workspace_elems(16, 2048, 64, 624) == 0             # E8 training shape: 4 096 waves
    # batch-1 E8: 256 waves -> 8 segments of 5 chunks (one workgroup per CU); per segment 8 waves x 2048 channels x 8
    # states + 2048 sums of delta
    assert lib.cum_scan_fwd_workspace_elems(1, 2048, 64, 624) == 8 * (8 * 2048 * 8 + 2048)
    assert lib.cum_scan_fwd_workspace_elems(1, 128, 16, 61) == 0                # four chunks: nothing to split
    # bad arguments are rejected before any launch (works without a GPU)
    s = hip.ScanShape()
    s.batch, s.dim, s.dstate, s.len = 1, 4, 200, 4
    rc = lib.cum_selective_scan_fwd(ctypes.byref(s), *([None] * 12))
    assert rc == -1 and b"d_state" in lib.cum_last_error()
    c = hip.ConvShape()
    c.batch, c.dim, c.len, c.width = 1, 4, 4, 9
    assert lib.cum_causal_conv1d_fwd(ctypes.byref(c), *([None] * 5)) == -1
