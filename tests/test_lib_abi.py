"""The C-ABI library loads without a GPU and exports exactly what include/vsr_hip.h declares."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "vsr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vsr_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    names = _declared_functions()
    assert len(names) >= 35
    for n in names:
        assert hasattr(built_lib.lib, n), f"{n} declared in include/vsr_hip.h but not exported"
        assert n in built_lib.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(built_lib.SIGNATURES) == names


def test_no_cpu_fallback(built_lib):
    """Without a device the compute entry points must fail loudly, never compute on the CPU."""
    import ctypes as C

    lib = built_lib.lib
    if lib.vsr_device_count() > 0:
        pytest.skip("GPU present")
    from vsr_amd.synth import make_state_dict
    from vsr_amd.engine import SttnEngine

    with pytest.raises(built_lib.VsrError):
        SttnEngine(make_state_dict(0), "auto", device=0)
    eng = SttnEngine(make_state_dict(0), "auto", device=None)     # host-side pack only
    buf = np.zeros(16, dtype=np.uint8)
    rc = lib.vsr_sttn_inpaint(eng.handle, buf.ctypes.data_as(C.c_void_p), 1, buf.ctypes.data_as(C.c_void_p), None, None)
    assert rc == built_lib.VSR_ERR_NOGPU
    assert "no CPU fallback" in built_lib.last_error()
    prob = built_lib.GGProblem()
    assert lib.vsr_run_gather_gemm(C.byref(prob), 1, 0, 0, None) == built_lib.VSR_ERR_NOGPU
    # the chunk entry points with the promises about the mask (rows, rows + columns) are no different
    ar = np.array([[0, 8, 0, 16]], np.int32)
    rc2 = np.array([[0, 8]], np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.vsr_sttn_auto_chunk_rows(eng.handle, P(buf), 1, 8, 16, P(buf), 1, P(ar), P(rc2), None, 0, None) == built_lib.VSR_ERR_NOGPU
    assert lib.vsr_sttn_auto_chunk_box(eng.handle, P(buf), 1, 8, 16, P(buf), 1, P(ar), P(rc2), P(rc2), None, 0, None) == built_lib.VSR_ERR_NOGPU
    eng.close()


def test_strict_state_dict(built_lib):
    from vsr_amd.synth import make_state_dict
    from vsr_amd.engine import SttnEngine

    sd = make_state_dict(0)
    assert len(sd) == 112 and sum(v.size for v in sd.values()) == 16556163
    bad = dict(sd)
    bad.pop("decoder.6.bias")
    with pytest.raises(built_lib.VsrError, match="missing key"):
        SttnEngine(bad, "auto", device=None)
    bad = dict(sd)
    bad["encoder.0.weight"] = np.zeros((64, 4, 3, 3), np.float32)
    with pytest.raises(built_lib.VsrError, match="shape mismatch"):
        SttnEngine(bad, "auto", device=None)
    bad = dict(sd)
    bad["module.extra"] = np.zeros(3, np.float32)
    with pytest.raises(built_lib.VsrError, match="unexpected key"):
        SttnEngine(bad, "auto", device=None)


def test_weight_packing_layout(built_lib):
    """[Cout][Cin][3][3] -> [Cout][K] with k = ((ci//32)*9 + ky*3+kx)*32 + ci%32 (channel-chunk major,
    taps inner: the default VSR_CONV_KORDER=1); the 3-channel first layer keeps k = tap*3 + c."""
    from vsr_amd.synth import make_state_dict
    from vsr_amd.engine import SttnEngine

    sd = make_state_dict(5)
    eng = SttnEngine(sd, "auto", device=None)
    packed = eng.packed_weights()
    w = sd["encoder.2.weight"]                       # first packed tensors: encoder.0 (K 27->32), bias, encoder.2
    off = 64 * 32 + 64
    got = packed[off: off + 64 * 576].reshape(64, 2, 9, 32)
    assert np.array_equal(got, w.reshape(64, 2, 32, 9).transpose(0, 1, 3, 2))
    w0 = sd["encoder.0.weight"]
    got0 = packed[: 64 * 32].reshape(64, 32)
    assert np.array_equal(got0[:, :27], w0.transpose(0, 2, 3, 1).reshape(64, 27))
    assert not got0[:, 27:].any()
    eng.close()
