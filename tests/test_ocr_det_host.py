"""Text detector, host side: program reader, oracle interpreter on the condensed fixtures, DB post-process."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.ppocr_det import run_graph, synthetic_weights
from vsr_amd import _lib
from vsr_amd.backend.tools import ocr_det
from vsr_amd.backend.tools.ocr import get_coordinates
from vsr_amd.backend.tools.paddle_graph import load_graph

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("fixture,nparams,nconv", [("ppocr_det_fast_graph.json", 1171721, 62), ("ppocr_det_graph.json", 21979682, 142)])
def test_program_reader_and_oracle(fixture, nparams, nconv):
    g = load_graph(os.path.join(GOLD, fixture))
    assert sum(int(np.prod(s)) for _, s in g.params.values()) == nparams
    assert sum(1 for k, *_ in g.ops if k in ("conv2d", "depthwise_conv2d")) == nconv
    y = run_graph(g, synthetic_weights(g), torch.randn(1, 3, 64, 96))
    assert y.shape == (1, 1, 64, 96) and float(y.min()) >= 0 and float(y.max()) <= 1
    assert load_graph(json.loads(json.dumps(g.to_json()))).ops == g.ops            # the condensed form round-trips


def test_min_area_rect_and_box_order():
    th = np.deg2rad(25.0)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    rect = np.array([[0, 0], [60, 0], [60, 20], [0, 20]], dtype=np.float64) @ R.T + [100, 50]
    pts = np.round(np.concatenate([rect, (rect[:, None] * 0.5 + rect[None] * 0.5).reshape(-1, 2)])).astype(int)
    corners, w, h = ocr_det.min_area_rect(pts)
    assert sorted([round(w), round(h)]) == pytest.approx([20, 60], abs=2)
    box = ocr_det._order_box(corners)
    assert box[0][0] <= box[1][0] and box[3][0] <= box[2][0] and box[0][1] <= box[3][1] and box[1][1] <= box[2][1]


def test_db_postprocess_on_a_synthetic_map():
    """two text lines at high probability, a weak blob (below box_thresh) and a speck (below min_size): boxes come back in
    source-image pixels, grown by area * 1.5 / perimeter, ordered tl, tr, br, bl -- and feed get_coordinates (tools/ocr.py)."""
    prob = np.zeros((136, 240), dtype=np.float32)
    prob[20:30, 40:160] = 0.9
    prob[60:72, 30:200] = 0.8
    prob[100:110, 50:90] = 0.45          # mean 0.45 < 0.6
    prob[120:122, 10:12] = 0.95          # too small
    boxes, scores = ocr_det.db_postprocess(prob, src_h=1088, src_w=1920)
    assert boxes.shape == (2, 4, 2) and len(scores) == 2
    sx, sy = 1920 / 240, 1088 / 136
    for b, (y0, y1, x0, x1) in zip(sorted(boxes.tolist(), key=lambda q: q[0][1]), ((20, 29, 40, 159), (60, 71, 30, 199))):
        w, h = x1 - x0, y1 - y0
        d = w * h * 1.5 / (2 * (w + h))
        exp = [(x0 - d) * sx, (x1 + d) * sx, (y0 - d) * sy, (y1 + d) * sy]
        xmin, xmax, ymin, ymax = get_coordinates([b])[0]
        assert [xmin, xmax, ymin, ymax] == pytest.approx(exp, abs=9)
    empty, _ = ocr_det.db_postprocess(np.zeros((64, 64), np.float32), 64, 64)
    assert empty.shape == (0, 4, 2)


def test_runner_needs_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU box")
    g = load_graph(os.path.join(GOLD, "ppocr_det_fast_graph.json"))
    with pytest.raises(_lib.VsrError) as ei:
        ocr_det.PaddleGraphRunner(g, synthetic_weights(g))
    assert ei.value.code == _lib.VSR_ERR_NOGPU


def _pb_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _write_pdiparams(path, arrays, packed=False, lod=False):
    """writer of the save_combine layout paddle_graph.read_pdiparams restates (test helper)"""
    import struct
    code = {"float32": 5, "float16": 4, "int64": 3, "float64": 6}
    with open(path, "wb") as f:
        for a in arrays:
            f.write(struct.pack("<I", 0))
            if lod:
                f.write(struct.pack("<QQ", 1, 16) + struct.pack("<QQ", 0, 3))
            else:
                f.write(struct.pack("<Q", 0))
            desc = b"\x08" + _pb_varint(code[str(a.dtype)])
            if packed:
                body = b"".join(_pb_varint(d) for d in a.shape)
                desc += b"\x12" + _pb_varint(len(body)) + body
            else:
                desc += b"".join(b"\x10" + _pb_varint(d) for d in a.shape)
            f.write(struct.pack("<Ii", 0, len(desc)) + desc + np.ascontiguousarray(a).tobytes())


@pytest.mark.parametrize("order,packed,lod", [("name", False, False), ("program", True, True)])
def test_pdiparams_reader_round_trip(tmp_path, order, packed, lod):
    from vsr_amd.backend.tools import paddle_graph
    g = paddle_graph.load_graph(os.path.join(GOLD, "ppocr_det_fast_graph.json"))
    w = synthetic_weights(g)
    plist = [g.params[k] for k in sorted(g.params)]
    if order == "name":
        plist = sorted(plist, key=lambda nv: nv[0])
    arrays = [np.asarray(w[name], dtype=np.float32) for name, _ in plist]
    arrays[3] = arrays[3].astype(np.float16)                         # a half-precision tensor is widened on read
    path = str(tmp_path / "inference.pdiparams")
    _write_pdiparams(path, arrays, packed=packed, lod=lod)
    got = paddle_graph.read_pdiparams(path, g)
    assert set(got) == {name for name, _ in plist}
    for (name, shape), a in zip(plist, arrays):
        assert got[name].dtype == np.float32 and got[name].shape == tuple(shape)
        np.testing.assert_array_equal(got[name], a.astype(np.float32))
    # a truncated stream and a wrong parameter count fail loudly
    with open(path, "rb") as f:
        raw = f.read()
    bad = str(tmp_path / "bad.pdiparams")
    with open(bad, "wb") as f:
        f.write(raw[: len(raw) // 2])
    with pytest.raises(ValueError):
        paddle_graph.read_pdiparams(bad, g)
    _write_pdiparams(bad, arrays[:-1])
    with pytest.raises(ValueError, match="parameters"):
        paddle_graph.read_pdiparams(bad, g)


def test_pdiparams_reader_on_literal_bytes(tmp_path):
    """A save_combine stream written out byte by byte from Paddle's published layout (paddle/fluid/framework/lod_tensor.cc
    SerializeToStream + tensor_util.cc TensorToStream + framework.proto VarType.TensorDesc {data_type = 1; dims = 2}), independent
    of the writer helper above: tensor 1 = fp32 [2,3] with one LoD level and unpacked dims, tensor 2 = fp16 [4] with packed dims."""
    from vsr_amd.backend.tools import paddle_graph

    t1 = bytes.fromhex(
        "00000000"                      # uint32 LoDTensor version 0
        "0100000000000000"              # uint64 lod levels = 1
        "1000000000000000"              # uint64 byte size of level 0 = 16
        "0000000000000000" "0300000000000000"    # size_t offsets {0, 3}
        "00000000"                      # uint32 tensor version 0
        "06000000"                      # int32 size of the TensorDesc message
        "0805" "1002" "1003"            # data_type = FP32 (5); dims = 2, 3 (field 2, unpacked varints)
    ) + np.array([[1.5, -2.0, 3.25], [0.0, 1e-3, -7.0]], "<f4").tobytes()
    t2 = bytes.fromhex(
        "00000000" "0000000000000000"   # version 0, no LoD
        "00000000" "05000000"           # tensor version 0, desc of 5 bytes
        "0804" "120104"                 # data_type = FP16 (4); dims packed: length 1, value 4
    ) + np.array([0.5, -1.0, 2.0, 65504.0], "<f2").tobytes()
    path = str(tmp_path / "lit.pdiparams")
    with open(path, "wb") as f:
        f.write(t1 + t2)
    a, b = paddle_graph.read_pdiparams_tensors(path)
    assert a.dtype == np.float32 and a.shape == (2, 3) and a.tolist() == [[1.5, -2.0, 3.25], [0.0, float(np.float32(1e-3)), -7.0]]
    assert b.dtype == np.float16 and b.shape == (4,) and b.astype(np.float32).tolist() == [0.5, -1.0, 2.0, 65504.0]


def test_det_resize_shape_and_same_padding():
    """DetResizeForTest as PaddleX's TextDetection applies it to the PP-OCRv5 det models (limit_side_len 960, limit_type "max":
    only shrink) and Paddle's padding_algorithm "SAME" for any stride"""
    from vsr_amd.backend.tools.ocr_det import det_resize_shape, same_padding

    assert det_resize_shape(1080, 1920) == (544, 960)            # int(1080 * 0.5) = 540 -> 544
    assert det_resize_shape(480, 852) == (480, 864)              # below the limit: no up-scaling, multiples of 32 only
    assert det_resize_shape(720, 1280) == (544, 960)
    assert det_resize_shape(2160, 3840) == (544, 960)
    assert det_resize_shape(100, 20) == (96, 32)
    assert det_resize_shape(480, 852, limit_type="long") == (544, 960)
    assert same_padding(17, 3, 1) == (1, 17) and same_padding(16, 3, 2) == (0, 8) and same_padding(17, 3, 2) == (1, 9)
    assert same_padding(10, 5, 1) == (2, 10) and same_padding(10, 2, 2) == (0, 5) and same_padding(7, 3, 1, 2) == (2, 7)


def test_from_env_without_weights_is_none(monkeypatch, tmp_path):
    from vsr_amd.backend.tools import ocr_det
    monkeypatch.delenv("VSR_DET_WEIGHTS", raising=False)
    monkeypatch.delenv("VSR_DET_MODEL_DIR", raising=False)
    assert ocr_det.from_env() is None
    monkeypatch.setenv("VSR_DET_MODEL_DIR", str(tmp_path))            # directory without inference.pdiparams
    assert ocr_det.from_env() is None


def test_conv_gemm_layout_against_conv2d():
    """the gather-GEMM description of a dense conv (offset tables + packed weights, ocr_det.conv_gemm_layout) replayed on the CPU with
    the descriptor semantics the kernels are tested against (tests/_replay.gemm_reference) equals F.conv2d"""
    import sys
    from types import SimpleNamespace

    import torch.nn.functional as F
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import _replay

    cases = [(65, 20, 28, 64, 3, 3, 1, False, (1, 1)), (40, 18, 25, 64, 9, 9, 1, True, None), (200, 9, 14, 256, 1, 1, 1, False, (0, 0)),
             (33, 21, 30, 48, 3, 3, 2, False, (1, 1)), (32, 15, 16, 32, 2, 2, 1, True, None), (96, 10, 12, 130, 3, 3, 1, False, (1, 1))]
    for ci_, (cin, h, w, cout, kh, kw, st, same, pad) in enumerate(cases):
        nb = 1 + ci_ % 3                                                         # batches of 1, 2 and 3 images
        rng = np.random.default_rng(cin + h)
        x = rng.standard_normal((nb, cin, h, w)).astype(np.float32)
        wt = rng.standard_normal((cout, cin, kh, kw)).astype(np.float32)
        if same:
            pt, pl, ho, wo = (kh - 1) // 2, (kw - 1) // 2, -(-h // st), -(-w // st)
            ref = F.conv2d(F.pad(torch.from_numpy(x), (pl, kw - 1 - pl, pt, kh - 1 - pt)), torch.from_numpy(wt), stride=st)
        else:
            pt, pl = pad
            ho, wo = (h + 2 * pt - kh) // st + 1, (w + 2 * pl - kw) // st + 1
            ref = F.conv2d(torch.from_numpy(x), torch.from_numpy(wt), stride=st, padding=pad)
        lay = ocr_det.conv_gemm_layout(cin, h, w, wt.shape, (st, st), pt, pl, ho, wo, n=nb)
        a = np.zeros((nb, lay["hp"], lay["wp"], lay["cp"]), np.float32)          # what k_det_nchw_to_nhwc writes
        a[:, pt:pt + h, pl:pl + w, :cin] = x.transpose(0, 2, 3, 1)
        t = lay["tables"]
        assert int(t["rowA"][:lay["M"]].max()) + int(t["colA"].max()) + 31 < a.size and lay["K"] % 32 == 0
        bufs = {1: a.reshape(-1), 2: ocr_det.pack_conv_weights(wt, lay["cp"]).reshape(-1), 3: np.zeros(lay["M"] * lay["ncs"], np.float32)}
        tables = [t[k].astype(np.int64) for k in ("rowA", "colA", "rowB", "colB", "rowC", "colC")]
        it = SimpleNamespace(M=lay["M"], N=lay["N"], K=lay["K"], bufA=1, bufB=2, bufC=3, bufR=-1, offA=0, offB=0, offC=0, offR=0, offBias=-1,
                             tRowA=0, tColA=1, tRowB=2, tColB=3, tRowC=4, tColC=5, tRowR=-1, splitK=1, chunksPerSplit=lay["K"] // 32,
                             splitStride=0, alpha=1.0, act=0)
        _replay.gemm_reference(it, 0, bufs, tables)
        out = bufs[3].reshape(nb, lay["P"], lay["ncs"])[:, :, :cout].transpose(0, 2, 1).reshape(nb, cout, ho, wo)   # what k_det_nhwc_to_nchw reads
        assert np.abs(out - ref.numpy()).max() <= 2e-5 * np.abs(ref.numpy()).max(), (cin, h, w, cout, kh, kw)


@pytest.mark.parametrize("fixture,nconv,min_affine", [("ppocr_det_fast_graph.json", 48, 40), ("ppocr_det_graph.json", 115, 100)])
def test_conv_fusion_analysis(fixture, nconv, min_affine):
    g = load_graph(os.path.join(GOLD, fixture))
    fuse = ocr_det.conv_fusions(g)
    assert len(fuse) == nconv == sum(1 for k, *_ in g.ops if k == "conv2d")
    readers = {}
    for kind, ins, outs, a in g.ops:
        for v in ins:
            readers[v] = readers.get(v, 0) + 1
    n_aff = 0
    for ci, (aff, act) in fuse.items():
        if aff is None:
            assert act is None
            continue
        n_aff += 1
        j = aff[1]
        assert g.ops[ci][2][0] in g.ops[j][1] and readers[g.ops[ci][2][0]] == 1
        assert (aff[0] == "bn" and g.ops[j][0] == "batch_norm_") or (aff[0] == "bias" and g.ops[j][0] == "add" and aff[2] in g.params
                                                                      and int(np.prod(g.params[aff[2]][1])) == g.params[g.ops[ci][1][1]][1][0])
        if act is not None:
            assert g.ops[act[0]][0] == {1: "relu", 2: "hardswish"}[act[1]] and g.ops[act[0]][1][0] == g.ops[j][2][0] and readers[g.ops[j][2][0]] == 1
    assert n_aff >= min_affine


def test_min_area_rect_is_minimal_and_encloses():
    """definition check, independent of any library: for random integer point clouds the rectangle encloses every point, and no
    orientation on a 0.02-degree grid gives a smaller bounding box (the minimum lies on a hull edge, so the grid can only lose)"""
    rng = np.random.default_rng(2)
    ang = np.deg2rad(np.arange(0.0, 90.0, 0.02))
    ca, sa = np.cos(ang), np.sin(ang)
    for trial in range(25):
        n = int(rng.integers(3, 60))
        th = rng.uniform(0, np.pi)
        base = rng.normal(size=(n, 2)) * [rng.uniform(5, 80), rng.uniform(2, 15)]
        pts = np.round(base @ np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]).T + [200, 100]).astype(int)
        corners, w, h = ocr_det.min_area_rect(pts)
        u = (corners[1] - corners[0]) / max(np.hypot(*(corners[1] - corners[0])), 1e-12)
        v = (corners[3] - corners[0]) / max(np.hypot(*(corners[3] - corners[0])), 1e-12)
        if w > 0 and h > 0:
            assert abs(u @ v) < 1e-9
            rel = pts - corners[0]
            assert (rel @ u >= -1e-6).all() and (rel @ u <= w + 1e-6).all() and (rel @ v >= -1e-6).all() and (rel @ v <= h + 1e-6).all()
        pu = pts[:, :1] * ca[None] + pts[:, 1:] * sa[None]
        pv = -pts[:, :1] * sa[None] + pts[:, 1:] * ca[None]
        brute = ((pu.max(0) - pu.min(0)) * (pv.max(0) - pv.min(0))).min()
        assert w * h <= brute + 1e-6, (trial, w * h, brute)
        assert w * h >= brute * (1 - 2e-3) - 1e-6, (trial, w * h, brute)


def test_transposed_conv_fusion_chains():
    """deconv_fusions (round 5: the 2x2 transposed conv as a gather-GEMM): the chains found in the two shipped programs -- the server
    head's 64 -> 64 deconv carries its bias add, batch_norm and ReLU; the 64 -> 1 one its bias add only (the sigmoid stays a kernel);
    every folded op has exactly one reader"""
    import os

    from vsr_amd.backend.tools.ocr_det import deconv_fusions
    from vsr_amd.backend.tools.paddle_graph import load_graph

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for fixture, first_cout in (("ppocr_det_graph.json", 64), ("ppocr_det_fast_graph.json", 24)):
        g = load_graph(os.path.join(gold, fixture))
        fz = deconv_fusions(g)
        assert len(fz) == 2
        (i0, (bias0, bn0, act0)), (i1, (bias1, bn1, act1)) = sorted(fz.items())
        assert g.params[g.ops[i0][1][1]][1][1] == first_cout and g.params[g.ops[i1][1][1]][1][1] == 1
        assert bias0 is not None and bn0 is not None and act0 is not None and act0[1] == 1
        assert g.ops[bias0[0]][0] == "add" and g.ops[bn0][0] == "batch_norm_" and g.ops[act0[0]][0] == "relu"
        assert i0 < bias0[0] < bn0 < act0[0]
        assert bias1 is not None and bn1 is None and act1 is None
