"""The NHWC-resident plan of the text detector (vsr_amd/backend/tools/ocr_det_nhwc.py) on the CPU: the compiled steps --
GEMM offset tables, folded batch_norm / bias / ReLU, residual adds, upsampled residual rows, concat slices, halos -- executed by
tests/_det_replay.py with the launchers' semantics and held to the program interpreter (oracle/ppocr_det.py).  No GPU: the same plan is
what ocr_det.PaddleGraphRunner.run_planned uploads and launches (tests/test_gpu_ocr_det.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle.ppocr_det import run_graph, synthetic_weights
from vsr_amd.backend.tools import ocr_det_nhwc as nhwc
from vsr_amd.backend.tools.paddle_graph import load_graph

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _det_replay  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _compiled(fixture, shape):
    g = load_graph(os.path.join(GOLD, fixture))
    w = synthetic_weights(g)
    params = {vid: np.asarray(w[name], np.float32) for vid, (name, _) in g.params.items()}
    return g, w, nhwc.compile_plan(g, params, shape)


@pytest.mark.parametrize("fixture,nb,H,W", [("ppocr_det_graph.json", 2, 64, 96), ("ppocr_det_graph.json", 1, 96, 64), ("ppocr_det_graph.json", 3, 32, 160),
                                            ("ppocr_det_fast_graph.json", 2, 64, 96), ("ppocr_det_fast_graph.json", 1, 96, 160)])
def test_plan_replay_matches_interpreter(fixture, nb, H, W):
    g, w, plan = _compiled(fixture, (nb, 3, H, W))
    x = np.random.default_rng(H + W + nb).standard_normal((nb, 3, H, W)).astype(np.float32)
    got, _ = _det_replay.run_plan(plan, x)
    ref64 = run_graph(g, w, torch.from_numpy(x), dtype=torch.float64).numpy()
    ref32 = run_graph(g, w, torch.from_numpy(x)).numpy()
    err, cpu32 = np.abs(got - ref64).max(), np.abs(ref32 - ref64).max()
    print(f"{fixture} {nb}x{H}x{W}: plan replay vs fp64 interpreter {err:.2e} (fp32 interpreter {cpu32:.2e}); steps {plan.stats}")
    assert got.shape == ref64.shape and err <= 1e-5
    # every image of a batch comes out as it does alone (the plan is per input shape; rows of different images share nothing)
    if nb > 1:
        _, _, plan1 = _compiled(fixture, (1, 3, H, W))
        one, _ = _det_replay.run_plan(plan1, x[1:2])
        assert np.abs(one[0] - got[1]).max() <= 2e-6


def test_server_program_stays_nhwc():
    """what the plan is for: of the server program's 144 convs only the 3-channel stem runs on the NCHW kernels (two layout passes and one im2col chunk -- the
    one-channel part of the head's concat -- lead into the NHWC buffers), no value goes back to
    NCHW planes, every residual add / FPN upsample rides on a GEMM, every concat is its producers' slices"""
    g, w, plan = _compiled("ppocr_det_graph.json", (2, 3, 64, 96))
    kinds = {}
    for k, _ in plan.steps:
        kinds[k] = kinds.get(k, 0) + 1
    assert kinds.get("from_view", 0) == 0 and kinds["to_view"] == 2 and kinds["im2col_view"] == 1 and kinds["conv_nchw"] == 1 and kinds["gemm"] == 114
    assert kinds["dwconv_view"] == 27 and kinds["dots_view"] == 2 and kinds["nearest_view"] == 4 and kinds.get("nearest_nchw", 0) == 0
    assert plan.stats["add as residual"] == 36 and plan.stats["nearest as residual rows"] == 3 and plan.stats["concat as slices"] == 9
    assert kinds.get("binary", 0) == 1 and kinds.get("copy", 0) == 0          # the final add of the two probability maps
    n_conv = sum(1 for k, *_ in g.ops if k in ("conv2d", "depthwise_conv2d", "conv2d_transpose"))
    assert n_conv == kinds["gemm"] + kinds["dwconv_view"] + kinds["dots_view"] + kinds["conv_nchw"]
    # residuals and outputs keep the 16-byte alignment the GEMM's float4 epilogue asks for
    for k, p in plan.steps:
        if k == "gemm":
            t = p["tables"]
            assert (plan.consts[t["rowC"]] % 4 == 0).all() and (plan.consts[t["colC"]] % 4 == 0).all() and p["K"] % 32 == 0 and p["N"] % 32 == 0
            if "rowR" in t:
                assert (plan.consts[t["rowR"]] % 4 == 0).all()
    # FLOPs: the program's, not the padded problems'
    shape, _ = nhwc.infer_shapes(g, (2, 3, 64, 96))
    fl = 0.0
    for kind, ins, outs, a in g.ops:
        if kind in ("conv2d", "depthwise_conv2d"):
            n, cin, _, _ = shape[ins[0]]
            cout, _, kh, kw = shape[ins[1]]
            fl += 2.0 * n * cout * shape[outs[0]][2] * shape[outs[0]][3] * kh * kw * (1 if kind == "depthwise_conv2d" else cin)
        elif kind == "conv2d_transpose":
            n, cin, h, w_ = shape[ins[0]]
            fl += 2.0 * n * h * w_ * 4 * shape[outs[0]][1] * cin
    assert abs(plan.flops - fl) <= 1e-9 * fl


def test_tables_are_refused_beyond_32_bits():
    g = load_graph(os.path.join(GOLD, "ppocr_det_graph.json"))
    params = {vid: np.zeros(shape, np.float32) + 0.01 for vid, (_, shape) in g.params.items()}
    with pytest.raises(ValueError):
        nhwc.compile_plan(g, params, (48, 3, 544, 960))


def test_thin_variant_rule():
    """which GEMM steps leave the persistent kernel (profiles/r06c_det_plan_variants.log): short K, narrow N with moderate K, few tiles"""
    assert nhwc.thin_variant(1044480, 32, 128, 4080) == 1            # K <= 256
    assert nhwc.thin_variant(261120, 64, 576, 2040) == 1             # N <= 96 and K <= 1152
    assert nhwc.thin_variant(16320, 512, 2176, 512) == 1             # <= 512 tiles and K <= 2400
    assert nhwc.thin_variant(261120, 64, 20736, 8160) == 3           # the neck's 9x9 convs
    assert nhwc.thin_variant(65280, 256, 704, 2040) == 3
    assert nhwc.thin_variant(4080, 1024, 3328, 256) == 3             # few tiles but a long K
