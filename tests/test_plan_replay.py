"""Host engine logic without a GPU: replay the plan (tables, descriptors, schedule, packed
weights) on the CPU and compare with the oracle's STTNInpaint.inpaint()."""
import numpy as np
import pytest

from oracle.sttn_auto import STTNInpaintOracle, calculate_psnr
from vsr_amd.synth import make_state_dict

from _replay import PlanView, replay


@pytest.fixture(scope="module")
def host_engine(built_lib):
    from vsr_amd.engine import SttnEngine

    sd = make_state_dict(0, "auto")
    eng = SttnEngine(sd, "auto", device=None, neighbor_stride=2, ref_length=3)
    yield sd, eng
    eng.close()


def test_plan_replay_matches_oracle(built_lib, host_engine):
    sd, eng = host_engine
    L = 6        # windows (stride 2, refs every 3): T = 4, 5, 5; counts [2,2,3,2,2,1]
    rng = np.random.default_rng(11)
    frames = rng.integers(0, 256, size=(L, 120, 640, 3), dtype=np.uint8)
    view = PlanView(built_lib, eng, L)
    try:
        comp, counts, bufs = replay(view, eng.packed_weights(), frames)
    finally:
        view.close()
    ref = STTNInpaintOracle(sd, "auto", neighbor_stride=2, ref_length=3).inpaint(list(frames))
    assert counts.tolist() == [2, 2, 3, 2, 2, 1]
    for i, r in enumerate(ref):
        assert (r.dtype == np.uint8) == (counts[i] == 1), "u8-vs-f32 path selection must follow the visit count"
    refa = np.stack([r.astype(np.float32) for r in ref])
    d = np.abs(comp - refa)
    # same fp32 arithmetic up to summation order: only truncation-boundary flips (+-1 before averaging)
    assert d.max() <= 1.0, d.max()
    assert (d > 0).mean() < 2e-3, (d > 0).mean()
    assert calculate_psnr(comp, refa) > 70.0
    assert 20.0 < comp.std() < 120.0, "synthetic weights should give a full-range image"
    # algorithmic flops = sum over the plan's GEMMs; cross-check with SURVEY.md 8(d) at L=50 elsewhere
    assert view.flops > 0


def test_plan_flops_L50_matches_survey(built_lib, host_engine):
    """32.14 TFLOP per full 50-frame chunk = 642.8 GFLOP per output frame."""
    from vsr_amd.engine import SttnEngine

    sd, _ = host_engine
    eng = SttnEngine(sd, "auto", device=None)
    try:
        f50 = eng.flops(50)
    finally:
        eng.close()
    assert abs(f50 / 50 / 1e9 - 642.8) < 1.0, f50 / 50 / 1e9


def test_plan_tables_stay_inside_buffers(built_lib, host_engine):
    """Every gathered address of every GEMM (including padded rows) lies inside its buffer."""
    sd, eng = host_engine
    view = PlanView(built_lib, eng, 7)
    try:
        for info, items in view.ops:
            if info.kind != 1:
                continue
            bm, bn = built_lib.TILE_DIMS[info.tile_cfg]
            for it in items:
                rowA, colA = view.tables[it.tRowA], view.tables[it.tColA]
                assert len(rowA) >= it.tilesM * bm and len(colA) >= it.K // 32
                lo = it.offA + rowA.min() + colA[: it.K // 32].min()
                hi = it.offA + rowA.max() + colA[: it.K // 32].max() + 31
                assert 0 <= lo and hi < view.buf_elems[it.bufA], info.tag
                rowB, colB = view.tables[it.tRowB], view.tables[it.tColB]
                if info.bmode == 0:
                    assert len(rowB) >= it.tilesN * bn and len(colB) >= it.K // 32
                    nb = it.K // 32
                else:
                    assert len(rowB) >= it.K and len(colB) >= it.tilesN * bn // 32
                    nb = it.tilesN * bn // 32
                lo = it.offB + rowB.min() + colB[:nb].min()
                hi = it.offB + rowB.max() + colB[:nb].max() + 31
                assert 0 <= lo and hi < view.buf_elems[it.bufB], info.tag
                rowC, colC = view.tables[it.tRowC], view.tables[it.tColC]
                assert len(rowC) >= it.tilesM * bm and len(colC) >= it.tilesN * bn // 32
                ncc = (it.N + 31) // 32
                hi = it.offC + (it.splitK - 1) * it.splitStride + rowC[: it.M].max() + colC[:ncc].max() + 31
                assert hi < view.buf_elems[it.bufC] + 32, info.tag
                assert it.splitK * it.chunksPerSplit >= it.K // 32
                assert (it.splitK - 1) * it.chunksPerSplit < it.K // 32, "no empty split"
    finally:
        view.close()
