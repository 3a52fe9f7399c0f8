"""Host engine logic without a GPU: replay the plan (tables, descriptors, schedule, packed
weights) on the CPU and compare with the oracle's STTNInpaint.inpaint()."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

from vsr_amd.synth import make_state_dict

from _replay import PlanView


@pytest.fixture(scope="module")
def host_engine(built_lib):
    from vsr_amd.engine import SttnEngine

    sd = make_state_dict(0, "auto")
    eng = SttnEngine(sd, "auto", device=None, neighbor_stride=2, ref_length=3)
    yield sd, eng
    eng.close()


def test_plan_replay_matches_oracle(built_lib):
    """Default tuning: 4 frames, stride 2 / refs every 3 -> windows T = 4, 4; counts [2,2,2,1]."""
    import _replay_check

    res = _replay_check.run()
    assert res["counts"] == [2, 2, 2, 1]


def test_plan_replay_degenerate_chunks(built_lib):
    """a 1-frame and a 2-frame chunk with the reference's default stride 5 / references every 10 (the ragged tail of a video,
    sttn_auto_inpaint.py:240-245): one window of T = 1 / T = 2, every frame visited once and returned as uint8"""
    import _replay_check

    assert _replay_check.run(L=1, ns=5, rl=10, seed=3)["counts"] == [1]
    assert _replay_check.run(L=2, ns=5, rl=10, seed=4)["counts"] == [1, 1]


def test_plan_replay_sttn_det(built_lib):
    """sttn-det geometry (432x240, patch table of network_sttn.py:69), pre-masked input, model-res blend."""
    import _replay_check

    res = _replay_check.run_det()
    assert res["counts"] == [2, 2]


def test_plan_flops_det_matches_survey(built_lib):
    """733.8 GFLOP per frame for a 50-frame sttn-det batch."""
    from vsr_amd.engine import SttnEngine
    from vsr_amd.synth import make_state_dict

    eng = SttnEngine(make_state_dict(1, "det"), "det", device=None)
    try:
        f50 = eng.flops(50)
    finally:
        eng.close()
    assert abs(f50 / 50 / 1e9 - 733.8) < 1.5, f50 / 50 / 1e9


def test_plan_replay_split_pv_and_square_tiles():
    """Non-default tuning in a fresh process: PV split-K + reduce-scatter, 128x128 conv tiles,
    tap-major K order (the library reads its tuning env once per process)."""
    env = dict(os.environ, VSR_PV_SPLIT_CHUNKS="10", VSR_CONV_TILE="0", VSR_PV_TILE="0", VSR_CONV_KORDER="0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_replay_check.py")], env=env, capture_output=True,
                       text=True, timeout=2400)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads(r.stdout.strip().splitlines()[-1])
    assert res["pv_split"] >= 2 and res["has_reduce"], res
    assert res["counts"] == [2, 2, 2, 1]


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
