"""Host-side mirror of the reference's bookkeeping (no GPU): CLI flags, config values, mask helpers, batching."""
import json
import os

import numpy as np
import pytest

import vsr_amd  # noqa: F401
from oracle import sttn_auto as oracle
from vsr_amd.backend.config import config
from vsr_amd.backend.tools import inpaint_tools as t
from vsr_amd.backend.tools.args_handler import parse_args
from vsr_amd.backend.tools.constant import InpaintMode

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_cli_flags_match_reference():
    a = parse_args(["-i", "v.mp4", "-o", "out.mp4", "-c", "950", "1070", "288", "1632", "-c", "0", "10", "0", "20",
                    "--inpaint-mode", "sttn-det"])
    assert a.input == "v.mp4" and a.output == "out.mp4"
    assert a.subtitle_area_coords == [[950, 1070, 288, 1632], [0, 10, 0, 20]]
    assert a.inpaint_mode is InpaintMode.STTN_DET
    assert parse_args(["-i", "x"]).inpaint_mode is InpaintMode.STTN_AUTO          # default (args_handler.py:23)
    assert parse_args(["-i", "x"]).subtitle_area_coords == []
    assert [m.value for m in InpaintMode] == ["sttn-auto", "sttn-det", "lama", "propainter", "opencv"]
    with pytest.raises(SystemExit):
        parse_args(["-i", "x", "--inpaint-mode", "nope"])


def test_config_defaults():
    assert config.sttnNeighborStride.value == 5 and config.sttnReferenceLength.value == 10
    assert config.getSttnMaxLoadNum() == 50 and config.propainterMaxLoadNum.value == 70
    assert config.subtitleAreaDeviationPixel.value == 10


def test_batch_generator_matches_reference_fixture():
    gold = json.load(open(os.path.join(GOLD, "batch_generator.json")))
    for key, sizes in gold.items():
        n, m = (int(v) for v in key.split(","))
        assert [len(b) for b in t.batch_generator(list(range(n)), m)] == sizes, key


@pytest.mark.parametrize("size,boxes", [
    ((480, 852), [(111, 766, 373, 452)]),                       # test/test.png: box y 373..452, x 111..766
    ((1080, 1920), [(288, 1632, 950, 1070)]),
    ((720, 1280), [(192, 1088, 620, 700)]),
    ((1080, 1920), [(288, 1632, 950, 1070), (100, 900, 40, 90)]),          # two separate strips
    ((1080, 1920), [(288, 800, 900, 960), (900, 1632, 980, 1040)]),        # two islands merged into one strip
    ((2160, 3840), [(576, 3264, 1900, 2140)]),
    ((480, 852), [(0, 5, 0, 5)]),                                          # clamped at 0 on the low side only
    ((480, 852), []),
])
def test_mask_helpers_match_oracle(size, boxes):
    H, W = size
    m1, m2 = t.create_mask(size, boxes), oracle.create_mask(size, boxes)
    assert np.array_equal(m1, m2)
    mask01 = t.threshold_mask(m1)
    assert mask01.shape == (H, W, 1) and set(np.unique(mask01)) <= {0, 1}
    for h, mult in ((int(W * 3 / 16), 1), (int(W * 5 / 18), 1), (int(W * 3 / 16), 8)):
        a1 = t.get_inpaint_area_by_mask(W, H, h, mask01, mult)
        a2 = oracle.get_inpaint_area_by_mask(W, H, h, mask01, mult)
        assert a1 == a2
        for ymin, ymax, xmin, xmax in a1:
            assert 0 <= ymin < ymax <= H
            if mult == 1:
                assert (xmin, xmax) == (0, W) and ymax - ymin == min(h, H)
            else:       # symmetric shrink to a multiple (tools/inpaint_tools.py:216-237)
                assert (ymax - ymin) % mult == 0 and (xmax - xmin) % mult == 0 and xmin == (W % mult) // 2


def test_mask_islands_without_scipy_equal_scipy_label():
    """inpaint_tools._islands (round 4: union-find over the runs of set pixels per row, no scipy import in front of the first chunk)
    gives the statistics and the label order of scipy.ndimage.label with 8-connectivity -- rectangles, specks touching by corners,
    diagonals, noise -- and hands masks with a great many runs to scipy."""
    rng = np.random.default_rng(0)
    for trial in range(120):
        H, W = int(rng.integers(20, 160)), int(rng.integers(20, 240))
        m = np.zeros((H, W), np.uint8)
        if trial % 3 == 0:
            for _ in range(int(rng.integers(1, 6))):
                y0, x0 = int(rng.integers(0, H - 2)), int(rng.integers(0, W - 2))
                m[y0:y0 + int(rng.integers(1, 40)), x0:x0 + int(rng.integers(1, 80))] = 255
        elif trial % 3 == 1:
            m = (rng.random((H, W)) < rng.uniform(0.02, 0.6)).astype(np.uint8) * 255
        else:
            for _ in range(int(rng.integers(1, 12))):
                y0, x0 = int(rng.integers(0, H - 1)), int(rng.integers(0, W - 1))
                m[y0:y0 + int(rng.integers(1, 6)), x0:x0 + int(rng.integers(1, 6))] = 1
            for i in range(min(H, W) // 2):
                m[i, i] = 1
        assert t._islands(m) == t._islands_scipy(m > 0)
    assert t._islands(np.zeros((8, 8), np.uint8)) == []
    noisy = (np.indices((400, 400)).sum(0) % 2).astype(np.uint8)          # 80 000 one-pixel runs: the scipy path
    assert t._islands(noisy) == t._islands_scipy(noisy > 0)


def test_mask_rows_promise_from_the_host_mask(built_lib, monkeypatch):
    """SttnEngine.mask_rows on the caller's host copy of the mask (what the plugins pass): per area the strip rows [lo, hi) that hold
    the mask's set pixels -- the promise vsr_sttn_auto_chunk_rows / vsr_sttn_det_batch_rows take -- and the model rows they turn
    into (whole groups of four, inside the image)."""
    import ctypes as C

    from vsr_amd._lib import lib
    from vsr_amd.engine import SttnEngine
    from vsr_amd.synth import make_state_dict

    H, W = 1080, 1920
    mask = t.threshold_mask(t.create_mask((H, W), [(288, 1632, 950, 1070), (300, 1600, 500, 560)]))
    areas = t.get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask)
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=None)
    try:
        rows = eng.mask_rows(mask, areas)
        assert rows.shape == (len(areas), 2) and len(areas) == 2
        for (ymin, ymax, _, _), (lo, hi) in zip(areas, rows):
            strip = mask[ymin:ymax, :, 0]
            assert 0 <= lo < hi <= ymax - ymin
            assert strip[lo].any() and strip[hi - 1].any() and not strip[:lo].any() and not strip[hi:].any()
            a, b = C.c_int32(), C.c_int32()
            assert lib.vsr_sttn_decode_rows(eng._h, int(ymax - ymin), int(lo), int(hi), C.byref(a), C.byref(b)) == 0
            assert 0 <= a.value < b.value <= 120 and a.value % 2 == 0 and (b.value % 4 == 0 or b.value == 120)
            # the model rows the first / last mask row is resized from lie inside the range
            assert a.value <= (lo + 0.5) * 120 / (ymax - ymin) - 0.5 + 1 and b.value >= (hi - 0.5) * 120 / (ymax - ymin) - 0.5
            assert lib.vsr_sttn_flops_rows(eng._h, 50, a.value, b.value) < eng.flops(50) < eng.flops(50, reference=True)
        assert (eng.mask_rows(np.zeros((H, W), np.uint8), areas) == 0).all()          # no set pixel: no promise
        cols = eng.mask_cols(mask, areas)              # the column half (vsr_sttn_auto_chunk_box, opt-in)
        for (ymin, ymax, _, _), (lo, hi) in zip(areas, cols):
            strip = mask[ymin:ymax, :, 0]
            assert 0 <= lo < hi <= W and strip[:, lo].any() and strip[:, hi - 1].any() and not strip[:, :lo].any() and not strip[:, hi:].any()
        assert (eng.mask_cols(np.zeros((H, W), np.uint8), areas) == 0).all()
        import torch                                   # the tensor route (what a caller without a host copy of the mask takes)
        mt = torch.from_numpy(np.ascontiguousarray(mask[:, :, 0]))
        assert np.array_equal(eng.mask_cols(mt, areas), cols) and np.array_equal(eng.mask_rows(mt, areas), rows)
        mt[0, 0] = 1                                   # an in-place change is seen (the rows are cached per tensor version)
        assert eng.mask_rows(mt, [(0, 100, 0, W)])[0].tolist() == [0, 1]
        for (lo, hi) in cols:                          # ... and the model columns they turn into: whole groups of eight around the taps
            a, b = C.c_int32(), C.c_int32()
            assert lib.vsr_sttn_decode_cols(eng._h, W, int(lo), int(hi), C.byref(a), C.byref(b)) == 0
            assert 0 <= a.value < b.value <= 640 and a.value % 8 == 0 and (b.value % 8 == 0 or b.value == 640)
            assert a.value <= (lo + 0.5) * 640 / W - 0.5 and b.value >= (hi - 0.5) * 640 / W - 0.5 + 1 or b.value == 640
            assert lib.vsr_sttn_flops_box(eng._h, 50, 76, 120, a.value, b.value) < lib.vsr_sttn_flops_rows(eng._h, 50, 76, 120)
        assert lib.vsr_sttn_decode_cols(eng._h, W, 5, 5, C.byref(a), C.byref(b)) != 0             # an empty promise is an error
        # chunk_flops prices what the engine runs: the host side asks the LIBRARY how it read the switch (once per process), so an
        # environment change in mid-process cannot make the two disagree; with the columns on (default since round 5) a
        # chunk costs less than the rows promise alone
        from vsr_amd import switches

        monkeypatch.setenv("VSR_DECODE_COLS", "0" if switches.on("VSR_DECODE_COLS") else "1")
        assert switches.on("VSR_DECODE_COLS") == (lib.vsr_switch_state(b"VSR_DECODE_COLS") == 1)
        monkeypatch.delenv("VSR_DECODE_COLS")
        rows_only = 0.0
        for (ymin, ymax, _, _), (lo, hi) in zip(areas, eng.mask_rows(mask[:, :, 0], areas)):
            a, b = C.c_int32(), C.c_int32()
            assert lib.vsr_sttn_decode_rows(eng._h, int(ymax - ymin), int(lo), int(hi), C.byref(a), C.byref(b)) == 0
            rows_only += lib.vsr_sttn_flops_rows(eng._h, 50, a.value, b.value)
        priced = eng.chunk_flops(50, mask[:, :, 0], areas)
        assert priced < rows_only if switches.on("VSR_DECODE_COLS") else priced == rows_only
    finally:
        eng.close()


def test_inpaint_area_1080p_value():
    mask = t.create_mask((1080, 1920), [(288, 1632, 950, 1070)])
    assert mask[940, 278] == 255 and mask[939, 278] == 0 and mask[1079, 1642] == 255 and mask[1079, 1643] == 0
    assert t.get_inpaint_area_by_mask(1920, 1080, 360, t.threshold_mask(mask)) == [(720, 1080, 0, 1920)]


def test_ab_sections():
    assert t.is_frame_number_in_ab_sections(5, None) and t.is_frame_number_in_ab_sections(5, [])
    assert t.is_frame_number_in_ab_sections(5, [range(0, 6)]) and not t.is_frame_number_in_ab_sections(6, [range(0, 6)])


def _tup(x):
    return [tuple(v) for v in x]


def test_bookkeeping_matches_reference_fixture():
    """Interval / region bookkeeping of the detector modes against outputs of the reference's own functions
    (tests/golden/bookkeeping.json, generated by oracle/make_golden.py)."""
    from vsr_amd.backend.tools.ocr import get_coordinates
    from vsr_amd.backend.tools.subtitle_detect import SubtitleDetect

    g = json.load(open(os.path.join(GOLD, "bookkeeping.json")))
    for c in g["filter_and_merge"]:
        assert SubtitleDetect.filter_and_merge_intervals(_tup(c["in"]), c["target"]) == _tup(c["out"])
    for c in g["expand"]:
        assert t.expand_frame_ranges(_tup(c["in"]), c["b"], c["f"]) == _tup(c["out"])
    for c in g["coords"]:
        assert get_coordinates(c["in"]) == _tup(c["out"])
    det = SubtitleDetect(None, [])
    for c in g["unify"]:
        got = det.unify_regions({int(k): _tup(v) for k, v in c["in"].items()})
        assert {str(k): v for k, v in got.items()} == {k: _tup(v) for k, v in c["out"].items()}
    for c in g["ranges"]:
        assert SubtitleDetect.find_continuous_ranges({k: 1 for k in c["in"]}) == _tup(c["out"])
    for c in g["ranges_same_mask"]:
        assert SubtitleDetect.find_continuous_ranges_with_same_mask({int(k): _tup(v) for k, v in c["in"].items()}) == _tup(c["out"])


def test_video_inpaint_driver_with_stub_detector_and_model():
    """SubtitleRemover.video_inpaint (main.py:260-333): which frames reach the plugin, in which batches, with which mask."""
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.video_io import ArrayVideo

    n, H, W = 130, 90, 160
    clip = np.zeros((n, H, W, 3), np.uint8)
    clip[:, 0, 0, 0] = np.arange(n)
    quad = np.array([[[40, 60], [120, 60], [120, 75], [40, 75]]])

    class Det:
        calls = 0

        def predict(self, img):
            Det.calls += 1
            no = int(img[0, 0, 0])                       # 0-based frame index
            return [{"dt_polys": quad if 20 <= no < 95 else np.zeros((0, 4, 2))}]

    seen = []

    def model(batch, mask):
        seen.append(([int(f[0, 0, 0]) for f in batch], mask.copy()))
        return [f + 1 for f in batch]

    sr = SubtitleRemover(ArrayVideo(clip, fps=25.0), device="cpu", model_path="unused")
    sr.sub_areas = [(0, H, 0, W)]
    sr.video_inpaint(None, model, text_detector=Det())
    out = np.stack(sr.video_writer.frames)
    assert out.shape == clip.shape
    assert Det.calls == (n + 1) // 2                     # fps < 30 -> SAMPLE_STEP 2
    touched = sorted(i for b, _ in seen for i in b)
    # detections on 0-based frames 20..94 sampled every 2nd -> 1-based 21..95, expanded by 3 each side
    assert touched == list(range(17, 98))
    assert [len(b) for b, _ in seen] == [len(x) for x in t.batch_generator(list(range(81)), 50)]
    for _, m in seen:
        assert m[50:86, 30:131].all() and m.sum() == 255 * 36 * 101       # box grown by 10 px
    assert (out[touched, 0, 0, 0] == clip[touched, 0, 0, 0] + 1).all()
    rest = sorted(set(range(n)) - set(touched))
    assert np.array_equal(out[rest], clip[rest])


def test_propainter_mode_driver_with_stub_detector_and_plugin():
    """SubtitleRemover.propainter_mode (main.py:159-245): intervals with one mask, cut at the scene changes, reach the plugin
    whole in batch_generator batches; frames without text pass through; split_range_by_scene follows subtitle_detect.py:135-155."""
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.video_io import ArrayVideo

    from vsr_amd.backend.tools.subtitle_detect import SubtitleDetect

    assert SubtitleDetect.split_range_by_scene([(5, 20), (30, 40)], [1, 12, 30, 35, 41]) == [(5, 11), (12, 20), (30, 34), (35, 40)]
    n, H, W = 200, 90, 160
    clip = np.zeros((n, H, W, 3), np.uint8)
    clip[:, 0, 0, 0] = np.arange(n)
    quad = np.array([[[40, 60], [120, 60], [120, 75], [40, 75]]])

    class Det:
        def predict(self, img):
            no = int(img[0, 0, 0])
            return [{"dt_polys": quad if 20 <= no < 180 else np.zeros((0, 4, 2))}]

    seen = []

    def plugin(batch, mask):
        seen.append(([int(f[0, 0, 0]) for f in batch], mask.copy()))
        return [f + 1 for f in batch]

    sr = SubtitleRemover(ArrayVideo(clip, fps=25.0), device="cpu", model_path="unused")
    sr.sub_areas = [(0, H, 0, W)]
    sr.propainter_mode(None, propainter_inpaint=plugin, text_detector=Det(), scene_div_points=[101])
    out = np.stack(sr.video_writer.frames)
    assert out.shape == clip.shape
    touched = sorted(i for b, _ in seen for i in b)
    assert touched == list(range(20, 179))                                  # detector samples every 2nd frame: 1-based 21..179
    # interval (21, 179) cut at scene change 101 -> (21, 100) and (101, 179), each batched by propainterMaxLoadNum = 70
    expect = [len(x) for x in t.batch_generator(list(range(80)), 70)] + [len(x) for x in t.batch_generator(list(range(79)), 70)]
    assert [len(b) for b, _ in seen] == expect
    assert (out[touched, 0, 0, 0] == clip[touched, 0, 0, 0] + 1).all()
    rest = sorted(set(range(n)) - set(touched))
    assert np.array_equal(out[rest], clip[rest])


def test_propainter_encoder_cache_plan():
    """encoder_cache_plan (the host side of vsr_pp_encode / vsr_pp_forward_cached): every frame gets one feature entry, every frame that is
    a reference frame of some window one token entry, the entries follow the order of the encode calls (tokens for the leading frames
    of a call), and a batch's windows ask for 3.2 times as many frame encodings as there are frames."""
    from vsr_amd.backend.inpaint.propainter_inpaint import encoder_cache_plan, get_ref_index

    for n, ref_num in ((70, -1), (23, -1), (7, -1), (100, 8)):
        windows = []
        for f in range(0, n, 5):
            nb = list(range(max(0, f - 5), min(n, f + 6)))
            windows.append((nb, get_ref_index(f, nb, n, 10, ref_num)))
        calls, feat_slot, tok_slot = encoder_cache_plan(windows, chunk=16)
        assert sorted(feat_slot) == list(range(n)) and sorted(feat_slot.values()) == list(range(n))
        refs = sorted({i for _, r in windows for i in r})
        assert sorted(tok_slot) == refs and sorted(tok_slot.values()) == list(range(len(refs)))
        fpos = tpos = 0
        for ids, ntok in calls:
            assert 1 <= len(ids) <= 16 and 0 <= ntok <= len(ids)
            assert [feat_slot[i] for i in ids] == list(range(fpos, fpos + len(ids)))
            assert [tok_slot[i] for i in ids[:ntok]] == list(range(tpos, tpos + ntok)) and not any(i in tok_slot for i in ids[ntok:])
            fpos, tpos = fpos + len(ids), tpos + ntok
        if n == 70:
            assert sum(len(nb) + len(r) for nb, r in windows) == 226 and refs == [0, 10, 20, 30, 40, 50, 60]


def test_decode_rows_and_cols_cover_what_the_resize_back_reads(built_lib):
    """vsr_sttn_decode_rows / _cols against the oracle's cv2.resize: a model-resolution image that is changed OUTSIDE the rows and columns
    the engine would decode gives the same strip pixels under the mask after the resize back (sttn_auto_inpaint.py:60-67) -- for the
    strip heights and widths of 480p ... 4K frames and masks at the edges, in the middle and one pixel wide.  (sttn-det goes the
    other way -- the mask is resized DOWN and the blend happens at model resolution: the rows / columns returned contain every
    non-zero pixel of the resized mask.)"""
    import ctypes as C

    from oracle import cv2_restate as cv2r
    from vsr_amd._lib import lib
    from vsr_amd.engine import SttnEngine
    from vsr_amd.synth import make_state_dict

    rng = np.random.default_rng(5)
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=None)
    det = SttnEngine(make_state_dict(0, "det"), "det", device=None)
    try:
        for W, sh in ((852, 159), (1280, 240), (1920, 360), (3840, 720), (1000, 187)):
            for _ in range(4):
                r0 = int(rng.integers(0, sh - 1)); r1 = int(rng.integers(r0 + 1, min(sh, r0 + 1 + sh // 2) + 1))
                c0 = int(rng.integers(0, W - 1)); c1 = int(rng.integers(c0 + 1, min(W, c0 + 1 + W // 2) + 1))
                if _ == 0:
                    r0, r1, c0, c1 = 0, 1, W - 1, W
                a, b, ca, cb = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
                assert lib.vsr_sttn_decode_rows(eng._h, sh, r0, r1, C.byref(a), C.byref(b)) == 0
                assert lib.vsr_sttn_decode_cols(eng._h, W, c0, c1, C.byref(ca), C.byref(cb)) == 0
                x = rng.integers(0, 256, size=(120, 640, 3), dtype=np.uint8)
                y = rng.integers(0, 256, size=(120, 640, 3), dtype=np.uint8)
                y[a.value:b.value, ca.value:cb.value] = x[a.value:b.value, ca.value:cb.value]
                assert np.array_equal(cv2r.resize_linear(x, (W, sh))[r0:r1, c0:c1], cv2r.resize_linear(y, (W, sh))[r0:r1, c0:c1]), (W, sh, r0, r1, c0, c1)
                # sttn-det: mask resized down to 432 x 240
                assert lib.vsr_sttn_decode_rows(det._h, sh, r0, r1, C.byref(a), C.byref(b)) == 0
                assert lib.vsr_sttn_decode_cols(det._h, W, c0, c1, C.byref(ca), C.byref(cb)) == 0
                m = np.zeros((sh, W, 1), np.uint8)
                m[r0:r1, c0:c1] = 255
                small = cv2r.resize_linear(m, (432, 240))[:, :, 0]
                ys, xs = np.flatnonzero(small.any(axis=1)), np.flatnonzero(small.any(axis=0))
                if ys.size:
                    assert a.value <= ys[0] and ys[-1] < b.value and ca.value <= xs[0] and xs[-1] < cb.value, (W, sh, r0, r1, c0, c1)
    finally:
        eng.close()
        det.close()


def test_propainter_raft_runs():
    """PropainterInpaint cuts a batch's RAFT pass into runs of consecutive pairs (memory; two RAFT lanes): every pair exactly once, in
    order, neighbouring runs sharing one frame; the sizes BASELINE config 4 produces"""
    from vsr_amd.backend.inpaint.propainter_inpaint import raft_runs

    for n in (2, 3, 18, 44, 68, 70, 71):
        for max_pairs, lanes in ((35, 1), (35, 2), (24, 1), (12, 2), (64, 1)):
            spans = raft_runs(n, max_pairs, lanes)
            pairs = [(a + i, a + i + 1) for a, b in spans for i in range(b - a - 1)]
            assert pairs == [(i, i + 1) for i in range(n - 1)]
            assert all(b - a - 1 <= max(1, max_pairs // lanes) for a, b in spans)
            assert all(spans[j + 1][0] == spans[j][1] - 1 for j in range(len(spans) - 1))
    assert [b - a - 1 for a, b in raft_runs(68, 35, 2)] == [17, 17, 17, 16]
    assert [b - a - 1 for a, b in raft_runs(68, 35, 1)] == [34, 33]
    assert [b - a - 1 for a, b in raft_runs(44, 35, 2)] == [15, 15, 13]
    assert raft_runs(20, 35, 1) == [(0, 20)] and raft_runs(1, 35, 2) == []
