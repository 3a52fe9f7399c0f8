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


def test_inpaint_area_1080p_value():
    mask = t.create_mask((1080, 1920), [(288, 1632, 950, 1070)])
    assert mask[940, 278] == 255 and mask[939, 278] == 0 and mask[1079, 1642] == 255 and mask[1079, 1643] == 0
    assert t.get_inpaint_area_by_mask(1920, 1080, 360, t.threshold_mask(mask)) == [(720, 1080, 0, 1920)]


def test_ab_sections():
    assert t.is_frame_number_in_ab_sections(5, None) and t.is_frame_number_in_ab_sections(5, [])
    assert t.is_frame_number_in_ab_sections(5, [range(0, 6)]) and not t.is_frame_number_in_ab_sections(6, [range(0, 6)])
