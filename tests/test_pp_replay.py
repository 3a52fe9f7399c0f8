"""ProPainter generator without a GPU: plans replayed on the CPU (tests/_replay_pp.py) against
oracle/propainter.py, which tests/test_oracle_golden.py pins to the reference module."""
import numpy as np
import pytest
import torch

import _replay_pp as rp
from oracle.make_golden import propainter_inputs
from oracle.propainter import ProPainterOracle
from vsr_amd import _lib
from vsr_amd.synth import make_propainter_state_dict


@pytest.mark.parametrize("t,H,W", [(2, 32, 40), (6, 64, 96)])
def test_image_propagation_replay_matches_oracle(built_lib, t, H, W):
    frames, masks, ff, fb = propainter_inputs(50 + t, t, t, H, W)
    masked = frames * (1 - masks)
    view = rp.imgprop_plan_view(_lib, t, H, W)
    got, gm = rp.replay_imgprop(view, masked, ff, fb, masks[:, 0].astype(np.uint8))
    o = ProPainterOracle({})
    ref, rm = o.img_propagation(torch.from_numpy(masked), torch.from_numpy(ff), torch.from_numpy(fb), torch.from_numpy(masks).clone())
    assert np.array_equal(gm, rm[:, 0].numpy().astype(np.uint8))
    assert np.array_equal(got, ref.numpy()), "nearest-neighbour propagation copies pixels: bit-exact"
    if t > 1:
        assert gm.sum() < masks.sum(), "some hole pixels must get filled from neighbouring frames"
    view.close()
