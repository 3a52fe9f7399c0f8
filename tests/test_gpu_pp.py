"""ProPainter generator stages on the MI355X through the C-ABI against oracle/propainter.py."""
import numpy as np
import pytest
import torch

from oracle.make_golden import propainter_inputs
from oracle.propainter import ProPainterOracle
from vsr_amd.engine import PpEngine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engine(built_lib, gpu_device):
    e = PpEngine(device=0)
    yield e
    e.close()


@pytest.mark.parametrize("t,H,W", [(2, 32, 40), (6, 64, 96), (5, 360, 640)])
def test_image_propagation_bit_exact(engine, gpu_device, t, H, W):
    frames, masks, ff, fb = propainter_inputs(60 + t, t, t, H, W)
    masked = frames * (1 - masks)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
    got, gm = engine.img_propagation(d(masked), d(ff), d(fb), d(masks[:, 0].astype(np.uint8)))
    torch.cuda.synchronize()
    ref, rm = ProPainterOracle({}).img_propagation(torch.from_numpy(masked), torch.from_numpy(ff), torch.from_numpy(fb),
                                                   torch.from_numpy(masks).clone())
    mism = (gm.cpu().numpy() != rm[:, 0].numpy().astype(np.uint8)).mean()
    diff = (got.cpu() != ref).float().mean().item()
    print(f"imgprop {t}x{H}x{W}: mask mismatches {mism:.2e}, pixel mismatches {diff:.2e}, filled {1 - rm.sum().item() / masks.sum():.3f}")
    # thresholded decisions (flow consistency, nearest rounding) can flip on last-bit differences of the sampling coordinates
    assert mism <= 1e-4 and diff <= 1e-4
