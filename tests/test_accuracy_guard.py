"""The accuracy guard of the reduced-precision modes (vsr_amd/engine.py AccuracyGuard) without a GPU: when a check is due, which mode
the verdict restores or demotes to, how demotions are counted.  The guarded engines themselves are exercised on the GPU by
tests/test_gpu_weight_sweep.py (weights on which the fp16-operand modes stay in range and still miss 50 dB)."""
import torch

import vsr_amd  # noqa: F401
from vsr_amd.engine import AccuracyGuard


class _Fake(AccuracyGuard):
    def __init__(self, mode="f16"):
        self.applied = []
        self._guard_init(mode, {"f16": "split-format", "split-format": "f32", "split": "f32"})

    def _apply_precision(self, m):
        self.applied.append(m)
        self.precision = m


def test_first_unit_is_checked_then_every_nth():
    f = _Fake()
    f._guard_every = 4
    assert [f._guard_due() for _ in range(9)] == [True, False, False, False, True, False, False, False, True]
    f.set_precision("f32")
    assert not any(f._guard_due() for _ in range(9))             # the exact mode is never checked
    f.set_precision("split")
    assert f._guard_due()                                          # a mode the caller picks starts with a check
    f._guard_every = 0
    f._guard_calls = 0
    assert not f._guard_due()                                      # VSR_F16_SELFCHECK_EVERY=0: off


def test_verdict_restores_or_demotes():
    f = _Fake()
    assert f._guard_due()
    mode = f._guard_exact()
    assert mode == "f16" and f.precision == "f32"                  # the reference run is exact
    a, b = torch.zeros(1000), torch.zeros(1000)
    b[0] = 0.5                                                     # 255 / sqrt(0.25 / 1000): 84 dB
    assert f._guard_verdict(mode, a, b, None, 255.0) and f.precision == "f16" and f.demotions == 0
    mode = f._guard_exact()
    b[:] = 30.0                                                    # 18.6 dB
    assert not f._guard_verdict(mode, a, b, None, 255.0)
    assert f.precision == "split-format" and f.demotions == 1 and f._guard_calls == 0      # demoted one step; its first unit is checked
    mode = f._guard_exact()
    assert not f._guard_verdict(mode, a, b, None, 255.0) and f.precision == "f32" and f.demotions == 2
    assert [m for m, _ in f.guard_log] == ["f16", "f16", "split-format"]


def test_verdict_over_a_region_and_non_finite():
    f = _Fake()
    a, b = torch.zeros(4, 4), torch.zeros(4, 4)
    b[0, 0] = 100.0
    where = torch.zeros(4, 4, dtype=torch.bool)
    where[1:, 1:] = True                                           # the differing pixel is outside the compared region
    assert f._guard_verdict(f._guard_exact(), a, b, where, 255.0)
    a[2, 2] = float("nan")
    assert not f._guard_verdict(f._guard_exact(), a, b, where, 255.0) and f.demotions == 1
