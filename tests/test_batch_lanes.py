"""tools/batch_lanes.py: the opt-in lanes over the independent batches of a resident run (host logic; the GPU side -- one stream per
lane -- is exercised by tests/test_gpu_io.py when VSR_BATCH_LANES is set)."""
import threading
import time

import numpy as np
import pytest

import vsr_amd  # noqa: F401
from vsr_amd.backend.tools import batch_lanes


class Plugin:
    """rewrites its batch in place (like the resident plugins) and remembers which thread did which batch"""

    clones = 0

    def __init__(self, fail_on=None):
        self.fail_on, self.seen = fail_on, []

    def clone(self):
        Plugin.clones += 1
        return Plugin(self.fail_on)

    def __call__(self, frames, mask):
        if self.fail_on is not None and int(frames[0, 0]) == self.fail_on:
            raise RuntimeError(f"batch {self.fail_on} failed")
        time.sleep(0.01)
        self.seen.append((int(frames[0, 0]), threading.current_thread().name))
        frames += mask


def jobs_over(arr, n):
    step = arr.shape[0] // n
    return [(arr[k * step:(k + 1) * step], 100) for k in range(n)]


def test_one_lane_is_the_plain_loop_in_the_callers_thread(monkeypatch):
    monkeypatch.delenv("VSR_BATCH_LANES", raising=False)
    assert batch_lanes.lanes_from_env() == 1
    arr = np.repeat(np.arange(8, dtype=np.int64)[:, None], 3, axis=1)
    p = Plugin()
    plugins = batch_lanes.lane_plugins(p, 1, {})
    assert plugins == [p]
    batch_lanes.run_jobs(jobs_over(arr, 8), plugins)
    assert [b for b, _ in p.seen] == list(range(8)) and {t for _, t in p.seen} == {threading.current_thread().name}
    assert (arr[:, 0] == np.arange(8) + 100).all()


@pytest.mark.parametrize("lanes", [2, 3])
def test_lanes_run_every_batch_once(monkeypatch, lanes):
    monkeypatch.setenv("VSR_BATCH_LANES", str(lanes))
    assert batch_lanes.lanes_from_env() == lanes
    Plugin.clones = 0
    cache = {}
    p = Plugin()
    plugins = batch_lanes.lane_plugins(p, lanes, cache)
    assert len(plugins) == lanes and plugins[0] is p and Plugin.clones == lanes - 1
    assert batch_lanes.lane_plugins(p, lanes, cache)[1] is plugins[1] and Plugin.clones == lanes - 1      # built once per run
    arr = np.repeat(np.arange(24, dtype=np.int64)[:, None], 3, axis=1)
    batch_lanes.run_jobs(jobs_over(arr, 12), plugins)
    done = sorted(b for q in plugins for b, _ in q.seen)
    assert done == list(range(0, 24, 2))                       # every batch exactly once ...
    assert (arr == np.repeat(np.arange(24)[:, None], 3, axis=1) + 100).all()   # ... with the plain loop's result
    assert sum(1 for q in plugins if q.seen) >= 2              # ... and more than one lane took part
    assert all(t.startswith("vsr-batch-lane-") for q in plugins for _, t in q.seen)


def test_a_failing_batch_stops_the_queue_and_is_raised(monkeypatch):
    arr = np.repeat(np.arange(40, dtype=np.int64)[:, None], 2, axis=1)
    p = Plugin(fail_on=4)
    plugins = batch_lanes.lane_plugins(p, 2, {})
    with pytest.raises(RuntimeError, match="batch 4 failed"):
        batch_lanes.run_jobs(jobs_over(arr, 20), plugins)
    assert sum(len(q.seen) for q in plugins) < 19              # the queue was abandoned, not drained


def test_a_callable_without_clone_stays_on_one_lane():
    calls = []
    f = lambda frames, mask: calls.append(int(frames[0, 0]))   # noqa: E731
    plugins = batch_lanes.lane_plugins(f, 2, {})
    assert plugins == [f]
    arr = np.repeat(np.arange(6, dtype=np.int64)[:, None], 2, axis=1)
    batch_lanes.run_jobs(jobs_over(arr, 3), plugins)
    assert calls == [0, 2, 4]


def test_detector_lanes_give_the_same_subtitle_frames(monkeypatch):
    """SubtitleDetect's pass over a resident clip (tools/subtitle_detect.py _find_resident) with VSR_DET_LANES=2: the sampled batches
    are spread over two detector instances and come back in order -- the same {frame_no: boxes} as on one lane."""
    import torch

    from vsr_amd.backend.tools.subtitle_detect import SubtitleDetect

    N, H, W = 61, 40, 120
    frames = torch.zeros((N, H, W, 3), dtype=torch.uint8)
    on = [i for i in range(N) if 5 <= i < 23 or 31 <= i < 48]
    for i in on:
        frames[i, 28:36, 20:100] = 255

    class Clip:
        pass

    clip = Clip()
    clip.frames = frames
    clip.__class__.__len__ = lambda self: N
    quad = np.array([[20, 28], [100, 28], [100, 36], [20, 36]])

    class Det:
        batch_size = 4
        made = 0

        def __init__(self):
            Det.made += 1
            self.batches = 0

        def clone(self):
            return Det()

        def predict_batch_device(self, fr):
            self.batches += 1
            time.sleep(0.002)
            return [{"dt_polys": quad[None] if int(f[30, 50, 0]) > 200 else np.zeros((0, 4, 2), np.int32)} for f in fr]

    got = {}
    for lanes in ("1", "2"):
        monkeypatch.setenv("VSR_DET_LANES", lanes)
        Det.made = 0
        det = Det()
        sd = SubtitleDetect(None, [(0, H, 0, W)], text_detector=det)
        got[lanes] = sd._find_resident(None, clip)
        assert Det.made == int(lanes)
        if lanes == "2":
            others = sd._det_lanes[id(det)]
            assert det.batches > 0 and others[0].batches > 0 and det.batches + others[0].batches == -(-len(range(0, N, sd.SAMPLE_STEP)) // 4)
    assert got["1"] == got["2"] and len(got["1"]) > 20
