"""N > 1 path of the detector-driven modes on CPU (gloo, world_size 2): SubtitleRemover.propainter_mode / video_inpaint deal
the plugin's batches round-robin over the ranks (tools/batch_parallel.py); the written video must equal the single-process
one and every batch must reach the plugin with the single-process boundaries and mask."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, H, W = 200, 90, 160


def _clip():
    clip = ((np.arange(N * H * W * 3, dtype=np.int64) * 7) % 200).astype(np.uint8).reshape(N, H, W, 3)
    clip[:, 0, 0, 0] = np.arange(N)
    return clip


class _Det:
    quad = np.array([[[40, 60], [120, 60], [120, 75], [40, 75]]])

    def predict(self, img):
        no = int(img[0, 0, 0])
        return [{"dt_polys": self.quad if 20 <= no < 180 else np.zeros((0, 4, 2))}]


def _run_mode(mode, rank_tag, log):
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.video_io import ArrayVideo

    def plugin(batch, mask):                   # stand-in for the GPU plugin: marks frame, batch position and mask coverage
        log.append(([int(f[0, 0, 0]) for f in batch], int(mask.sum() // 255)))
        out = []
        for j, f in enumerate(batch):
            g = f.copy()
            g[1, 1, 0] = j
            g[1, 1, 1] = len(batch)
            g[mask > 0] = 255 - g[mask > 0]
            out.append(g)
        return out

    sr = SubtitleRemover(ArrayVideo(_clip(), fps=25.0), device="cpu", model_path="unused")
    sr.sub_areas = [(0, H, 0, W)]
    if mode == "propainter":
        sr.propainter_mode(None, propainter_inpaint=plugin, text_detector=_Det(), scene_div_points=[101])
    else:
        sr.video_inpaint(None, plugin, text_detector=_Det())
    return np.stack(sr.video_writer.frames) if sr.video_writer.frames else None


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import vsr_amd  # noqa: F401

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    log = []
    out = _run_mode(mode, rank, log)
    q.put((rank, out, log))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["propainter", "sttn-det"])
def test_two_rank_batch_parallel_equals_single_process(mode):
    import vsr_amd  # noqa: F401

    ref_log = []
    ref = _run_mode(mode, 0, ref_log)
    assert ref.shape == _clip().shape and len(ref_log) >= 3
    world = 2
    port = 31500 + (os.getpid() % 2000) + (7 if mode == "propainter" else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = {m[0]: m for m in (q.get(timeout=180) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert msgs[1][1] is None, "only rank 0 owns the sink"
    assert np.array_equal(msgs[0][1], ref), "the written video must not depend on the number of ranks"
    # batches: same boundaries and masks as the single-process loop, dealt round-robin within each round of `world`
    assert sorted(msgs[0][2] + msgs[1][2]) == sorted(ref_log)
    assert msgs[0][2] == ref_log[0::2] and msgs[1][2] == ref_log[1::2]


def test_driver_orders_passthrough_and_work_items():
    from vsr_amd.backend.tools.batch_parallel import run_batch_parallel

    items = [("pass", 0), ("work", [1, 2], np.zeros((2, 2), np.uint8)), ("pass", 3), ("work", [4], np.zeros((2, 2, 1), np.uint8)), ("pass", 5)]
    out = []
    run_batch_parallel(iter(items), lambda frames, mask: [f * 10 for f in frames] if mask.ndim == 2 else None, out.append)
    assert out == [0, 10, 20, 3, 40, 5]
    out = []
    run_batch_parallel(iter([]), None, out.append)
    assert out == []


def test_prefetch_admits_a_batch_larger_than_its_budget():
    """a WORK batch with more frames than `prefetch_frames` (config.sttnMaxLoadNum goes to 300, the budget defaults to
    256) used to hang the reader thread; it now passes alone, the budget still bounds how far the reader runs ahead"""
    import threading

    from vsr_amd.backend.tools.batch_parallel import _Prefetch, run_batch_parallel

    items = [("pass", 0)] + [("work", list(range(10 * k, 10 * k + 10)), np.zeros((2, 2), np.uint8)) for k in range(1, 4)] + [("pass", 99)]
    out = []
    t = threading.Thread(target=lambda: run_batch_parallel(iter(items), lambda frames, mask: [f + 1000 for f in frames], out.append, prefetch_frames=8),
                         daemon=True)
    t.start()
    t.join(20)
    assert not t.is_alive(), "the prefetcher deadlocked on a batch larger than its budget"
    assert out == [0] + [f + 1000 for f in range(10, 40)] + [99]
    # the budget is respected when items fit: never more than 8 frames queued ahead of the consumer
    peak = [0]

    def gen():
        for i in range(50):
            yield ("pass", i)

    pf = _Prefetch(gen(), 8)
    got = []
    for v in pf:
        peak[0] = max(peak[0], pf.q.qsize())
        got.append(v[1])
    assert got == list(range(50)) and peak[0] <= 8
    pf.close()


def _bounded_worker(rank, world, port, fail, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools.batch_parallel import run_batch_parallel

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    written, seen, owners = [], [], []
    fr = lambda v: np.full((2, 3, 3), v, np.uint8)

    def items():
        yield ("work", [fr(1), fr(2)], np.zeros((2, 3), np.uint8))          # batch 0 -> rank 0
        for i in range(150):                                              # a long subtitle-free stretch
            if i == 149:
                seen.append(len(written))                                 # how much was flushed before the stretch ended
            yield ("pass", fr(10))
        if fail:
            raise ValueError("reader died")
        yield ("work", [fr(3)], np.zeros((2, 3), np.uint8))                # batch 1 -> rank 1 (ownership survives the partial round)
        yield ("pass", fr(11))

    def process(frames, mask):
        owners.append(rank)
        return [f + 100 for f in frames]

    err = None
    try:
        run_batch_parallel(items() if rank == 0 else (), process, lambda f: written.append(int(f[0, 0, 0])), dist=dist,
                           max_pending=64, prefetch_frames=0)
    except ValueError as e:
        err = str(e)
    q.put((rank, written, seen, owners, err))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [False, True])
def test_partial_round_is_flushed_and_peers_are_released(fail):
    """(a) a batch followed by a long pass-through stretch must not buffer the stretch; (b) an exception on rank 0
    must not leave the peers blocked in recv."""
    world = 2
    port = 33500 + (os.getpid() % 2000) + (3 if fail else 0)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bounded_worker, args=(r, world, port, fail, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = {m[0]: m for m in (q.get(timeout=120) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0, "no rank may hang or die"
    _, written, seen, owners0, err = msgs[0]
    assert seen and seen[0] >= 2 + 64, "the queued batch and the first 64 pass-through frames are out before the stretch ends"
    if fail:
        assert err == "reader died" and msgs[1][3] == []
        assert written[:2] == [101, 102] and written[2:] == [10] * (len(written) - 2)
    else:
        assert err is None
        assert written == [101, 102] + [10] * 150 + [103, 11]
        assert owners0 == [0] and msgs[1][3] == [1]
