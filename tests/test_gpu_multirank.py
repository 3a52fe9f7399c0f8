"""The N > 1 plugin path on the GPU: two ranks (sharing the one GPU of the test box; transport gloo, bounced through host
memory -- RCCL does not build a communicator for two ranks on one device) run STTNAutoInpaint.__call__ chunk-parallel; the
video rank 0 writes must be the single-process video bit for bit (same chunk boundaries, same kernels, same bytes)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, W, N, GAP = 480, 852, 20, 6
BOX = (400, 450, 100, 760)
AB = [(2, 9), (13, 19)]


def _run(dist_on, ab):
    import vsr_amd  # noqa: F401
    from vsr_amd import synth
    from vsr_amd.backend.config import config
    from vsr_amd.backend.inpaint.sttn_auto_inpaint import STTNAutoInpaint
    from vsr_amd.backend.tools.inpaint_tools import create_mask
    from vsr_amd.backend.tools.video_io import ArrayVideo, ArrayWriter

    old = config.sttnMaxLoadNum.value, config.sttnNeighborStride.value, config.sttnReferenceLength.value
    config.sttnMaxLoadNum.value, config.sttnNeighborStride.value, config.sttnReferenceLength.value = GAP, 1, 6
    try:
        return _run_configured(ab, synth, config, STTNAutoInpaint, create_mask, ArrayVideo, ArrayWriter)
    finally:                                   # the config object is process-global: later tests must see the defaults again
        config.sttnMaxLoadNum.value, config.sttnNeighborStride.value, config.sttnReferenceLength.value = old


def _run_configured(ab, synth, config, STTNAutoInpaint, create_mask, ArrayVideo, ArrayWriter):
    clip = synth.make_clip(N, H, W, BOX, seed=9)
    mask = create_mask((H, W), [(BOX[2], BOX[3], BOX[0], BOX[1])])

    class Host:
        gui_mode = False
        ab_sections = [range(a, e) for a, e in AB] if ab else None
        video_writer = ArrayWriter()
        ticks = 0

        def update_progress(self, tbar, increment):
            Host.ticks += increment

    plug = STTNAutoInpaint("cuda:0", {"netG": synth.make_state_dict(0, "auto")}, ArrayVideo(clip.copy()))
    assert plug.clip_gap == GAP
    plug(input_mask=mask, input_sub_remover=Host(), tbar=object())
    plug.sttn_inpaint.engine.close()
    return (np.stack(Host.video_writer.frames) if Host.video_writer.frames else None), Host.ticks, clip, mask


def _worker(rank, world, port, ab, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out, ticks, _, _ = _run(True, ab)
    q.put((rank, out, ticks))
    dist.destroy_process_group()


@pytest.mark.parametrize("ab", [False, True])
def test_two_ranks_write_the_single_process_video(built_lib, gpu_device, ab):
    ref, ticks, clip, mask = _run(False, ab)
    assert ref.shape == clip.shape and ticks == N
    world = 2
    port = 35500 + (os.getpid() % 2000) + int(ab)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ab, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = {m[0]: m for m in (q.get(timeout=300) for _ in range(world))}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert msgs[1][1] is None and msgs[1][2] == 0, "only rank 0 owns the sink"
    assert msgs[0][2] == N
    assert np.array_equal(msgs[0][1], ref), "the written video must not depend on the number of ranks"
    m = mask > 127
    assert (ref[:, m] != clip[:, m]).mean() > (0.2 if ab else 0.5)


# ---- file to file: every rank reads and writes its own chunks by offset (tools/rank_io.py) ---------------------------------------
def _run_file(src, out, ab):
    import vsr_amd  # noqa: F401
    from vsr_amd import synth
    from vsr_amd.backend.config import config
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.constant import InpaintMode

    keys = {"sttnMaxLoadNum": GAP, "sttnNeighborStride": 1, "sttnReferenceLength": 6}
    old = {k: getattr(config, k).value for k in keys}
    old_mode = config.inpaintMode.value
    try:
        for k, v in keys.items():
            getattr(config, k).value = v
        config.inpaintMode.value = InpaintMode.STTN_AUTO
        sr = SubtitleRemover(src, model_path={"netG": synth.make_state_dict(0, "auto")})
        sr.sub_areas = [BOX]
        if ab:
            sr.ab_sections = [range(a, e) for a, e in AB]
        sr.video_out_path = out
        ticks = []
        sr.update_progress = lambda tbar, increment: ticks.append(increment)
        sr.sttn_auto_mode(tbar=object())
        if sr._video_writer is not None:
            sr.video_writer.release()
        return sum(ticks)
    finally:
        for k, v in old.items():
            getattr(config, k).value = v
        config.inpaintMode.value = old_mode


def _file_worker(rank, world, port, src, out, ab, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ticks = _run_file(src, out, ab)
    q.put((rank, ticks))
    dist.destroy_process_group()


@pytest.mark.parametrize("ab", [False, True])
def test_two_ranks_write_the_single_process_file_by_offset(built_lib, gpu_device, tmp_path, monkeypatch, ab):
    """*.y4m -> *.y4m with two ranks: each preads the records of its own chunks, inpaints them and pwrites the result records at their
    offsets of the pre-sized sink -- no frame crosses a rank boundary; the file is the single-process file byte for byte"""
    from vsr_amd import synth
    from vsr_amd.backend.tools import video_io

    clip = synth.make_clip(N, H, W, BOX, seed=9)
    src = str(tmp_path / "in.y4m")
    monkeypatch.setenv("VSR_IO_COLOR", "host")
    w = video_io.Y4mWriter(src, 25.0, (W, H), chroma="420")
    for f in clip:
        w.write(f)
    w.release()
    monkeypatch.setenv("VSR_IO_COLOR", "device")
    ref = str(tmp_path / "ref.y4m")
    assert _run_file(src, ref, ab) == N
    world = 2
    out = str(tmp_path / "out.y4m")
    port = 36500 + (os.getpid() % 2000) + int(ab)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_file_worker, args=(r, world, port, src, out, ab, q)) for r in range(world)]
    for p in procs:
        p.start()
    msgs = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert msgs[0] == N and msgs[1] == 0
    assert open(out, "rb").read() == open(ref, "rb").read()
