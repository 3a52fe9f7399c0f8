"""The GPU colour conversion of the *.y4m transport (csrc/io_kernels.hip through backend/tools/video_io.py) against the numpy
statement of the same BT.601 integer matrices (video_io._yuv_to_bgr / _bgr_to_yuv): bit-exact, every chroma layout the reader
accepts, odd sizes (edge replication of the 4:2:0 sub-sampler), both ranges."""
import os

import numpy as np
import pytest

from vsr_amd.backend.tools import video_io

pytestmark = pytest.mark.gpu


def _write_raw_y4m(path, planes, H, W, tag, full):
    with open(path, "wb") as f:
        f.write(f"YUV4MPEG2 W{W} H{H} F25:1 Ip A1:1 C{tag}{' XCOLORRANGE=FULL' if full else ''}\n".encode())
        for rec in planes:
            f.write(b"FRAME\n")
            f.write(rec.tobytes())


def _read_all(path):
    r = video_io.Y4mVideo(path)
    out = []
    while True:
        ok, fr = r.read()
        if not ok:
            break
        out.append(fr)
    r.release()
    return np.stack(out)


@pytest.mark.parametrize("tag,H,W,full", [("420jpeg", 1080, 1920, False), ("420mpeg2", 37, 53, False), ("420paldv", 64, 130, True),
                                          ("422", 36, 51, True), ("444", 35, 50, False), ("mono", 33, 47, False)])
def test_y4m_reader_on_the_device_equals_numpy(built_lib, gpu_device, tmp_path, monkeypatch, tag, H, W, full):
    rng = np.random.default_rng(H * W)
    cw, ch = {"420": ((W + 1) // 2, (H + 1) // 2), "422": ((W + 1) // 2, H), "444": (W, H), "mon": (0, 0)}[tag[:3]]
    planes = [rng.integers(0, 256, size=H * W + 2 * cw * ch, dtype=np.uint8) for _ in range(3)]
    planes[1][: H * W] = np.repeat(np.arange(256, dtype=np.uint8), (H * W + 255) // 256)[: H * W]     # every luma level with random chroma
    p = str(tmp_path / "v.y4m")
    _write_raw_y4m(p, planes, H, W, tag, full)
    monkeypatch.setenv("VSR_IO_COLOR", "host")
    want = _read_all(p)
    monkeypatch.setenv("VSR_IO_COLOR", "device")
    r = video_io.Y4mVideo(p)
    assert r._dc is not None, "the GPU path was not taken"
    r.release()
    got = _read_all(p)
    assert got.shape == want.shape == (3, H, W, 3) and got.dtype == np.uint8
    assert np.array_equal(got, want)
    assert got[0].flags.owndata or got[0].base is not None          # a frame of its own (the plugins patch rows into it)
    got[0][:] = 0
    assert not np.array_equal(got[0], got[1])


@pytest.mark.parametrize("chroma,H,W", [("444", 1080, 1920), ("420", 1080, 1920), ("420", 37, 53), ("444", 35, 51), ("420", 2, 2)])
def test_y4m_writer_on_the_device_equals_numpy(built_lib, gpu_device, tmp_path, monkeypatch, chroma, H, W):
    rng = np.random.default_rng(H + W)
    frames = rng.integers(0, 256, size=(3, H, W, 3), dtype=np.uint8)
    frames[0, :, :, :] = np.arange(H * W * 3, dtype=np.int64).reshape(H, W, 3) % 256
    files = {}
    for mode in ("host", "device"):
        monkeypatch.setenv("VSR_IO_COLOR", mode)
        p = str(tmp_path / f"{mode}.y4m")
        w = video_io.Y4mWriter(p, 25.0, (W, H), chroma=chroma)
        assert (w._dc is not None) == (mode == "device")
        for fr in frames:
            w.write(fr)
        w.write(frames[1].astype(np.float32) + 0.4)                  # non-u8 frames are clipped and truncated first, as the reference's writer does
        w.release()
        files[mode] = open(p, "rb").read()
    assert files["host"] == files["device"]


def test_all_bgr_triples_and_all_yuv_triples(built_lib, gpu_device, tmp_path, monkeypatch):
    """every (Y, U, V) and every (B, G, R) byte triple through both kernels, both ranges on the way in"""
    g = np.arange(256, dtype=np.uint8)
    yy, uu, vv = np.meshgrid(g, g, g, indexing="ij")
    H, W = 4096, 4096
    rec = np.concatenate([yy.reshape(-1), uu.reshape(-1), vv.reshape(-1)])
    for full in (False, True):
        p = str(tmp_path / f"all{int(full)}.y4m")
        _write_raw_y4m(p, [rec], H, W, "444", full)
        monkeypatch.setenv("VSR_IO_COLOR", "device")
        got = _read_all(p)[0]
        want = video_io._yuv_to_bgr(yy.reshape(H, W), uu.reshape(H, W), vv.reshape(H, W), full)
        assert np.array_equal(got, want)
    frame = np.stack([yy.reshape(H, W), uu.reshape(H, W), vv.reshape(H, W)], axis=-1)        # read as (B, G, R)
    dc = video_io._DeviceColor(H, W, 3 * H * W, batch=1)
    dc.bgr_buffer()[0] = frame
    got = dc.from_bgr(1, False, False)[0].copy()
    y, u, v = video_io._bgr_to_yuv(frame, False)
    assert np.array_equal(got, np.concatenate([y.reshape(-1), u.reshape(-1), v.reshape(-1)]))


@pytest.mark.parametrize("ab", [False, True])
def test_resident_chunk_loop_writes_the_same_file(built_lib, gpu_device, tmp_path, monkeypatch, ab):
    """*.y4m in -> SubtitleRemover.run() (sttn-auto) -> *.y4m out: with the decoded frames resident in HBM (planes up, GPU colour
    conversion, planes down) the written file is byte for byte the one the host-frame loop writes (numpy colour conversion on
    both sides of it), ragged last chunk and A/B sections included."""
    from vsr_amd import synth
    from vsr_amd.backend.config import config
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.constant import InpaintMode

    H, W, N, GAP = 480, 852, 20, 6
    box = (400, 450, 100, 760)
    clip = synth.make_clip(N, H, W, box, seed=11)
    src = str(tmp_path / "in.y4m")
    monkeypatch.setenv("VSR_IO_COLOR", "host")
    w = video_io.Y4mWriter(src, 25.0, (W, H), chroma="420")
    for f in clip:
        w.write(f)
    w.release()
    keys = {"sttnMaxLoadNum": GAP, "sttnNeighborStride": 1, "sttnReferenceLength": 6}
    old = {k: getattr(config, k).value for k in keys}
    old_mode = config.inpaintMode.value
    outs = {}
    try:
        for k, v in keys.items():
            getattr(config, k).value = v
        config.inpaintMode.value = InpaintMode.STTN_AUTO
        from vsr_amd.backend.tools.pinned import PinnedPool

        # (the "-late-pins" passes hold the page-locking thread back, so that the first transfers of the loop take the pageable path)
        # ("by-offset": the per-rank file access of tools/rank_io.py, forced for this single process: pread / pwrite of the records)
        for mode, color, resident in (("host", "host", "0"), ("device-frames", "device", "0"), ("resident", "device", "1"),
                                      ("device-frames-late-pins", "device", "0"), ("resident-late-pins", "device", "1"),
                                      ("by-offset", "device", "1"), ("by-offset-late-pins", "device", "1")):
            monkeypatch.setattr(PinnedPool, "test_delay", 0.25 if mode.endswith("late-pins") else 0.0)
            monkeypatch.setenv("VSR_IO_COLOR", color)
            monkeypatch.setenv("VSR_IO_RESIDENT", resident)
            monkeypatch.setenv("VSR_IO_PER_RANK", "1" if mode.startswith("by-offset") else "0")
            sr = SubtitleRemover(src, model_path={"netG": synth.make_state_dict(0, "auto")})
            sr.sub_areas = [box]
            if ab:
                sr.ab_sections = [range(2, 9), range(13, 19)]
            sr.video_out_path = str(tmp_path / f"out_{mode}.y4m")
            ticks = []
            sr.update_progress = lambda tbar, increment: ticks.append(increment)
            sr.sttn_auto_mode(tbar=object())
            sr.video_writer.release()
            outs[mode] = open(sr.video_out_path, "rb").read()
            assert sum(ticks) == N
    finally:
        for k, v in old.items():
            getattr(config, k).value = v
        config.inpaintMode.value = old_mode
    assert outs["host"] == outs["device-frames"] == outs["resident"] == outs["device-frames-late-pins"] == outs["resident-late-pins"]
    assert outs["by-offset"] == outs["host"] and outs["by-offset-late-pins"] == outs["host"]
    monkeypatch.setenv("VSR_IO_COLOR", "host")
    got, want = _read_all(str(tmp_path / "out_resident.y4m")), _read_all(src)
    assert got.shape == want.shape
    ymin, ymax, xmin, xmax = box
    assert (got[:, ymin:ymax, xmin:xmax] != want[:, ymin:ymax, xmin:xmax]).mean() > (0.15 if ab else 0.4)


def test_bt601_colour_bars_on_the_device(built_lib, gpu_device, tmp_path, monkeypatch):
    """known answers: the published BT.601 colour-bar code values out of vsr_io_bgr_to_yuv, the primaries back out of vsr_io_yuv_to_bgr"""
    from tests.test_video_io import BT601_BARS

    monkeypatch.setenv("VSR_IO_COLOR", "device")
    bars = list(BT601_BARS.items())
    H, W = 2, 4 * len(bars)
    frame = np.zeros((H, W, 3), np.uint8)
    for i, ((r, g, b), _) in enumerate(bars):
        frame[:, 4 * i: 4 * i + 4] = (b, g, r)
    p = str(tmp_path / "bars.y4m")
    w = video_io.Y4mWriter(p, 25.0, (W, H), chroma="444")
    assert w._dc is not None
    w.write(frame)
    w.release()
    raw = open(p, "rb").read()
    rec = np.frombuffer(raw[raw.index(b"FRAME\n") + 6:], np.uint8).reshape(3, H, W)
    for i, (_, (y, cb, cr)) in enumerate(bars):
        assert (int(rec[0, 0, 4 * i]), int(rec[1, 0, 4 * i]), int(rec[2, 0, 4 * i])) == (y, cb, cr)
    back = _read_all(p)[0]
    assert np.abs(back.astype(int) - frame.astype(int)).max() <= 1


@pytest.mark.parametrize("mode", ["sttn-det", "lama", "propainter"])
def test_resident_detector_modes_write_the_same_file(built_lib, gpu_device, tmp_path, monkeypatch, mode):
    """*.y4m in -> SubtitleRemover.run() in a detector-driven mode -> *.y4m out (round 3, tools/resident.py): with the decoded video
    resident in HBM -- planes up once, the detector sampling its frames from the device tensor, the scene-cut kernels reading it, the
    plugin working in place on slices, planes down once -- the written file is byte for byte the one the host-frame loop writes.
    The subtitle is on screen in two intervals with a gap (pass-through frames), the second one ends with the clip."""
    from vsr_amd import synth
    from vsr_amd.backend.config import config
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.constant import InpaintMode

    # (RAFT wants strips of at least 128 rows: the propainter case runs on a larger frame)
    H, W, N = (480, 852, 34) if mode == "propainter" else (240, 432, 34)
    box = (400, 450, 100, 760) if mode == "propainter" else (180, 214, 60, 380)        # ymin, ymax, xmin, xmax
    clip = synth.make_clip(N, H, W, box, seed=5)
    on = [i for i in range(N) if 3 <= i < 15 or i >= 22]
    plain = synth.make_clip(N, H, W, (0, 1, 0, 1), seed=5)
    for i in range(N):
        if i not in on:
            clip[i] = plain[i]
    src = str(tmp_path / "in.y4m")
    monkeypatch.setenv("VSR_IO_COLOR", "host")
    w = video_io.Y4mWriter(src, 25.0, (W, H), chroma="420")
    for f in clip:
        w.write(f)
    w.release()
    quad = np.array([[[box[2], box[0]], [box[3], box[0]], [box[3], box[1]], [box[2], box[1]]]])

    class Det:                                             # the reference's host signature; sees device-decoded or host-decoded frames alike
        batch_size = 4

        def __init__(self):
            self.calls = 0

        def predict(self, img):
            self.calls += 1
            white = (img[box[0] + 8:box[1] - 8, box[2] + 8:box[3] - 8] > 200).mean()
            return [{"dt_polys": quad if white > 0.05 else np.zeros((0, 4, 2), np.int32)}]

    if mode == "sttn-det":
        from vsr_amd.backend.inpaint.sttn_det_inpaint import STTNDetInpaint
        plugin = STTNDetInpaint("cuda:0", {"netG": synth.make_state_dict(0, "det")})
    elif mode == "lama":
        from vsr_amd.backend.inpaint.lama_inpaint import LamaInpaint
        plugin = LamaInpaint("cuda:0", synth.make_lama_state_dict(3, 2))
    else:
        from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
        plugin = PropainterInpaint("cuda:0", {"raft": synth.make_raft_state_dict(0), "rfc": synth.make_rfc_state_dict(0),
                                              "propainter": synth.make_propainter_state_dict(0)})
        plugin.raft_iter = 4
    keys = {"sttnMaxLoadNum": 8, "propainterMaxLoadNum": 8}
    old = {k: getattr(config, k).value for k in keys}
    old_mode = config.inpaintMode.value
    outs, phases = {}, {}
    try:
        for k, v in keys.items():
            getattr(config, k).value = v
        config.inpaintMode.value = {"sttn-det": InpaintMode.STTN_DET, "lama": InpaintMode.LAMA, "propainter": InpaintMode.PROPAINTER}[mode]
        # opt-in third pass (VSR_TEST_BATCH_LANES=1): the resident run again over two batch lanes (tools/batch_lanes.py, off by default)
        passes = [("host", "host", "0", "1"), ("resident", "device", "1", "1")]
        if os.environ.get("VSR_TEST_BATCH_LANES") == "1":
            passes.append(("lanes", "device", "1", "2"))
        for how, color, resident, lanes in passes:
            monkeypatch.setenv("VSR_IO_COLOR", color)
            monkeypatch.setenv("VSR_IO_RESIDENT", resident)
            monkeypatch.setenv("VSR_BATCH_LANES", lanes)
            sr = SubtitleRemover(src, device="cuda:0")
            sr.sub_areas = [(0, H, 0, W)]
            det = Det()
            sr.video_out_path = str(tmp_path / f"out_{how}.y4m")
            ticks = []
            sr.update_progress = lambda tbar, increment: ticks.append(increment)
            if mode == "propainter":
                sr.propainter_mode(object(), propainter_inpaint=plugin, text_detector=det, single_frame_inpaint=None)
            else:
                sr.video_inpaint(object(), plugin, text_detector=det)
            sr.video_writer.release()
            outs[how] = open(sr.video_out_path, "rb").read()
            phases[how] = dict(sr.phase_seconds)
            assert sum(ticks) == N and det.calls >= N // 3
    finally:
        for k, v in old.items():
            getattr(config, k).value = v
        config.inpaintMode.value = old_mode
        if hasattr(plugin, "close"):
            plugin.close()
    assert "read + upload + YUV->BGR" in phases["resident"] and "read + upload + YUV->BGR" not in phases["host"]
    assert outs["host"] == outs["resident"], "the HBM-resident loop must write the host loop's file"
    if "lanes" in outs:
        assert outs["lanes"] == outs["resident"], "batch lanes must not change a byte"
    monkeypatch.setenv("VSR_IO_COLOR", "host")
    got, want = _read_all(str(tmp_path / "out_resident.y4m")), _read_all(src)
    assert got.shape == want.shape
    changed = (got != want).any(axis=(1, 2, 3))
    assert changed[on].mean() > 0.7        # (the timeline expansion / interval merging of main.py decide about the frames around)
