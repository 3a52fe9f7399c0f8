"""Frame sources / sinks of tools/video_io.py and the CLI's output contract (`-i IN -o OUT` must write OUT or fail)."""
import os

import numpy as np
import pytest

from vsr_amd.backend.tools import video_io as v


def _clip(n=9, H=38, W=52, seed=0):
    yy, xx = np.mgrid[0:H, 0:W]
    base = np.stack([(xx * 3 + yy) % 256, (xx + yy * 2) % 256, (xx * 2 + 90) % 256], axis=-1)
    return np.stack([np.roll(base, i, axis=1) for i in range(n)]).astype(np.uint8)


def test_npy_container_is_lossless(tmp_path):
    clip = _clip()
    for frames in (None, len(clip), len(clip) + 3, len(clip) - 2):          # unknown, exact, over- and under-estimated frame count
        p = str(tmp_path / f"a{frames}.npy")
        w = v.AsyncWriter(v.open_writer(p, 25.0, (clip.shape[2], clip.shape[1]), frames=frames))
        for f in clip:
            w.write(f)
        w.release()
        w.release()                                                        # plugin and run() both release: idempotent
        r = v.open_video(p)
        assert r.info() == {"W_ori": 52, "H_ori": 38, "fps": 25.0, "len": len(clip)}
        got = []
        while True:
            ok, f = r.read()
            if not ok:
                break
            got.append(f)
        assert np.array_equal(np.stack(got), clip)
        r.release()


@pytest.mark.parametrize("chroma,tol", [("444", 2), ("420", 40)])
def test_y4m_round_trip(tmp_path, chroma, tol):
    clip = _clip(H=37, W=51)                                                # odd sizes: chroma planes are rounded up
    p = str(tmp_path / "c.y4m")
    w = v.Y4mWriter(p, 30000 / 1001, (51, 37), chroma=chroma)
    for f in clip:
        w.write(f)
    w.release()
    head = open(p, "rb").readline()
    assert head.startswith(b"YUV4MPEG2 W51 H37 F30000:1001 Ip") and (b"C444" in head) == (chroma == "444")
    r = v.FramePrefetcher(v.open_video(p), buffer_size=3)                   # the reference's threaded reader contract
    info = r.info()
    assert (info["W_ori"], info["H_ori"], info["len"]) == (51, 37, len(clip)) and abs(info["fps"] - 29.97) < 0.01
    got = np.stack([r.read()[1] for _ in range(len(clip))])
    assert r.read()[0] is False
    r.release()
    err = np.abs(got.astype(int) - clip.astype(int))
    assert err.max() <= tol and (chroma == "420" or err.mean() < 0.6)


def test_writer_for_a_codec_container_fails_loudly_without_a_tool(tmp_path, monkeypatch):
    monkeypatch.delenv("VSR_FFMPEG", raising=False)
    if v.ffmpeg_path() is not None:
        pytest.skip("an ffmpeg binary exists on this machine")
    try:
        import cv2  # noqa: F401
        pytest.skip("opencv exists on this machine")
    except ImportError:
        pass
    with pytest.raises(RuntimeError, match="cannot write"):
        v.open_writer(str(tmp_path / "x.mp4"), 25, (52, 38))
    with pytest.raises(RuntimeError, match="cannot read"):
        v.open_video(str(tmp_path / "x.mp4"))


@pytest.fixture
def fake_ffmpeg(tmp_path, monkeypatch):
    """an `ffmpeg` / `ffprobe` pair (tests/tools/fake_ffmpeg.py) on the other end of the product's pipes"""
    import stat
    import sys

    tool = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "fake_ffmpeg.py")
    d = tmp_path / "bin"
    d.mkdir()
    for name in ("ffmpeg", "ffprobe"):
        sh = d / name
        sh.write_text(f"#!/bin/sh\nFAKE_TOOL={name} exec {sys.executable} {tool} \"$@\"\n")
        sh.chmod(sh.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("VSR_FFMPEG", str(d / "ffmpeg"))
    return d


def _write_fakevid(path, clip, fps="25/1", rotation=0, known=True):
    import json

    n, h, w, _ = clip.shape
    with open(path, "wb") as f:
        f.write(b"FAKEVID1\n")
        f.write(json.dumps({"w": w, "h": h, "fps": fps, "frames": n, "rotation": rotation, "known": known}).encode() + b"\n")
        f.write(clip.tobytes())


def test_ffmpeg_pipes_round_trip(tmp_path, fake_ffmpeg):
    """the reference's transport (video_io.py:54-103 writer, cv2/ffmpeg reader): frames through a real pipe to a process and back"""
    clip = np.random.default_rng(2).integers(0, 256, (7, 20, 36, 3), dtype=np.uint8)
    out = str(tmp_path / "o.mp4")
    w = v.open_writer(out, 25, (36, 20))
    assert type(getattr(w, "sink", w)).__name__ == "FFmpegVideoWriter" or type(w).__name__ in ("FFmpegVideoWriter", "AsyncWriter")
    for f in clip:
        w.write(f)
    w.release()
    r = v.open_video(out)
    info = r.info()
    assert (info["W_ori"], info["H_ori"], info["len"]) == (36, 20, 7) and info["fps"] == 25.0
    got = np.stack([r.read()[1] for _ in range(7)])
    assert r.read()[0] is False
    r.release()
    assert np.array_equal(got, clip)


def test_ffmpeg_reader_rotation_and_unknown_length(tmp_path, fake_ffmpeg):
    """a quarter-turn display matrix swaps the sides of the frames that arrive (ffmpeg autorotates, as cv2 does);
    a container that does not know its packet count ('N/A') falls back to duration x rate"""
    clip = np.random.default_rng(3).integers(0, 256, (4, 12, 30, 3), dtype=np.uint8)
    src = str(tmp_path / "phone.mp4")
    _write_fakevid(src, clip, fps="30000/1001", rotation=-90, known=False)
    r = v.open_video(src)
    info = r.info()
    assert (info["W_ori"], info["H_ori"], info["len"]) == (12, 30, 4) and abs(info["fps"] - 29.97) < 0.01
    ok, f0 = r.read()
    assert ok and f0.shape == (30, 12, 3) and np.array_equal(f0, np.rot90(clip[0], -1))
    r.release()


@pytest.mark.parametrize("how", ["encode", "exit"])
def test_ffmpeg_writer_reports_a_dead_or_failing_encoder(tmp_path, fake_ffmpeg, monkeypatch, how):
    """a broken pipe used to be ignored and the exit code never read -- 'written' over a truncated file"""
    monkeypatch.setenv("FAKE_FFMPEG_FAIL", how)
    w = v.FFmpegVideoWriter(str(tmp_path / "o.mp4"), 25, (64, 48))
    frame = np.zeros((48, 64, 3), np.uint8)
    with pytest.raises(RuntimeError, match="out of tea|muxer said no|exit code"):
        for _ in range(20000):                 # the pipe buffer absorbs the first frames after the encoder died
            w.write(frame)
        w.release()


def test_async_writer_surfaces_sink_errors():
    class Bad:
        def write(self, f):
            raise IOError("disk full")

        def release(self):
            pass

    w = v.AsyncWriter(Bad(), buffer_size=2)
    w.write(np.zeros((2, 2, 3), np.uint8))
    with pytest.raises(IOError):
        for _ in range(50):
            w.write(np.zeros((2, 2, 3), np.uint8))
        w.release()


@pytest.mark.parametrize("out_ext", [".npy", ".y4m"])
def test_run_writes_the_output_file(tmp_path, out_ext):
    """SubtitleRemover(path).run() streams to `video_out_path` (here in lama mode with a stand-in plugin on the CPU: the driver
    and the IO are what is under test)"""
    from vsr_amd.backend.config import config
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.constant import InpaintMode

    clip = _clip(n=24, H=90, W=160)
    src = str(tmp_path / "in.npy")
    np.save(src, clip)
    quad = np.array([[[40, 60], [120, 60], [120, 75], [40, 75]]])

    class Det:
        def predict(self, img):
            return [{"dt_polys": quad}]

    def plugin(frames, mask):
        return [np.where(mask[:, :, None] > 0, 255 - f, f) for f in frames]

    old = config.inpaintMode.value
    config.inpaintMode.value = InpaintMode.LAMA
    try:
        sr = SubtitleRemover(src, device="cpu")
        assert sr.video_out_path.endswith("in_no_sub.npy")
        sr.video_out_path = str(tmp_path / ("out" + out_ext))
        sr.sub_areas = [(0, 90, 0, 160)]
        sr.text_detector = Det()
        sr.lama_inpaint = plugin
        sr.run()
    finally:
        config.inpaintMode.value = old
    assert os.path.exists(sr.video_out_path)
    r = v.open_video(sr.video_out_path)
    assert r.info()["len"] == 24
    got = np.stack([r.read()[1] for _ in range(24)])
    from vsr_amd.backend.tools.inpaint_tools import create_mask

    m = create_mask((90, 160), [(40, 120, 60, 75)]) > 0
    want = np.where(m[None, :, :, None], 255 - clip, clip)
    if out_ext == ".npy":
        assert np.array_equal(got, want)
    else:
        assert np.abs(got.astype(int) - want.astype(int)).max() <= 2


def test_lama_mode_without_weights_is_an_error(tmp_path, monkeypatch):
    from vsr_amd.backend.config import config
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools.constant import InpaintMode
    from vsr_amd.backend.tools.video_io import ArrayVideo

    monkeypatch.delenv("LAMA_MODEL_PATH", raising=False)
    old = config.inpaintMode.value
    config.inpaintMode.value = InpaintMode.LAMA
    try:
        sr = SubtitleRemover(ArrayVideo(_clip()), device="cpu")
        with pytest.raises(Exception, match="lama needs its weights"):
            sr.run()
    finally:
        config.inpaintMode.value = old


def test_y4m_planes_access_round_trip(tmp_path):
    """the raw access the HBM-resident chunk loop uses (planes as stored, no conversion): records read with read_planes_into and
    written back with write_planes reproduce the file; without a GPU both ends report that no device conversion is available"""
    from vsr_amd.backend.tools import video_io

    H, W, N = 18, 26, 7
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, size=(N, H, W, 3), dtype=np.uint8)
    src, dst = str(tmp_path / "a.y4m"), str(tmp_path / "b.y4m")
    w = video_io.Y4mWriter(src, 25.0, (W, H), chroma="420")
    for f in frames:
        w.write(f)
    w.release()
    r = video_io.Y4mVideo(src)
    fsize = W * H + 2 * ((W + 1) // 2) * ((H + 1) // 2)
    buf = np.zeros((5, fsize), np.uint8)
    w2 = video_io.Y4mWriter(dst, 25.0, (W, H), chroma="420")
    got = 0
    while True:
        k = r.read_planes_into(buf)
        if k == 0:
            break
        w2.write_planes(buf[:k])
        got += k
    r.release()
    w2.release()
    assert got == N
    assert open(src, "rb").read() == open(dst, "rb").read()
    import torch

    if not torch.cuda.is_available():
        assert video_io.Y4mVideo(src).planes_format() is None and video_io.Y4mWriter(str(tmp_path / "c.y4m"), 25.0, (W, H)).planes_format() is None
    aw = video_io.AsyncWriter(video_io.Y4mWriter(str(tmp_path / "d.y4m"), 25.0, (W, H), chroma="420"))
    r = video_io.Y4mVideo(src)
    k = r.read_planes_into(buf)
    aw.write_planes(buf[:k])                      # batches keep their place between frames in the writer thread's queue
    aw.write(frames[5])
    aw.release()
    out = video_io.Y4mVideo(str(tmp_path / "d.y4m"))
    assert out.info()["len"] == 6


# the 100 % colour bars of ITU-R BT.601 in 8-bit studio range, as every table of the standard's worked values lists them
BT601_BARS = {  # (R, G, B) -> (Y, Cb, Cr)
    (255, 255, 255): (235, 128, 128), (255, 255, 0): (210, 16, 146), (0, 255, 255): (170, 166, 16), (0, 255, 0): (145, 54, 34),
    (255, 0, 255): (106, 202, 222), (255, 0, 0): (81, 90, 240), (0, 0, 255): (41, 240, 110), (0, 0, 0): (16, 128, 128)}


def test_bt601_colour_bars_known_answers():
    """the integer matrices of the y4m transport reproduce the published BT.601 colour-bar code values (the numpy statement the
    GPU kernels are held to, tests/test_gpu_io.py), and the way back lands within one level of the primaries"""
    from vsr_amd.backend.tools import video_io

    for (r, g, b), want in BT601_BARS.items():
        y, u, v = (int(p[0, 0]) for p in video_io._bgr_to_yuv(np.array([[[b, g, r]]], np.uint8), False))
        assert (y, u, v) == want, ((r, g, b), (y, u, v), want)
        back = video_io._yuv_to_bgr(np.array([[y]], np.uint8), np.array([[u]], np.uint8), np.array([[v]], np.uint8), False)[0, 0]
        assert np.abs(back.astype(int) - np.array([b, g, r])).max() <= 1
