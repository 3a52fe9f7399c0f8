"""Frame sources / sinks for the plugins (reference backend/tools/video_io.py: FramePrefetcher :12-51, FFmpegVideoWriter :54-103).

Codec work stays out of scope: nothing here decodes or encodes a compressed stream itself.  What is
provided is what the hot path needs to be a drop-in -- `read() -> (ok, frame)` / `write(frame)` objects over containers of RAW
frames, an ffmpeg pipe when a binary exists (the reference's own transport, same command line), threads on both sides so that
reading and writing overlap the GPU work, and a loud failure when nothing can serve a path:

  source                      sink
  ArrayVideo (in memory)      ArrayWriter / CountingWriter
  *.npy   uint8 [N,H,W,3] BGR NpyWriter         (memory-mapped, lossless: the parity container)
  *.y4m   YUV4MPEG2           Y4mWriter         (C444 / C420 / mono, BT.601; what `ffmpeg -i x.mp4 x.y4m` writes; the colour
                                                conversion runs on the GPU, csrc/io_kernels.hip)
  anything else               FFmpegVideoWriter (needs an `ffmpeg` on PATH or $VSR_FFMPEG; else cv2 if importable; else an error)
"""
import os
import queue
import shutil
import subprocess
import threading

import numpy as np


# ------------------------------------------------------------------------------------------------------------------------
# in-memory
# ------------------------------------------------------------------------------------------------------------------------
class ArrayVideo:
    """In-memory clip: uint8 [N,H,W,3] BGR.  Stands in for a video path."""

    def __init__(self, frames, fps=30.0):
        self.frames = frames
        self.fps = float(fps)
        self._pos = 0

    def info(self):
        n, h, w, _ = self.frames.shape
        return {"W_ori": int(w), "H_ori": int(h), "fps": self.fps, "len": int(n)}

    def read(self):
        if self._pos >= len(self.frames):
            return False, None
        f = self.frames[self._pos]
        self._pos += 1
        return True, f

    def release(self):
        pass


class ArrayWriter:
    """Collects written frames (the reference's writer converts non-uint8 by clip+cast, video_io.py:85-92)."""

    def __init__(self):
        self.frames = []

    def write(self, frame):
        if frame.dtype != np.uint8:
            frame = np.clip(frame, 0, 255).astype(np.uint8)
        self.frames.append(frame.copy())        # the caller may hand out views of a recycled (pinned) buffer

    def release(self):
        pass


class CountingWriter:
    """Sink that only counts (benchmarks: the encoder is out of scope)."""

    def __init__(self):
        self.count = 0
        self.checksum = 0

    def write(self, frame):
        self.count += 1
        self.checksum = (self.checksum + int(frame[::97, ::89].sum())) & 0xFFFFFFFF

    def release(self):
        pass


# ------------------------------------------------------------------------------------------------------------------------
# raw containers
# ------------------------------------------------------------------------------------------------------------------------
class NpyVideo:
    """*.npy holding uint8 [N,H,W,3] BGR frames, memory-mapped."""

    def __init__(self, path, fps=None):
        self.frames = np.load(path, mmap_mode="r")
        if self.frames.ndim != 4 or self.frames.shape[3] != 3 or self.frames.dtype != np.uint8:
            raise RuntimeError(f"{path}: expected a uint8 array [N,H,W,3], found {self.frames.dtype} {self.frames.shape}")
        side = os.path.splitext(path)[0] + ".fps"
        self.fps = float(fps) if fps else (float(open(side).read()) if os.path.exists(side) else 30.0)
        self._pos = 0

    def info(self):
        n, h, w, _ = self.frames.shape
        return {"W_ori": int(w), "H_ori": int(h), "fps": self.fps, "len": int(n)}

    def read(self):
        if self._pos >= self.frames.shape[0]:
            return False, None
        f = np.array(self.frames[self._pos])
        self._pos += 1
        return True, f

    def release(self):
        self.frames = None


class NpyWriter:
    """Appends frames to a memory-mapped *.npy (header rewritten on release with the final count)."""

    def __init__(self, path, fps, size, capacity=None):
        self.path, self.fps = path, float(fps)
        self.w, self.h = int(size[0]), int(size[1])
        self.capacity = int(capacity) if capacity else 0
        self.n = 0
        self._mm = None
        self._done = False
        self._tmp = path + ".part"
        self._raw = open(self._tmp, "wb") if not self.capacity else None
        if self.capacity:
            self._mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.uint8, shape=(self.capacity, self.h, self.w, 3))

    def write(self, frame):
        if frame.dtype != np.uint8:
            frame = np.clip(frame, 0, 255).astype(np.uint8)
        if frame.shape != (self.h, self.w, 3):
            raise ValueError(f"frame {frame.shape} does not match the writer's {(self.h, self.w, 3)}")
        if self._mm is not None and self.n < self.capacity:
            self._mm[self.n] = frame
        else:
            if self._raw is None:
                self._raw = open(self._tmp, "wb")
            self._raw.write(np.ascontiguousarray(frame).tobytes())
        self.n += 1

    def release(self):
        if self._done:
            return
        self._done = True
        if self._mm is not None and self._raw is None and self.n == self.capacity:
            self._mm.flush()
            self._mm = None
        else:                                       # unknown or wrong frame count: assemble the final file
            head = np.array(self._mm[: min(self.n, self.capacity)]) if self._mm is not None else None
            self._mm = None
            if self._raw is not None:
                self._raw.close()
            out = np.lib.format.open_memmap(self.path + ".new", mode="w+", dtype=np.uint8, shape=(self.n, self.h, self.w, 3))
            k = 0
            if head is not None:
                out[: head.shape[0]] = head
                k = head.shape[0]
            if os.path.exists(self._tmp) and self.n > k:
                out[k:] = np.fromfile(self._tmp, dtype=np.uint8).reshape(self.n - k, self.h, self.w, 3)
            out.flush()
            del out
            os.replace(self.path + ".new", self.path)
        if os.path.exists(self._tmp):
            os.remove(self._tmp)
        with open(os.path.splitext(self.path)[0] + ".fps", "w") as f:
            f.write(repr(self.fps))


# BT.601 studio-swing integer matrices (the ones libswscale and OpenCV use for 8-bit YCbCr <-> RGB), 16.16 fixed point
def _yuv_to_bgr(y, u, v, full_range):
    y = y.astype(np.int32)
    u = u.astype(np.int32) - 128
    v = v.astype(np.int32) - 128
    if full_range:
        c = y << 16
        r = (c + 91881 * v + 32768) >> 16
        g = (c - 22554 * u - 46802 * v + 32768) >> 16
        b = (c + 116130 * u + 32768) >> 16
    else:
        c = 76309 * (y - 16)
        r = (c + 104597 * v + 32768) >> 16
        g = (c - 25675 * u - 53279 * v + 32768) >> 16
        b = (c + 132201 * u + 32768) >> 16
    return np.clip(np.stack([b, g, r], axis=-1), 0, 255).astype(np.uint8)


def _bgr_to_yuv(frame, full_range):
    b, g, r = (frame[..., k].astype(np.int32) for k in range(3))
    if full_range:
        y = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16
        u = ((-11059 * r - 21709 * g + 32768 * b + 32768) >> 16) + 128
        v = ((32768 * r - 27439 * g - 5329 * b + 32768) >> 16) + 128
    else:
        y = ((16829 * r + 33039 * g + 6416 * b + 32768) >> 16) + 16
        u = ((-9714 * r - 19070 * g + 28784 * b + 32768) >> 16) + 128
        v = ((28784 * r - 24103 * g - 4681 * b + 32768) >> 16) + 128
    return (np.clip(p, 0, 255).astype(np.uint8) for p in (y, u, v))


def _device_color_enabled():
    """the GPU does the colour conversion whenever there is one (VSR_IO_COLOR=host keeps the numpy statement, which is also what a
    machine without a GPU -- where nothing but IO can run anyway -- uses)"""
    if os.environ.get("VSR_IO_COLOR", "device") == "host":
        return False
    try:
        import torch

        return torch.cuda.is_available()
    except ImportError:
        return False


class _DeviceColor:
    """Planar YCbCr <-> packed BGR on the GPU (csrc/io_kernels.hip: vsr_io_yuv_to_bgr / vsr_io_bgr_to_yuv, the integer matrices
    of _yuv_to_bgr / _bgr_to_yuv bit for bit).  numpy needs 47 ms to turn one 1080p 4:2:0 frame into BGR and 26 ms for the way
    back -- the whole 50-frame STTN chunk takes 310 ms on the GPU -- so the raw planes go up as they are on disk (half the bytes
    of the BGR frame), the kernel converts, and the frames come back.  Frames move in batches of up to `batch` (one upload, one
    launch, one download, one synchronisation per batch: beside a busy compute stream every single operation waits for a slot).
    One instance per reader / writer (own stream and pinned staging: the reader works in the caller's or the prefetch thread, the
    writer in AsyncWriter's)."""

    def __init__(self, H, W, frame_bytes, batch=8):
        import ctypes as C

        import torch

        from ..._lib import check, lib

        self.torch, self.C, self.check, self.lib = torch, C, check, lib
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.H, self.W, self.frame_bytes, self.batch = H, W, frame_bytes, max(1, int(batch))
        self.stream = torch.cuda.Stream(self.dev)
        B, u8 = self.batch, torch.uint8
        with torch.cuda.stream(self.stream):
            self.d_planes = torch.empty((B, frame_bytes), dtype=u8, device=self.dev)
            self.d_bgr = torch.empty((B, H, W, 3), dtype=u8, device=self.dev)
        self.pin_planes = torch.empty((B, frame_bytes), dtype=u8).pin_memory()
        self.pin_bgr = torch.empty((B, H, W, 3), dtype=u8).pin_memory()

    def _p(self, t):
        return self.C.c_void_p(t.data_ptr())

    def planes_buffer(self):
        """pinned host records [batch][frame_bytes]: where the reader puts stored frames / where from_bgr leaves converted ones"""
        return self.pin_planes.numpy()

    def bgr_buffer(self):
        """pinned host frames [batch][H][W][3]: where the writer collects frames / where to_bgr leaves converted ones"""
        return self.pin_bgr.numpy()

    def to_bgr(self, n, cw, ch, full_range):
        """the first n records of planes_buffer() -> the first n frames of bgr_buffer()"""
        t = self.torch
        with t.cuda.stream(self.stream):
            self.d_planes[:n].copy_(self.pin_planes[:n], non_blocking=True)
            self.check(self.lib.vsr_io_yuv_to_bgr(self._p(self.d_planes), self.frame_bytes, self.H, self.W, cw, ch, int(bool(full_range)),
                                                  self._p(self.d_bgr), n, self.C.c_void_p(self.stream.cuda_stream)))
            self.pin_bgr[:n].copy_(self.d_bgr[:n], non_blocking=True)
        self.stream.synchronize()
        return self.pin_bgr.numpy()[:n]

    def from_bgr(self, n, subsample_420, full_range):
        """the first n frames of bgr_buffer() -> the first n records of planes_buffer()"""
        t = self.torch
        with t.cuda.stream(self.stream):
            self.d_bgr[:n].copy_(self.pin_bgr[:n], non_blocking=True)
            self.check(self.lib.vsr_io_bgr_to_yuv(self._p(self.d_bgr), self.H, self.W, int(bool(subsample_420)), int(bool(full_range)),
                                                  self._p(self.d_planes), self.frame_bytes, n, self.C.c_void_p(self.stream.cuda_stream)))
            self.pin_planes[:n].copy_(self.d_planes[:n], non_blocking=True)
        self.stream.synchronize()
        return self.pin_planes.numpy()[:n]


class Y4mVideo:
    """YUV4MPEG2 reader: 8-bit C420* / C422 / C444 / Cmono, progressive; frames come out as BGR (BT.601)."""

    def __init__(self, path):
        self.path = path
        self._f = open(path, "rb")
        head = self._f.readline()
        if not head.startswith(b"YUV4MPEG2"):
            raise RuntimeError(f"{path}: not a YUV4MPEG2 stream")
        self.w = self.h = 0
        self.fps, self.chroma = 30.0, "420jpeg"
        for tok in head.split()[1:]:
            t = tok.decode("ascii", "replace")
            if t[0] == "W":
                self.w = int(t[1:])
            elif t[0] == "H":
                self.h = int(t[1:])
            elif t[0] == "F":
                n, d = t[1:].split(":")
                self.fps = float(n) / float(d or 1)
            elif t[0] == "C":
                self.chroma = t[1:]
            elif t[0] == "I" and t[1:] not in ("p", "?"):
                raise RuntimeError(f"{path}: interlaced y4m ({t}) is not supported")
        if self.chroma.startswith("420"):
            self.cw, self.ch = (self.w + 1) // 2, (self.h + 1) // 2
        elif self.chroma.startswith("422"):
            self.cw, self.ch = (self.w + 1) // 2, self.h
        elif self.chroma.startswith("444"):
            self.cw, self.ch = self.w, self.h
        elif self.chroma == "mono":
            self.cw = self.ch = 0
        else:
            raise RuntimeError(f"{path}: chroma format C{self.chroma} is not supported (8-bit 420 / 422 / 444 / mono)")
        if any(s in self.chroma for s in ("p10", "p12", "p14", "p16")):
            raise RuntimeError(f"{path}: only 8-bit y4m is supported")
        self.full_range = b"XCOLORRANGE=FULL" in head
        self._data0 = self._f.tell()
        self._fsize = self.w * self.h + 2 * self.cw * self.ch
        self._count = None
        self._dc_args = (self.h, self.w, self._fsize) if _device_color_enabled() else None     # the converter is built on first use
        self._dc_obj = None
        self._ready = []                              # frames converted ahead (device path): popped from the end

    @property
    def _dc(self):
        if self._dc_obj is None and self._dc_args is not None:
            self._dc_obj = _DeviceColor(*self._dc_args)
        return self._dc_obj

    def info(self):
        if self._count is None:                       # constant-size records: "FRAME\n" + planes (frame parameters are rare; counted exactly)
            total = os.path.getsize(self.path) - self._data0
            rec = 6 + self._fsize
            if total % rec == 0:
                self._count = total // rec
            else:
                pos = self._f.tell()
                self._f.seek(self._data0)
                n = 0
                while True:
                    line = self._f.readline()
                    if not line.startswith(b"FRAME"):
                        break
                    self._f.seek(self._fsize, 1)
                    n += 1
                self._f.seek(pos)
                self._count = n
        return {"W_ori": self.w, "H_ori": self.h, "fps": self.fps, "len": int(self._count)}

    def read(self):
        if self._dc is not None:                      # planes as stored -> pinned memory -> GPU -> BGR frames, a batch at a time
            if not self._ready:
                n = self.read_planes_into(self._dc.planes_buffer())
                if n == 0:
                    return False, None
                bgr = self._dc.to_bgr(n, self.cw, self.ch, self.full_range)
                self._ready = [bgr[k].copy() for k in range(n - 1, -1, -1)]       # frames of their own: the plugins patch rows into them
            return True, self._ready.pop()
        line = self._f.readline()
        if not line.startswith(b"FRAME"):
            return False, None
        buf = self._f.read(self._fsize)
        if len(buf) < self._fsize:
            return False, None
        a = np.frombuffer(buf, dtype=np.uint8)
        y = a[: self.w * self.h].reshape(self.h, self.w)
        if self.cw == 0:
            return True, np.repeat(_yuv_to_bgr(y, np.full_like(y, 128), np.full_like(y, 128), self.full_range)[..., :1], 3, axis=2)
        n = self.cw * self.ch
        u = a[self.w * self.h: self.w * self.h + n].reshape(self.ch, self.cw)
        v = a[self.w * self.h + n:].reshape(self.ch, self.cw)
        if (self.ch, self.cw) != (self.h, self.w):    # nearest chroma up-sampling
            ry, rx = (1 if self.ch == self.h else 2), (1 if self.cw == self.w else 2)
            u = np.repeat(np.repeat(u, ry, axis=0), rx, axis=1)[: self.h, : self.w]
            v = np.repeat(np.repeat(v, ry, axis=0), rx, axis=1)[: self.h, : self.w]
        return True, _yuv_to_bgr(y, u, v, self.full_range)

    # raw access for the device-resident chunk loop (STTNAutoInpaint._run): the planes travel as stored, the GPU converts
    def planes_format(self):
        """None without a GPU; else the layout of one stored frame: [Y: H*W][U: ch*cw][V: ch*cw], `frame_bytes` in all"""
        if self._dc_args is None:
            return None
        if self._ready:
            raise RuntimeError("read() and read_planes_into() cannot be mixed on one reader")
        return {"frame_bytes": self._fsize, "cw": self.cw, "ch": self.ch, "full_range": self.full_range}

    def record_layout(self):
        """{path, data_offset, prefix, frame_bytes, count} when every frame of the file is one fixed-size record (no per-frame parameters):
        what tools/rank_io.py needs to read a chunk by offset; None otherwise"""
        total = os.path.getsize(self.path) - self._data0
        rec = 6 + self._fsize
        if total % rec != 0:
            return None
        return {"path": os.path.abspath(self.path), "data_offset": self._data0, "prefix": b"FRAME\n", "frame_bytes": self._fsize, "count": total // rec}

    def read_planes_into(self, out):
        """fill out[k] (uint8 [n][frame_bytes], e.g. pinned memory) with the next frames as stored; returns how many were read"""
        k = 0
        while k < out.shape[0]:
            line = self._f.readline()
            if not line.startswith(b"FRAME") or self._f.readinto(out[k]) < self._fsize:
                break
            k += 1
        return k

    def release(self):
        self._f.close()


class Y4mWriter:
    """YUV4MPEG2 writer, 8-bit, BT.601 studio range; chroma "444" (default: no sub-sampling loss) or "420"."""

    def __init__(self, path, fps, size, chroma="444"):
        self.w, self.h = int(size[0]), int(size[1])
        self.chroma = chroma
        num, den = (int(round(fps * 1001)), 1001) if abs(fps - round(fps)) > 1e-3 else (int(round(fps)), 1)
        self._f = open(path, "wb", buffering=1 << 22)
        tag = "444" if chroma == "444" else "420mpeg2"
        self._f.write(f"YUV4MPEG2 W{self.w} H{self.h} F{num}:{den} Ip A1:1 C{tag} XCOLORRANGE=LIMITED\n".encode())
        self._header_bytes = self._f.tell()
        cw, ch = (self.w, self.h) if chroma == "444" else ((self.w + 1) // 2, (self.h + 1) // 2)
        self._dc_args = (self.h, self.w, self.w * self.h + 2 * cw * ch) if _device_color_enabled() else None
        self._dc_obj = None
        self._pending = 0

    @property
    def _dc(self):
        if self._dc_obj is None and self._dc_args is not None:
            self._dc_obj = _DeviceColor(*self._dc_args)
        return self._dc_obj

    def _flush(self):
        if self._pending:
            n, self._pending = self._pending, 0
            self.write_planes(self._dc.from_bgr(n, self.chroma != "444", False))

    def write(self, frame):
        if frame.dtype != np.uint8:
            frame = np.clip(frame, 0, 255).astype(np.uint8)
        if self._dc is not None:                      # collected in pinned memory, converted a batch at a time
            self._dc.bgr_buffer()[self._pending] = frame
            self._pending += 1
            if self._pending == self._dc.batch:
                self._flush()
            return
        y, u, v = _bgr_to_yuv(frame, False)
        if self.chroma != "444":
            h2, w2 = (self.h + 1) // 2 * 2, (self.w + 1) // 2 * 2
            def sub(p):
                p = np.pad(p, ((0, h2 - self.h), (0, w2 - self.w)), mode="edge").astype(np.uint16)
                return ((p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2] + 2) >> 2).astype(np.uint8)
            u, v = sub(u), sub(v)
        self._f.write(b"FRAME\n")
        self._f.write(y.tobytes())
        self._f.write(u.tobytes())
        self._f.write(v.tobytes())

    def planes_format(self):
        """None without a GPU; else what write_planes expects per frame (BT.601 studio range)"""
        if self._dc_args is None:
            return None
        return {"frame_bytes": self._dc_args[2], "subsample_420": self.chroma != "444", "full_range": False}

    def write_planes(self, recs):
        """frames already converted on the device: uint8 [n][frame_bytes]"""
        if self._pending:
            self._flush()
        for rec in recs:
            self._f.write(b"FRAME\n")
            self._f.write(rec)

    def record_layout(self, count):
        """For tools/rank_io.py: the header goes to disk, the file is grown to its final size of `count` frames, and the layout of its
        records comes back ({path, data_offset, prefix, frame_bytes}); the ranks then write their records by offset and this object
        writes nothing more.  None when frames were already written through it, or without the device colour conversion."""
        if self._dc_args is None or self._pending or self._f.tell() != self._header_bytes:
            return None
        self._f.flush()
        os.truncate(self._f.name, self._header_bytes + (6 + self._dc_args[2]) * int(count))
        return {"path": os.path.abspath(self._f.name), "data_offset": self._header_bytes, "prefix": b"FRAME\n", "frame_bytes": self._dc_args[2]}

    def release(self):
        if self._dc_obj is not None:
            self._flush()
        self._f.close()


IMAGE_EXTS = (".png", ".jpg", ".jpeg", ".bmp", ".webp", ".tif", ".tiff")      # tools/common_tools.py:8-9


class ImageVideo:
    """A still image as a one-frame source (main.py:353-356 read_image); PIL decodes, frames are BGR like cv2's."""

    def __init__(self, path):
        from PIL import Image

        self.frame = np.ascontiguousarray(np.array(Image.open(path).convert("RGB"))[:, :, ::-1])
        self._done = False

    def info(self):
        return {"W_ori": int(self.frame.shape[1]), "H_ori": int(self.frame.shape[0]), "fps": 1.0, "len": 1}

    def read(self):
        if self._done:
            return False, None
        self._done = True
        return True, self.frame.copy()

    def release(self):
        pass


class ImageWriter:
    def __init__(self, path):
        self.path = path

    def write(self, frame):
        from PIL import Image

        Image.fromarray(np.ascontiguousarray(frame[:, :, ::-1])).save(self.path)

    def release(self):
        pass


# ------------------------------------------------------------------------------------------------------------------------
# ffmpeg pipe / cv2 (only when the tool exists)
# ------------------------------------------------------------------------------------------------------------------------
def ffmpeg_path():
    """$VSR_FFMPEG, else an `ffmpeg` on PATH (the reference ships its own binary, tools/ffmpeg_cli.py:24-35: a missing blob)."""
    p = os.environ.get("VSR_FFMPEG") or shutil.which("ffmpeg")
    return p if p and os.path.exists(p) else None


def probe_stream(probe, path):
    """(width, height, fps, frames) of the first video stream AS DECODED FRAMES ARRIVE: `ffmpeg -i` applies the display-matrix
    rotation (phone portrait clips) by default -- and so does cv2.VideoCapture, the reference's reader, which reports the rotated
    size -- so a quarter-turn rotation swaps width and height here (round 2 took the stored size: every frame of such a clip was
    reshaped with the sides exchanged, silently).  The frame count falls back from the packet count to nb_frames to duration x rate
    when the container does not know it (`N/A`)."""
    import json

    out = subprocess.check_output([probe, "-v", "error", "-select_streams", "v:0", "-count_packets", "-show_entries",
                                   "stream=width,height,r_frame_rate,avg_frame_rate,nb_read_packets,nb_frames,duration:stream_tags=rotate:"
                                   "stream_side_data=rotation:format=duration", "-of", "json", path], text=True)
    doc = json.loads(out)
    st = doc["streams"][0]
    w, h = int(st["width"]), int(st["height"])
    num, den = (st.get("r_frame_rate") or st.get("avg_frame_rate") or "0/1").split("/")
    fps = float(num) / float(den or 1) if float(den or 1) else 0.0
    rot = 0.0
    for sd in st.get("side_data_list", []) or []:
        if "rotation" in sd:
            rot = float(sd["rotation"])
    if not rot and (st.get("tags") or {}).get("rotate") not in (None, ""):
        rot = float(st["tags"]["rotate"])
    if int(round(abs(rot))) % 180 == 90:
        w, h = h, w

    def as_int(v):
        try:
            return int(v)
        except (TypeError, ValueError):
            return None

    n = as_int(st.get("nb_read_packets"))
    if n is None:
        n = as_int(st.get("nb_frames"))
    if n is None:
        dur = st.get("duration") or (doc.get("format") or {}).get("duration")
        try:
            n = int(round(float(dur) * fps))
        except (TypeError, ValueError):
            raise RuntimeError(f"{path}: ffprobe reports neither a packet count, nor nb_frames, nor a duration") from None
    return w, h, fps, n


class FFmpegVideo:
    """`ffmpeg -i path -f rawvideo -pix_fmt bgr24 -` as a frame source; stream facts from `ffprobe` next to the binary."""

    def __init__(self, path):
        ff = ffmpeg_path()
        if ff is None:
            raise RuntimeError("no ffmpeg binary")
        probe = os.path.join(os.path.dirname(ff), "ffprobe")
        probe = probe if os.path.exists(probe) else (shutil.which("ffprobe") or "ffprobe")
        self.w, self.h, self.fps, self.n = probe_stream(probe, path)
        self._p = subprocess.Popen([ff, "-loglevel", "error", "-i", path, "-f", "rawvideo", "-pix_fmt", "bgr24", "-"],
                                   stdout=subprocess.PIPE, stdin=subprocess.DEVNULL, bufsize=1 << 24)

    def info(self):
        return {"W_ori": self.w, "H_ori": self.h, "fps": self.fps, "len": self.n}

    def read(self):
        need = self.w * self.h * 3
        buf = self._p.stdout.read(need)
        if len(buf) < need:
            return False, None
        return True, np.frombuffer(buf, dtype=np.uint8).reshape(self.h, self.w, 3).copy()

    def release(self):
        try:
            self._p.stdout.close()
            self._p.terminate()
            self._p.wait(timeout=5)
        except Exception:
            pass


class FFmpegVideoWriter:
    """Raw bgr24 frames into `ffmpeg ... -c:v libx264 -crf 18 -preset fast` -- the reference's writer (video_io.py:54-103).
    Unlike the reference's, it does not lose errors: a broken pipe (the encoder died) is remembered and raised by the next write /
    by release(), and release() raises when ffmpeg exits non-zero, with what it wrote to stderr (a run must not print
    'written' over a truncated file)."""

    def __init__(self, output_path, fps, size):
        import tempfile

        ff = ffmpeg_path()
        if ff is None:
            raise RuntimeError("no ffmpeg binary")
        w, h = size
        self.output_path = output_path
        cmd = [ff, "-y", "-f", "rawvideo", "-vcodec", "rawvideo", "-s", f"{w}x{h}", "-pix_fmt", "bgr24", "-r", str(fps), "-i", "-",
               "-c:v", "libx264", "-pix_fmt", "yuv420p", "-crf", "18", "-preset", "fast", "-loglevel", "error", output_path]
        self._log = tempfile.TemporaryFile()
        self._process = subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.DEVNULL, stderr=self._log)
        self._broken = None

    def _fail(self, why):
        self._log.seek(0)
        tail = self._log.read()[-2000:].decode("utf-8", "replace").strip()
        return RuntimeError(f"ffmpeg writing {self.output_path}: {why}" + (f"\n{tail}" if tail else ""))

    def write(self, frame):
        if self._broken is not None:
            raise self._fail(f"the encoder is gone ({self._broken})")
        if frame.dtype != np.uint8:
            frame = np.clip(frame, 0, 255).astype(np.uint8)
        try:
            self._process.stdin.write(frame.tobytes())
        except (BrokenPipeError, OSError) as e:
            self._broken = e
            raise self._fail(f"the encoder closed its input ({e})") from e

    def release(self):
        try:
            self._process.stdin.close()
        except (BrokenPipeError, OSError) as e:
            self._broken = self._broken or e
        try:
            self._process.wait(timeout=600)
        except subprocess.TimeoutExpired:
            self._process.terminate()
            self._process.wait(timeout=5)
            raise self._fail("no exit within 600 s after the last frame") from None
        if self._process.returncode != 0:
            raise self._fail(f"exit code {self._process.returncode}")
        if self._broken is not None:
            raise self._fail(f"the encoder closed its input early ({self._broken})")


class Cv2Video:
    """cv2.VideoCapture-backed source (sttn_auto_inpaint.py:168-180); needs opencv-python."""

    def __init__(self, path):
        import cv2

        self._cv2 = cv2
        self.cap = cv2.VideoCapture(path)

    def info(self):
        cv2 = self._cv2
        return {"W_ori": int(self.cap.get(cv2.CAP_PROP_FRAME_WIDTH) + 0.5),
                "H_ori": int(self.cap.get(cv2.CAP_PROP_FRAME_HEIGHT) + 0.5),
                "fps": self.cap.get(cv2.CAP_PROP_FPS),
                "len": int(self.cap.get(cv2.CAP_PROP_FRAME_COUNT) + 0.5)}

    def read(self):
        return self.cap.read()

    def release(self):
        self.cap.release()


# ------------------------------------------------------------------------------------------------------------------------
# threads: reading and writing overlap the GPU work
# ------------------------------------------------------------------------------------------------------------------------
class FramePrefetcher:
    """Background thread that keeps up to `buffer_size` decoded frames ready; same contract as the reference's
    FramePrefetcher (video_io.py:12-51: read / get-like info / stop / release)."""

    def __init__(self, video_cap, buffer_size=10):
        self.cap = video_cap
        self._buffer = queue.Queue(maxsize=buffer_size)
        self._stopped = False
        self._thread = threading.Thread(target=self._read_loop, daemon=True)
        self._thread.start()

    def _read_loop(self):
        while not self._stopped:
            try:
                ret, frame = self.cap.read()
            except Exception:
                ret, frame = False, None
            while not self._stopped:
                try:
                    self._buffer.put((ret, frame), timeout=0.2)
                    break
                except queue.Full:
                    continue
            if not ret:
                break

    def read(self):
        return self._buffer.get()

    def info(self):
        return self.cap.info()

    def stop(self):
        self._stopped = True
        try:
            while not self._buffer.empty():
                self._buffer.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5)

    def release(self):
        self.stop()
        self.cap.release()


class AsyncWriter:
    """Writer thread in front of a sink: `write` returns at once (a copy is queued, at most `buffer_size` frames deep), so colour
    conversion / pipe writes / disk IO run beside the GPU; `release` drains, closes the sink and re-raises a sink error."""

    def __init__(self, sink, buffer_size=32):
        self.sink = sink
        self._q = queue.Queue(maxsize=buffer_size)
        self._err = None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()

    def _run(self):
        while True:
            f = self._q.get()
            if f is None:
                return
            if self._err is None:
                try:
                    if isinstance(f, tuple):
                        self.sink.write_planes(f[1])
                    else:
                        self.sink.write(f)
                except Exception as e:      # keep draining so the producer never blocks on a dead sink
                    self._err = e

    def write(self, frame):
        if self._err is not None:
            raise self._err
        self._q.put(np.array(frame, copy=True))

    def planes_format(self):
        fmt = getattr(self.sink, "planes_format", None)
        return fmt() if fmt is not None else None

    def record_layout(self, count):
        """see Y4mWriter.record_layout (None for sinks without offsets: pipes, image files)"""
        fn = getattr(self.sink, "record_layout", None)
        if fn is None or self._err is not None or not self._q.empty():
            return None
        return fn(count)

    def write_planes(self, recs):
        """a batch of frames converted on the device (see Y4mWriter.write_planes); keeps its place in the frame order"""
        if self._err is not None:
            raise self._err
        self._q.put(("planes", np.array(recs, copy=True)))

    def release(self):
        if self._t is None:                 # already released (the plugin and run() both release, as in the reference)
            return
        self._q.put(None)
        self._t.join()
        self._t = None
        self.sink.release()
        if self._err is not None:
            raise self._err


# ------------------------------------------------------------------------------------------------------------------------
def open_video(video):
    """A fresh reader positioned at frame 0 (the reference re-opens the file for every pass over the video)."""
    if isinstance(video, ArrayVideo):
        return ArrayVideo(video.frames, video.fps)
    if hasattr(video, "read") and hasattr(video, "info"):
        return video
    path = os.fspath(video)
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return NpyVideo(path)
    if ext == ".y4m":
        return Y4mVideo(path)
    if ext in IMAGE_EXTS:
        return ImageVideo(path)
    if ffmpeg_path() is not None:
        return FFmpegVideo(path)
    try:
        return Cv2Video(path)
    except ImportError as e:
        raise RuntimeError(f"cannot read {path}: no ffmpeg binary (PATH / $VSR_FFMPEG) and no opencv-python; raw containers (*.y4m, "
                           f"*.npy) and in-memory ArrayVideo clips need neither") from e


def open_writer(path, fps, size, frames=None):
    """Sink for `path` (size = (W, H)); raises when nothing on this machine can write that container -- never a silent
    in-memory fallback for a file the caller asked for."""
    path = os.fspath(path)
    ext = os.path.splitext(path)[1].lower()
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    if ext == ".npy":
        return NpyWriter(path, fps, size, capacity=frames)
    if ext == ".y4m":
        return Y4mWriter(path, fps, size)
    if ext in IMAGE_EXTS:
        return ImageWriter(path)
    if ffmpeg_path() is not None:
        return FFmpegVideoWriter(path, fps, size)
    try:
        import cv2

        return cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, size)
    except ImportError as e:
        raise RuntimeError(f"cannot write {path}: no ffmpeg binary (PATH / $VSR_FFMPEG) and no opencv-python on this machine; "
                           f"choose a raw container (-o out.y4m or -o out.npy)") from e
