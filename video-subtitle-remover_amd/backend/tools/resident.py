"""The decoded video resident in HBM for the detector-driven modes (sttn-det, lama, propainter).

The reference walks the file up to three times on the CPU -- the detector pass (tools/subtitle_detect.py:84-132), the scene-cut pass of
the propainter mode (:158-170), the inpainting pass (main.py:159-245, :260-333) -- decoding every frame each time, and every plugin
call copies its batch to the GPU and back.  With 288 GB of HBM a whole clip fits on the device (a 1080p frame is 6.2 MB as BGR:
1200 frames = 7.5 GB): the stored planes go up ONCE (half the bytes of the BGR frames), vsr_io_yuv_to_bgr converts them there, the
detector samples its frames from that tensor, the scene-cut kernels read it, the plugins work in place on slices of it, and
vsr_io_bgr_to_yuv + one download per batch feed the writer.  The host touches no pixel (round 2 had this for
sttn-auto only).

Eligible: a raw planar source and sink whose colour conversion runs on the GPU (*.y4m, tools/video_io.py), one process, and a clip
that fits VSR_RESIDENT_GB (default 64) as BGR.  Anything else keeps the host-frame loop; VSR_IO_RESIDENT=0 forces it.
"""
import ctypes as C
import os

import torch

from ..._lib import check, lib


class ResidentClip:
    BATCH = 32

    def __init__(self, frames, fmt_in):
        self.frames = frames                     # uint8 [N,H,W,3] BGR on the device
        self.fmt_in = fmt_in

    @staticmethod
    def formats(reader, writer):
        """(reader planes format, writer planes format) or None"""
        if os.environ.get("VSR_IO_RESIDENT", "1") == "0":
            return None
        rf, wf = getattr(reader, "planes_format", None), getattr(writer, "planes_format", None)
        rf, wf = (rf() if rf is not None else None), (wf() if wf is not None else None)
        return (rf, wf) if rf is not None and wf is not None else None

    @staticmethod
    def fits(n, H, W):
        return n * H * W * 3 <= float(os.environ.get("VSR_RESIDENT_GB", "64")) * 2 ** 30

    @classmethod
    def load(cls, reader, rf, n, H, W, device):
        """read the stored planes of the whole clip, BATCH frames at a time through two pinned buffers, convert on the device"""
        dev = torch.device(device)
        frames = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
        pins = [torch.empty((cls.BATCH, rf["frame_bytes"]), dtype=torch.uint8).pin_memory() for _ in range(2)]
        dplanes = [torch.empty((cls.BATCH, rf["frame_bytes"]), dtype=torch.uint8, device=dev) for _ in range(2)]
        events = [None, None]
        got, b = 0, 0
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev)
            while got < n:
                if events[b] is not None:
                    events[b].synchronize()          # the upload + conversion that last used this pair of buffers is done
                want = min(cls.BATCH, n - got)
                k = reader.read_planes_into(pins[b].numpy()[:want])
                if k:
                    dplanes[b][:k].copy_(pins[b][:k], non_blocking=True)
                    check(lib.vsr_io_yuv_to_bgr(C.c_void_p(dplanes[b].data_ptr()), rf["frame_bytes"], H, W, rf["cw"], rf["ch"], int(rf["full_range"]),
                                                C.c_void_p(frames[got:].data_ptr()), k, C.c_void_p(stream.cuda_stream)))
                    events[b] = torch.cuda.Event()
                    events[b].record(stream)
                got += k
                b ^= 1
                if k < want:                         # a short file: the clip ends with the frames read (reference :259-261)
                    frames = frames[:got]
                    break
            torch.cuda.synchronize(dev)
        return cls(frames, rf)

    def __len__(self):
        return int(self.frames.shape[0])

    def store(self, writer, wf, lo, hi, tick=None):
        """frames [lo, hi) -> the writer's planes, converted on the device, in order"""
        dev = self.frames.device
        n, H, W, _ = self.frames.shape
        if getattr(self, "_store_bufs", None) is None or self._store_bufs[0] != wf["frame_bytes"]:      # (kept: store() may be called range by range)
            self._store_bufs = (wf["frame_bytes"],
                                [torch.empty((self.BATCH, wf["frame_bytes"]), dtype=torch.uint8).pin_memory() for _ in range(2)],
                                [torch.empty((self.BATCH, wf["frame_bytes"]), dtype=torch.uint8, device=dev) for _ in range(2)])
        _, pins, dout = self._store_bufs
        b = 0
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev)
            for s in range(lo, hi, self.BATCH):
                k = min(self.BATCH, hi - s)
                check(lib.vsr_io_bgr_to_yuv(C.c_void_p(self.frames[s:].data_ptr()), H, W, int(wf["subsample_420"]), int(wf["full_range"]),
                                            C.c_void_p(dout[b].data_ptr()), wf["frame_bytes"], k, C.c_void_p(stream.cuda_stream)))
                pins[b][:k].copy_(dout[b][:k], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(stream)
                ev.synchronize()
                writer.write_planes(pins[b].numpy()[:k])         # the writer thread takes its own copy
                if tick is not None:
                    for _ in range(k):
                        tick()
                b ^= 1


class StreamingStore:
    """Writes the frames of a ResidentClip in order AS THEY BECOME FINAL, on its own thread and stream.

    The batches of a detector-driven run are inpainted in frame order and nothing after batch j touches a frame in front of batch
    j + 1, so the conversion, download and file write of what is finished run under the inpainting of what is not -- the
    reference writes every batch as soon as its plugin call returns (main.py:239-245, :326-332); round 3's resident path wrote the
    whole clip after the last batch (1.0 s of a 12.3 s run at 1080p x 1200, profiles/r03_e2e_configs_3_4.log).

        st = StreamingStore(clip, writer, wf, tick)
        st.ready(hi, event)      frames [.., hi) are final once `event` (a torch.cuda.Event, or None = now) has completed
        st.finish()              everything up to len(clip); joins the thread, re-raises its error"""

    def __init__(self, clip, writer, wf, tick=None):
        import queue
        import threading

        self.clip, self.writer, self.wf, self.tick = clip, writer, wf, tick
        self._q = queue.Queue()
        self._error = None
        self._thread = threading.Thread(target=self._run, name="vsr-streaming-store", daemon=True)
        self._thread.start()

    def _run(self):
        dev = self.clip.frames.device
        lo = 0
        with torch.cuda.device(dev), torch.cuda.stream(torch.cuda.Stream(dev)):
            while True:
                item = self._q.get()
                if item is None:
                    return
                if self._error is not None:       # after a failed write: drain the queue until the sentinel, write nothing more
                    continue
                try:
                    hi, event = item
                    if event is not None:
                        event.synchronize()
                    hi = min(int(hi), len(self.clip))
                    if hi > lo:
                        self.clip.store(self.writer, self.wf, lo, hi, self.tick)
                        lo = hi
                except BaseException as e:        # noqa: BLE001 -- re-raised by ready() / finish() in the caller's thread
                    self._error = e

    def ready(self, hi, event=None):
        if self._error is not None:               # fail fast: do not inpaint the rest of the clip for a file that cannot be written
            raise self._error
        self._q.put((hi, event))

    def abort(self):
        """the run failed: stop writing, release the thread"""
        self._q.put(None)
        self._thread.join()

    def finish(self):
        self._q.put((len(self.clip), None))
        self._q.put(None)
        self._thread.join()
        if self._error is not None:
            raise self._error
