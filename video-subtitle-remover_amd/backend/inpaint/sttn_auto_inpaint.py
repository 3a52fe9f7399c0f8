"""sttn-auto plugins with the reference's signatures, running on the MI355X engine.

Mirrors backend/inpaint/sttn_auto_inpaint.py:
  STTNInpaint(device, model_path)                         :28-41
      __call__(input_frames, input_mask) -> frames        :43-97   (generic plugin contract)
      inpaint(frames) -> comp frames                      :122-164
      get_ref_index(neighbor_ids, length)                 :107-120
  STTNAutoInpaint(device, model_path, video_path, mask_path=None, clip_gap=None)   :182-197
      __call__(input_mask=None, input_sub_remover=None, tbar=None)                 :199-336

Arithmetic happens in libvsr_hip.so (there is no torch model and no CPU path here); this file
is the host loop: move a chunk of frames to HBM, one C call per chunk, move it back, write.
``model_path`` may also be an already loaded state_dict (the shipped checkpoints are absent
from the reference mount); ``video_path`` may be an ArrayVideo.
"""
import os

import numpy as np
import torch

from ..config import config
from ..tools.inpaint_tools import get_inpaint_area_by_mask, is_frame_number_in_ab_sections, threshold_mask
from ..tools.video_io import ArrayWriter, open_video
from ...engine import SttnEngine


def _load_state_dict(model_path):
    """the generator's state dict: a dict (tests), or a checkpoint file read as the reference reads it
    (`torch.load(path)['netG']`, sttn_auto_inpaint.py:34).  A DataParallel-saved checkpoint (`module.` on every key) is unwrapped;
    anything else that does not match the generator's keys and shapes is refused by the engine (strict load)."""
    if isinstance(model_path, dict):
        sd = model_path.get("netG", model_path)
    else:
        ck = torch.load(model_path, map_location="cpu")
        if not isinstance(ck, dict) or "netG" not in ck:
            raise KeyError(f"{model_path}: not an STTN checkpoint (no 'netG' entry; keys: {list(ck)[:5] if isinstance(ck, dict) else type(ck).__name__})")
        sd = ck["netG"]
    if len(sd) and all(k.startswith("module.") for k in sd):
        sd = {k[7:]: v for k, v in sd.items()}
    return sd


def _device_index(device):
    if isinstance(device, int):
        return device
    d = torch.device(device)
    if d.type != "cuda":
        raise RuntimeError(f"the MI355X path needs a HIP device, got '{device}' (no CPU fallback)")
    return 0 if d.index is None else d.index


class STTNInpaint:
    def __init__(self, device, model_path):
        self.device = device
        self.neighbor_stride = config.sttnNeighborStride.value
        self.ref_length = config.sttnReferenceLength.value
        self.engine = SttnEngine(_load_state_dict(model_path), "auto", device=_device_index(device),
                                 neighbor_stride=self.neighbor_stride, ref_length=self.ref_length)
        self.model_input_width, self.model_input_height = 640, 120

    def __call__(self, input_frames, input_mask):
        mask = threshold_mask(input_mask)
        H_ori, W_ori = mask.shape[:2]
        split_h = int(W_ori * 3 / 16)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, split_h, mask)
        if not inpaint_area or len(input_frames) == 0:
            return [f.copy() for f in input_frames]
        dev = self.engine.device
        frames = torch.from_numpy(np.ascontiguousarray(np.stack(input_frames))).to(dev, non_blocking=True)
        dmask = torch.from_numpy(np.ascontiguousarray(mask[:, :, 0])).to(dev, non_blocking=True)
        self.engine.auto_chunk(frames, dmask, inpaint_area, mask_host=mask[:, :, 0])
        out = frames.cpu().numpy()
        return [out[i] for i in range(out.shape[0])]

    @staticmethod
    def read_mask(path):
        from PIL import Image

        img = np.array(Image.open(path).convert("L"))
        return threshold_mask(img)

    def get_ref_index(self, neighbor_ids, length):
        return [i for i in range(0, length, self.ref_length) if i not in neighbor_ids]

    def inpaint(self, frames):
        """frames: list of 120x640x3 uint8 BGR -> list of comp frames (uint8 where decoded once, else float32), RGB."""
        dev = self.engine.device
        d = torch.from_numpy(np.ascontiguousarray(np.stack(frames))).to(dev)
        comp, counts = self.engine.inpaint(d)
        comp = comp.cpu().numpy()
        return [comp[i].astype(np.uint8) if counts[i] == 1 else comp[i] for i in range(len(frames))]


class _ResidentFrames:
    """the decoded frames of a chunk that stay in HBM while its strip rows are away (len() = frames actually read)"""

    def __init__(self, full, n):
        self.full, self.n = full, n

    def __len__(self):
        return self.n


class STTNAutoInpaint:
    def __init__(self, device, model_path, video_path, mask_path=None, clip_gap=None):
        self.sttn_inpaint = STTNInpaint(device, model_path)
        self.video_path = video_path
        self.mask_path = mask_path
        if isinstance(video_path, (str, os.PathLike)):
            self.video_out_path = os.path.join(
                os.path.dirname(os.path.abspath(video_path)),
                f"{os.path.basename(video_path).rsplit('.', 1)[0]}_no_sub.mp4")
        else:
            self.video_out_path = None
        self.clip_gap = config.getSttnMaxLoadNum() if clip_gap is None else clip_gap
        self.writer = None

    def _distributed(self):
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist
        return None

    def _run(self, dist, input_mask, input_sub_remover, tbar):
        """The chunk loop (:242-328) as a pipeline: read + upload of chunk i+1, inpainting of chunk i and download + write of
        chunk i-1 overlap (backend/tools/chunk_parallel.py; the reference is strictly serial).  With one process per GPU
        (`dist`) the chunks are dealt round-robin and rank 0 owns the frame source and sink.  Chunk boundaries are the
        reference's (clip_gap), so every frame sees exactly the temporal context it sees in the reference.  Only the rows
        between the first and the last strip travel to the GPUs (nothing else can change, :314-315); the decoded frames stay
        on the host and the returned rows are patched in before writing."""
        from ..tools import chunk_parallel as cp

        rank = dist.get_rank() if dist is not None else 0
        engine = self.sttn_inpaint.engine
        reader = open_video(self.video_path)
        frame_info = reader.info()
        W_ori, H_ori = frame_info["W_ori"], frame_info["H_ori"]
        ab_sections = input_sub_remover.ab_sections if input_sub_remover is not None else None
        writer = (input_sub_remover.video_writer if input_sub_remover is not None else ArrayWriter()) if rank == 0 else None
        self.writer = writer
        gui = input_sub_remover is not None and getattr(input_sub_remover, "gui_mode", False)
        mask = self.sttn_inpaint.read_mask(self.mask_path) if input_mask is None else threshold_mask(input_mask)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, int(W_ori * 3 / 16), mask)
        # the reference clamps clip_gap by free VRAM / (W*H*12 B) (:228-238); 288 GB never binds
        ranges = cp.chunk_ranges(frame_info["len"], self.clip_gap)
        y_lo = min((a[0] for a in inpaint_area), default=0)
        y_hi = max((a[1] for a in inpaint_area), default=0)
        local_areas = [(a[0] - y_lo, a[1] - y_lo, a[2], a[3]) for a in inpaint_area]
        dmask = torch.from_numpy(np.ascontiguousarray(mask[y_lo:y_hi, :, 0])).to(engine.device) if inpaint_area else None
        mask_rows_host = mask[y_lo:y_hi, :, 0] if inpaint_area else None     # the engine reads the rows that hold the mask off this copy
        kept = {}

        def tick(original, frame):
            if input_sub_remover is not None:
                if tbar is not None:
                    input_sub_remover.update_progress(tbar, increment=1)
                if original is not None:
                    input_sub_remover.update_preview_with_comp(original, frame)

        def load(i, out):
            s, e = ranges[i]
            frames = []
            for j in range(s, e):
                ok, image = reader.read()
                if not ok:                               # :259-261: a short read ends the chunk with the frames read so far
                    print(f"Warning: Failed to read frame {j}.")
                    out[j - s:e - s] = 0
                    break
                if not image.flags.owndata or not image.flags.writeable:
                    image = image.copy()                 # an in-memory source hands out views of the clip
                out[j - s] = image[y_lo:y_hi]
                frames.append(image)
            if not frames:
                print(f"Warning: No valid frames found in range {s + 1}-{e}. Skipping this segment.")
            kept[i] = frames

        def process(i, rows):
            s, e = ranges[i]
            n = len(kept[i]) if i in kept else e - s     # the owner of the frame source knows how many frames were read
            sel = [j - s for j in range(s, s + n) if is_frame_number_in_ab_sections(j, ab_sections)]
            if sel:
                engine.auto_chunk(rows[:n], dmask, local_areas, sel=None if len(sel) == n else sel, mask_host=mask_rows_host)

        def store(i, rows):
            for j, frame in enumerate(kept.pop(i)):
                original = frame.copy() if gui else None
                frame[y_lo:y_hi] = rows[j]
                writer.write(frame)
                tick(original, frame)

        local = self._rank_local_io(dist, rank, reader, writer, gui, inpaint_area, frame_info["len"])
        if local is not None:
            # every rank reads and writes its own chunks by offset (tools/rank_io.py): no rank-0 funnel, no collective on the data path
            try:
                self._run_rank_local(local, dist, rank, engine, ranges, mask, inpaint_area, (H_ori, W_ori), ab_sections, tick, frame_info["len"])
            finally:
                reader.release()
                if writer:
                    writer.release()
            return
        resident = self._resident_io(rank, reader, writer, gui, inpaint_area)
        if resident is not None:
            load, store = self._resident_load_store(resident, engine, reader, writer, ranges, kept, (H_ori, W_ori), (y_lo, y_hi), tick)
        try:
            if not inpaint_area:                         # nothing to inpaint anywhere: rank 0 copies the video through
                if rank == 0:
                    while True:
                        ok, image = reader.read()
                        if not ok:
                            break
                        writer.write(image)
                        tick(None, image)
                if dist is not None:
                    dist.barrier()
            else:
                # `io` only changes what rank 0 hands to load / store (pinned host rows or device rows); the exchange is the same
                cp.run_chunk_parallel(ranges, (y_hi - y_lo, W_ori, 3), load, process, store, dist=dist, device=engine.device,
                                      io="device" if resident is not None else "host")
        finally:
            getattr(store, "close", lambda: None)()      # the resident path's page-locking thread (tools/pinned.py)
            reader.release()
            if writer:
                writer.release()

    @staticmethod
    def _rank_local_io(dist, rank, reader, writer, gui, inpaint_area, total):
        """(source layout, sink layout, source plane format, sink plane format) when every rank can read and write its own chunks by
        offset: more than one rank (VSR_IO_PER_RANK=1 forces it for a single process too, =0 keeps the rank-0 funnel), a *.y4m source
        with fixed-size records and a *.y4m sink on rank 0, colour conversion on the GPU, no preview consumer.  Rank 0 decides and
        tells the others (the sink is its object); the sink is grown to its final size before anybody writes."""
        want = os.environ.get("VSR_IO_PER_RANK", "1" if dist is not None else "0")
        rl = getattr(reader, "record_layout", None)
        pf = getattr(reader, "planes_format", None)
        src = rl() if (rl is not None and want == "1" and not gui and inpaint_area) else None
        rf = pf() if (pf is not None and src is not None) else None
        # The decision is COLLECTIVE: a rank that cannot take the path (the file is not visible on its node, colour conversion
        # off there, VSR_IO_PER_RANK differing per rank) must not fall through to chunk_parallel's exchanges while rank 0 waits in
        # run_rank_local.  Every rank says whether IT can; only when all can does rank 0 size the sink and publish its layout.
        able = src is not None and rf is not None
        if dist is not None:
            flag = torch.tensor([1 if able else 0], dtype=torch.int64)
            if dist.get_backend() == "nccl":
                flag = flag.cuda()
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            able = bool(int(flag.item()))
        decision = [None]
        if rank == 0 and able:
            wl = getattr(writer, "record_layout", None)
            wpf = getattr(writer, "planes_format", None)
            wf = wpf() if wpf is not None else None
            if wl is not None and wf is not None and src["count"] == total:
                dst = wl(total)
                if dst is not None:
                    decision = [(dst, wf)]
        if dist is not None:
            dist.broadcast_object_list(decision, src=0)         # also orders "the sink has its final size" before every rank's first write
        if decision[0] is None or not able:
            return None
        return src, decision[0][0], rf, decision[0][1]

    def _run_rank_local(self, local, dist, rank, engine, ranges, mask, inpaint_area, size, ab_sections, tick, total):
        """the chunk loop of _run with per-rank file access: chunk i's stored planes -> pinned -> HBM -> BGR (vsr_io_yuv_to_bgr) ->
        vsr_sttn_auto_chunk in place on the whole frames -> planes (vsr_io_bgr_to_yuv) -> pinned -> the sink, at the records' offsets"""
        import ctypes as C

        from ..._lib import check, lib
        from ..tools import rank_io
        from ..tools.pinned import PinnedPool

        src, dst, rf, wf = local
        (H, W), dev = size, engine.device
        maxn = max((e - s for s, e in ranges), default=0)
        u8 = torch.uint8
        d_in = torch.empty((maxn, rf["frame_bytes"]), dtype=u8, device=dev)
        d_out = torch.empty((maxn, wf["frame_bytes"]), dtype=u8, device=dev)
        full = torch.empty((maxn, H, W, 3), dtype=u8, device=dev)
        dmask = torch.from_numpy(np.ascontiguousarray(mask[:, :, 0])).to(dev)
        mask_host = mask[:, :, 0]
        # staging is page-locked by a helper thread in the order of first use (tools/pinned.py); asked for with wait=True: the first
        # chunk waits for ITS buffer only
        order = [("in", 0), ("out", 0), ("in", 1), ("out", 1)]
        pool = PinnedPool([(maxn, rf["frame_bytes"] if k == "in" else wf["frame_bytes"]) for k, _ in order], device=dev)
        tensors = {}

        def alloc(kind, b, shape):
            t = pool.get(order.index((kind, b)), wait=True)
            if t is None:                                    # page-locking failed: ordinary memory
                t = torch.empty(shape, dtype=u8)
            tensors[t.numpy().ctypes.data] = t
            return t.numpy()

        ptr = lambda t: C.c_void_p(t.data_ptr())
        cur = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def work(i, inp, out):
            s, _ = ranges[i]
            k = inp.shape[0]
            ti, to = tensors[inp.ctypes.data][:k], tensors[out.ctypes.data][:k]
            with torch.cuda.device(dev):
                d_in[:k].copy_(ti, non_blocking=True)
                check(lib.vsr_io_yuv_to_bgr(ptr(d_in), rf["frame_bytes"], H, W, rf["cw"], rf["ch"], int(rf["full_range"]), ptr(full), k, cur()))
                sel = [j - s for j in range(s, s + k) if is_frame_number_in_ab_sections(j, ab_sections)]
                if sel:
                    engine.auto_chunk(full[:k], dmask, inpaint_area, sel=None if len(sel) == k else sel, mask_host=mask_host)
                check(lib.vsr_io_bgr_to_yuv(ptr(full), H, W, int(wf["subsample_420"]), int(wf["full_range"]), ptr(d_out), wf["frame_bytes"], k, cur()))
                to.copy_(d_out[:k], non_blocking=True)
                torch.cuda.current_stream(dev).synchronize()

        done = {"n": 0}

        def ticks(n):
            done["n"] += n
            if rank == 0:
                for _ in range(n):
                    tick(None, None)

        try:
            rank_io.run_rank_local(ranges, src, dst, work, dist=dist, alloc=alloc, tick=ticks)
        finally:
            pool.close()
        if rank == 0:
            for _ in range(total - done["n"]):               # the other ranks' frames: the progress bar ends at 100 %
                tick(None, None)

    @staticmethod
    def _resident_io(rank, reader, writer, gui, inpaint_area):
        """(reader format, writer format) when the frames of this run can stay in HBM from the stored planes to the stored planes:
        a raw planar source and sink whose colour conversion runs on the GPU (*.y4m, tools/video_io.py), no preview consumer.
        VSR_IO_RESIDENT=0 keeps the host-frame loop."""
        if rank != 0 or gui or not inpaint_area or os.environ.get("VSR_IO_RESIDENT", "1") == "0":
            return None
        rf = getattr(reader, "planes_format", None)
        wf = getattr(writer, "planes_format", None)
        rf, wf = (rf() if rf is not None else None), (wf() if wf is not None else None)
        return (rf, wf) if rf is not None and wf is not None else None

    @staticmethod
    def _resident_load_store(fmt, engine, reader, writer, ranges, kept, size, rows, tick):
        """load / store of the chunk loop with the decoded frames resident in HBM (io="device" of tools/chunk_parallel.py): a chunk's
        stored planes go to pinned memory and up as they are (half the bytes of the BGR frames), vsr_io_yuv_to_bgr converts, the strip
        rows are copied out for the owner of the chunk; on the way back the rows are patched in, vsr_io_bgr_to_yuv converts and the
        planes come down into pinned memory for the writer thread.  The host touches no pixel: the reference's loop
        (sttn_auto_inpaint.py:254-262 read, :314-328 write) is cv2 / libswscale work on the CPU, 47 + 26 ms per 1080p frame in numpy."""
        import ctypes as C

        from ..._lib import check, lib

        (rf, wf), (H, W), (y_lo, y_hi) = fmt, size, rows
        dev = engine.device
        maxn = max((e - s for s, e in ranges), default=0)
        u8 = torch.uint8
        # pinned staging, page-locked by a helper thread in the order of first use (tools/pinned.py; the first buffer index used on
        # either side is 1): a transfer whose buffer is not there yet goes through pageable memory once
        from ..tools.pinned import PinnedPool

        pool = PinnedPool([(maxn, rf["frame_bytes"]), (maxn, wf["frame_bytes"]), (maxn, rf["frame_bytes"]), (maxn, wf["frame_bytes"])], device=dev)
        slot = {("in", 1): 0, ("out", 1): 1, ("in", 0): 2, ("out", 0): 3}
        pageable = {}

        def host_buf(kind, b):
            t = pool.get(slot[(kind, b)])
            if t is not None:
                return t, True
            if kind not in pageable:
                pageable[kind] = torch.empty((maxn, (rf if kind == "in" else wf)["frame_bytes"]), dtype=u8)
            return pageable[kind], False

        ev_in = [None, None]
        d_in = torch.empty((maxn, rf["frame_bytes"]), dtype=u8, device=dev)
        d_out = torch.empty((maxn, wf["frame_bytes"]), dtype=u8, device=dev)
        free, turn = [], {"in": 0, "out": 0}
        ptr = lambda t: C.c_void_p(t.data_ptr())
        cur = lambda: C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def load(i, out):                                # rank 0, called on the io stream of the chunk loop
            s, e = ranges[i]
            b = turn["in"] = turn["in"] ^ 1
            if ev_in[b] is not None:
                ev_in[b].synchronize()                   # the upload that last used this pinned buffer is done
                ev_in[b] = None
            hb, pinned = host_buf("in", b)
            k = reader.read_planes_into(hb.numpy()[: e - s])
            if k < e - s:
                print(f"Warning: Failed to read frame {s + k}.")                 # :259-261: the chunk ends with the frames read so far
            full = free.pop() if free else torch.empty((maxn, H, W, 3), dtype=u8, device=dev)
            if k:
                d_in[:k].copy_(hb[:k], non_blocking=pinned)    # (a pageable source is copied before the call returns)
                if pinned:
                    ev_in[b] = torch.cuda.Event()
                    ev_in[b].record(torch.cuda.current_stream(dev))
                check(lib.vsr_io_yuv_to_bgr(ptr(d_in), rf["frame_bytes"], H, W, rf["cw"], rf["ch"], int(rf["full_range"]), ptr(full), k, cur()))
                out[:k].copy_(full[:k, y_lo:y_hi])
            else:
                print(f"Warning: No valid frames found in range {s + 1}-{e}. Skipping this segment.")
            out[k:].zero_()
            kept[i] = _ResidentFrames(full, k)

        def store(i, rows_dev):                          # rank 0, io stream, chunk order
            kf = kept.pop(i)
            k = len(kf)
            if k:
                kf.full[:k, y_lo:y_hi].copy_(rows_dev[:k])
                b = turn["out"] = turn["out"] ^ 1
                check(lib.vsr_io_bgr_to_yuv(ptr(kf.full), H, W, int(wf["subsample_420"]), int(wf["full_range"]), ptr(d_out), wf["frame_bytes"], k, cur()))
                hb, pinned = host_buf("out", b)
                hb[:k].copy_(d_out[:k], non_blocking=pinned)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                ev.synchronize()
                writer.write_planes(hb.numpy()[:k])              # the writer thread takes its own copy
                for _ in range(k):
                    tick(None, None)
            free.append(kf.full)

        store.close = pool.close                         # called by _run when the loop is over
        return load, store

    def __call__(self, input_mask=None, input_sub_remover=None, tbar=None):
        """The reference swallows every error here and prints it (:329-331); its caller then reports success over a missing or
        truncated file.  The print is kept, the error is kept too: `last_error` holds it and SubtitleRemover.run() raises it, so
        that `-o OUT` is written or the run fails.  The writer is opened before the loop, outside the swallowed region
        of old: a missing sink is an error of the call, not of some chunk."""
        self.last_error = None
        try:
            self._run(self._distributed(), input_mask, input_sub_remover, tbar)
            # (the reference collects garbage here, :326, to return its per-chunk frame lists; this loop holds none -- 40 ms saved)
        except Exception as e:
            self.last_error = e
            print(f"Error during video processing: {str(e)}")
