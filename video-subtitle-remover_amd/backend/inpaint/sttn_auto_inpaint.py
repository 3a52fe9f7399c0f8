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
import gc
import os

import numpy as np
import torch

from ..config import config
from ..tools.inpaint_tools import get_inpaint_area_by_mask, is_frame_number_in_ab_sections, threshold_mask
from ..tools.video_io import ArrayWriter, open_video
from ...engine import SttnEngine


def _load_state_dict(model_path):
    if isinstance(model_path, dict):
        return model_path.get("netG", model_path)
    return torch.load(model_path, map_location="cpu")["netG"]      # sttn_auto_inpaint.py:34


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
        self.engine.auto_chunk(frames, dmask, inpaint_area)
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

    def _call_chunk_parallel(self, dist, input_mask, input_sub_remover, tbar):
        """One process per GPU: the chunks are dealt round-robin, rank 0 owns the frame source and sink
        (backend/tools/chunk_parallel.py).  Chunk boundaries are the reference's (clip_gap), so every frame
        sees exactly the temporal context it sees in the single-GPU run.  Only the rows between the first and the
        last strip travel to the GPUs (nothing else can change, :314-315); rank 0 keeps the decoded frames and
        patches the returned rows in before writing."""
        from ..tools import chunk_parallel as cp

        rank = dist.get_rank()
        engine = self.sttn_inpaint.engine
        reader = open_video(self.video_path)
        frame_info = reader.info()
        W_ori, H_ori = frame_info["W_ori"], frame_info["H_ori"]
        ab_sections = input_sub_remover.ab_sections if input_sub_remover is not None else None
        writer = (input_sub_remover.video_writer if input_sub_remover is not None else ArrayWriter()) if rank == 0 else None
        self.writer = writer
        mask = self.sttn_inpaint.read_mask(self.mask_path) if input_mask is None else threshold_mask(input_mask)
        inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, int(W_ori * 3 / 16), mask)
        ranges = cp.chunk_ranges(frame_info["len"], self.clip_gap)
        y_lo = min((a[0] for a in inpaint_area), default=0)
        y_hi = max((a[1] for a in inpaint_area), default=0)
        local_areas = [(a[0] - y_lo, a[1] - y_lo, a[2], a[3]) for a in inpaint_area]
        dmask = torch.from_numpy(np.ascontiguousarray(mask[y_lo:y_hi, :, 0])).to(engine.device) if inpaint_area else None
        kept = {}

        def tick():
            if input_sub_remover is not None and tbar is not None:
                input_sub_remover.update_progress(tbar, increment=1)

        def load(i, out):
            s, e = ranges[i]
            frames = []
            for j in range(s, e):
                ok, image = reader.read()
                if not ok:
                    raise RuntimeError(f"Failed to read frame {j}.")
                if not image.flags.owndata or not image.flags.writeable:
                    image = image.copy()                 # an in-memory source hands out views of the clip
                out[j - s] = image[y_lo:y_hi]
                frames.append(image)
            kept[i] = frames

        def process(i, rows):
            s, e = ranges[i]
            sel = [j - s for j in range(s, e) if is_frame_number_in_ab_sections(j, ab_sections)]
            if sel:
                engine.auto_chunk(rows, dmask, local_areas, sel=None if len(sel) == e - s else sel)

        def store(i, rows):
            for j, frame in enumerate(kept.pop(i)):
                frame[y_lo:y_hi] = rows[j]
                writer.write(frame)
                tick()

        try:
            if not inpaint_area:                         # nothing to inpaint anywhere: rank 0 copies the video through
                if rank == 0:
                    while True:
                        ok, image = reader.read()
                        if not ok:
                            break
                        writer.write(image)
                        tick()
                dist.barrier()
            else:
                cp.run_chunk_parallel(ranges, (y_hi - y_lo, W_ori, 3), load, process, store, dist=dist, device=engine.device)
        finally:
            reader.release()
            if writer:
                writer.release()

    def __call__(self, input_mask=None, input_sub_remover=None, tbar=None):
        reader = None
        writer = None
        try:
            dist = self._distributed()
            if dist is not None:
                return self._call_chunk_parallel(dist, input_mask, input_sub_remover, tbar)
            reader = open_video(self.video_path)
            frame_info = reader.info()
            if input_sub_remover is not None:
                ab_sections = input_sub_remover.ab_sections
                writer = input_sub_remover.video_writer
            else:
                ab_sections = None
                writer = ArrayWriter()
            self.writer = writer
            W_ori, H_ori = frame_info["W_ori"], frame_info["H_ori"]
            split_h = int(W_ori * 3 / 16)
            mask = self.sttn_inpaint.read_mask(self.mask_path) if input_mask is None else threshold_mask(input_mask)
            inpaint_area = get_inpaint_area_by_mask(W_ori, H_ori, split_h, mask)
            # the reference clamps clip_gap by free VRAM / (W*H*12 B) (:228-238); 288 GB never binds
            clip_gap = self.clip_gap
            engine = self.sttn_inpaint.engine
            dmask = torch.from_numpy(np.ascontiguousarray(mask[:, :, 0])).to(engine.device)
            torch.cuda.synchronize(engine.device)
            total = frame_info["len"]
            rec_time = total // clip_gap if total % clip_gap == 0 else total // clip_gap + 1
            # Three-stage pipeline over the chunks (reference: read -> inpaint -> write, strictly serial,
            # sttn_auto_inpaint.py:242-328): pinned host buffers, H2D / compute / D2H on separate HIP streams,
            # so that chunk i computes while chunk i+1 is read + uploaded and chunk i-1 is downloaded + written.
            dev = engine.device
            s_h2d, s_cmp, s_d2h = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
            shape = (clip_gap, H_ori, W_ori, 3)
            pin_in = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
            pin_out = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
            dbuf = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(2)]
            ev_up = [torch.cuda.Event() for _ in range(2)]
            ev_cmp = [torch.cuda.Event() for _ in range(2)]
            ev_down = [torch.cuda.Event() for _ in range(2)]
            pending = None                                   # (slot, n_frames, originals) waiting to be written

            def flush(p):
                slot, n, originals = p
                ev_down[slot].synchronize()
                out = pin_out[slot].numpy()
                for j in range(n):
                    writer.write(out[j])
                    if input_sub_remover is not None:
                        if tbar is not None:
                            input_sub_remover.update_progress(tbar, increment=1)
                        if originals is not None:
                            input_sub_remover.update_preview_with_comp(originals[j], out[j])

            for i in range(rec_time):
                start_f, end_f = i * clip_gap, min((i + 1) * clip_gap, total)
                slot = i & 1
                ev_down[slot].synchronize()                  # pin_out[slot] / dbuf[slot] of chunk i-2 are free again
                host = pin_in[slot].numpy()
                n, sel = 0, []
                for j in range(start_f, end_f):
                    ok, image = reader.read()
                    if not ok:
                        print(f"Warning: Failed to read frame {j}.")
                        break
                    host[n] = image
                    if is_frame_number_in_ab_sections(j, ab_sections):
                        sel.append(n)
                    n += 1
                if n == 0:
                    print(f"Warning: No valid frames found in range {start_f + 1}-{end_f}. Skipping this segment.")
                    continue
                gui = input_sub_remover is not None and getattr(input_sub_remover, "gui_mode", False)
                originals = host[:n].copy() if gui else None
                with torch.cuda.stream(s_h2d):
                    dbuf[slot][:n].copy_(pin_in[slot][:n], non_blocking=True)
                    ev_up[slot].record(s_h2d)
                with torch.cuda.stream(s_cmp):
                    s_cmp.wait_event(ev_up[slot])
                    if inpaint_area and sel:
                        engine.auto_chunk(dbuf[slot][:n], dmask, inpaint_area, sel=None if len(sel) == n else sel)
                    ev_cmp[slot].record(s_cmp)
                with torch.cuda.stream(s_d2h):
                    s_d2h.wait_event(ev_cmp[slot])
                    pin_out[slot][:n].copy_(dbuf[slot][:n], non_blocking=True)
                    ev_down[slot].record(s_d2h)
                if pending is not None:
                    flush(pending)                           # chunk i-1 is written while chunk i computes
                pending = (slot, n, originals)
            if pending is not None:
                flush(pending)
            gc.collect()
        except Exception as e:          # the reference swallows every error here (:329-331)
            print(f"Error during video processing: {str(e)}")
        finally:
            if reader:
                reader.release()
            if writer:
                writer.release()
