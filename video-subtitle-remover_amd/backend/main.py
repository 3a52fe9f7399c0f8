"""Command-line surface of the reference's backend/main.py for the accelerated path.

    python -m vsr_amd.backend.main -i IN -o OUT [-c YMIN YMAX XMIN XMAX]... [--inpaint-mode sttn-auto]

Same flags, enum names and call order as the reference (backend/main.py:473-488, tools/args_handler.py:6-30):
SubtitleRemover(path).sub_areas / .ab_sections / .video_out_path / .run() with the modes sttn-auto, sttn-det, lama and
propainter on the MI355X engines.  Frames are read and written through tools/video_io.py (raw containers, an ffmpeg pipe
when a binary exists; `-o OUT` is written or the run fails -- there is no silent in-memory sink for a file input); audio is
muxed with the reference's two ffmpeg commands when ffmpeg exists (main.py:418-460).  GUI transport, temp files and the
opencv mode stay with the reference (surface only, not accelerated).
"""
import os
import sys
import time
from pathlib import Path

import numpy as np

from .config import config
from .inpaint.sttn_auto_inpaint import STTNAutoInpaint
from .tools.args_handler import parse_args
from .tools.constant import InpaintMode
from .tools.inpaint_tools import batch_generator, create_mask, expand_frame_ranges
from .tools.subtitle_detect import SubtitleDetect
from .tools.video_io import IMAGE_EXTS, ArrayWriter, AsyncWriter, ffmpeg_path, open_video, open_writer


class SubtitleRemover:
    def __init__(self, vd_path, gui_mode=False, device="cuda:0", model_path=None, video_writer=None):
        self.sub_areas = []                      # [(ymin, ymax, xmin, xmax)] (args_handler.py:19 order)
        self.gui_mode = gui_mode
        self.video_path = vd_path
        self.device = device
        self.ab_sections = None
        info = open_video(vd_path).info() if not isinstance(vd_path, (str, os.PathLike)) else None
        if info is None:
            src = open_video(vd_path)
            info = src.info()
            src.release()
        self.frame_count = info["len"]
        self.fps = info["fps"]
        self.frame_height, self.frame_width = info["H_ori"], info["W_ori"]
        self.mask_size = (self.frame_height, self.frame_width)
        self.is_path = isinstance(vd_path, (str, os.PathLike))
        self.is_picture = False
        # in-memory input: in-memory sink.  File input: the sink is opened on first use for `video_out_path` (the caller may
        # still change it, main.py:486) and streams -- the reference writes through FFmpegVideoWriter as it goes (main.py:66-69)
        self._video_writer = video_writer if video_writer is not None else (None if self.is_path else ArrayWriter())
        if self.is_path:
            self.vd_name = Path(vd_path).stem
            ext = os.path.splitext(str(vd_path))[1].lower()
            self.is_picture = ext in IMAGE_EXTS
            out_ext = ext if ext in (".y4m", ".npy") else ".mp4"
            self.video_out_path = os.path.abspath(os.path.join(os.path.dirname(vd_path), f"{self.vd_name}_no_sub{out_ext}"))
            if self.is_picture:                                          # main.py:73-77
                self.video_out_path = os.path.abspath(os.path.join(os.path.dirname(vd_path), "no_sub", f"{self.vd_name}{ext}"))
        else:
            self.vd_name, self.video_out_path = "clip", None
        self.passed_through_single_frames = 0
        self.model_path = model_path or os.environ.get(
            "STTN_AUTO_MODEL_PATH", os.path.join(os.path.dirname(__file__), "models", "sttn-auto", "infer_model.pth"))
        self.progress_total = 0
        self.isFinished = False
        self.phase_seconds = {}

    @property
    def video_writer(self):
        if self._video_writer is None:
            sink = open_writer(self.video_out_path, self.fps, (self.frame_width, self.frame_height), frames=self.frame_count)
            self._video_writer = AsyncWriter(sink)
        return self._video_writer

    @video_writer.setter
    def video_writer(self, w):
        self._video_writer = w

    # hooks the plugin calls on its host object (main.py:109-151)
    def update_progress(self, tbar, increment):
        if tbar is not None:
            tbar.update(increment)

    def update_preview_with_comp(self, original, frame):
        pass

    def append_output(self, *args):
        print(*args)

    def sttn_auto_mode(self, tbar):
        """backend/main.py:247-258."""
        mask_area_coordinates = []
        for ymin, ymax, xmin, xmax in self.sub_areas:
            mask_area_coordinates.append((xmin, xmax, ymin, ymax))       # tuple order flips here (main.py:253-255)
        mask = create_mask(self.mask_size, mask_area_coordinates)
        sttn_video_inpaint = STTNAutoInpaint(self.device, self.model_path, self.video_path)
        sttn_video_inpaint(input_mask=mask, input_sub_remover=self, tbar=tbar)
        if sttn_video_inpaint.last_error is not None:
            # the plugin prints and returns like the reference's (sttn_auto_inpaint.py:329-331); the run must not report success
            # over a file that was not (completely) written
            raise sttn_video_inpaint.last_error

    @staticmethod
    def is_current_frame_no_start(frame_no, continuous_frame_no_list):          # main.py:90-97
        return any(start_no == frame_no for start_no, _ in continuous_frame_no_list)

    @staticmethod
    def find_frame_no_end(frame_no, continuous_frame_no_list):                   # main.py:100-107
        for start_no, end_no in continuous_frame_no_list:
            if start_no <= frame_no <= end_no:
                return end_no
        return -1

    def _distributed_rank(self):
        d = self._distributed()
        return d.get_rank() if d is not None else 0

    @staticmethod
    def _distributed():
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist
        return None

    def _run_items(self, tbar, items, process):
        """Drive the (pass-through frame | batch + mask) items of one video: on one GPU in order, on several dealt
        round-robin by tools/batch_parallel.py (rank 0 reads, detects and writes; peers only run `process`)."""
        from .tools.batch_parallel import run_batch_parallel

        def write(frame):
            self.video_writer.write(frame)
            self.update_progress(tbar, increment=1)

        run_batch_parallel(items, process, write, dist=self._distributed(), device=self.device if self._distributed() else "cpu")

    def _open_resident(self):
        """(ResidentClip, writer planes format) when this run can keep the decoded video in HBM (tools/resident.py: raw planar source
        and sink, one process, the clip fits), else None: the host-frame loop then runs as before."""
        from .tools.resident import ResidentClip

        if self._distributed() is not None or getattr(self, "gui_mode", False) or not self.is_path:
            return None
        reader = open_video(self.video_path)
        try:
            fmts = ResidentClip.formats(reader, self.video_writer)
            info = reader.info()
            if fmts is None or not ResidentClip.fits(info["len"], info["H_ori"], info["W_ori"]):
                return None
            t0 = time.time()
            clip = ResidentClip.load(reader, fmts[0], info["len"], info["H_ori"], info["W_ori"], self.device)
            self.phase_seconds["read + upload + YUV->BGR"] = time.time() - t0
            return clip, fmts[1]
        finally:
            reader.release()

    def _run_resident_jobs(self, jobs, plugin, clip, store=None):
        """the independent batches of a resident run: one after the other, or over VSR_BATCH_LANES plugin instances (tools/batch_lanes.py).
        store: a tools/resident.StreamingStore -- with one lane the frames in front of the next batch are handed to it as each batch
        is enqueued (behind an event on the compute stream), so that they are converted, downloaded and written under the batches
        that follow; jobs are slices of clip.frames in frame order."""
        import torch

        from .tools import batch_lanes

        if not hasattr(self, "_lane_cache"):
            self._lane_cache = {}
        plugins = batch_lanes.lane_plugins(plugin, batch_lanes.lanes_from_env(), self._lane_cache)
        if store is None or len(plugins) > 1 or os.environ.get("VSR_STREAM_STORE", "1") == "0":      # (0: everything is written after the last batch)
            batch_lanes.run_jobs(jobs, plugins, clip.frames.device)
            return
        dev = clip.frames.device
        frame_elems = clip.frames[0].numel()
        first = [(job[0].data_ptr() - clip.frames.data_ptr()) // frame_elems for job in jobs]     # first frame of every batch
        store.ready(first[0] if jobs else len(clip))
        for j, job in enumerate(jobs):
            plugin(*job)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(dev))
            store.ready(first[j + 1] if j + 1 < len(jobs) else len(clip), ev)

    @staticmethod
    def _clip_kw(clip):
        """the HBM-resident clip is handed on only when there is one: without it the calls keep the reference's signatures"""
        return {} if clip is None else {"clip": clip}

    def _timed(self, name, fn, *a, **kw):
        """phase timer of run(): detector pass / scene cuts / inpainting / writing, reported by main() and scripts/bench_e2e.py"""
        import torch

        t0 = time.time()
        r = fn(*a, **kw)
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.phase_seconds[name] = self.phase_seconds.get(name, 0.0) + time.time() - t0
        return r

    def propainter_mode(self, tbar, propainter_inpaint=None, text_detector=None, scene_div_points=None, single_frame_inpaint=None):
        """backend/main.py:159-245.  Intervals of frames with the same mask, cut at scene changes, are read whole and
        handed to the plugin in batch_generator batches of propainterMaxLoadNum.  The scene-change frame numbers come
        from tools/scene_detect.py (reference: scenedetect's ContentDetector, subtitle_detect.py:158-170) unless given; the
        detector is injected; isolated single frames go to LaMa as in the reference (main.py:217-224,231-237) when its weights
        are configured (`lama_inpaint`), otherwise they pass through and are counted in `passed_through_single_frames`."""
        if propainter_inpaint is None:
            from .inpaint.propainter_inpaint import PropainterInpaint

            model_dir = os.environ.get("PROPAINTER_MODEL_DIR", os.path.join(os.path.dirname(__file__), "models", "propainter"))
            propainter_inpaint = PropainterInpaint(self.device, model_dir, config.propainterMaxLoadNum.value)
        dist = self._distributed()
        if dist is not None and dist.get_rank() != 0:
            return self._run_items(tbar, (), propainter_inpaint)
        if single_frame_inpaint is None:
            lama = self.lama_inpaint
            single_frame_inpaint = lama.inpaint if lama is not None else None
        detector = SubtitleDetect(self.video_path, self.sub_areas, text_detector=text_detector)
        # one detector lane here unless VSR_DET_LANES says otherwise: with two, the detector pass of config 4 gains 0.45 s and the
        # propainter inpainting that follows loses 2-4 s (62.9-63.1 -> 65.0-67.5 s, same box, profiles/r04_pp_e2e_ab.log; the
        # sttn-det inpainting does not care) -- the second instance's streams stay alive beside the plugin's
        detector.det_lanes_default = 1
        # the clip stays in HBM only for a plugin that takes device tensors (an injected callable, or the cv2 plugin, gets host frames)
        resident = self._open_resident() if getattr(propainter_inpaint, "accepts_device_frames", False) else None
        clip = resident[0] if resident is not None else None
        sub_list = self._timed("detector pass", detector.find_subtitle_frame_no, sub_remover=self, **self._clip_kw(clip))
        if len(sub_list) == 0:
            self._run_items(tbar, (), propainter_inpaint)                  # releases the peers before failing
            raise Exception(f"No subtitle detected in {self.video_path}")
        ranges = detector.find_continuous_ranges_with_same_mask(sub_list)
        if scene_div_points is None:                 # main.py:165: self.sub_detector.get_scene_div_frame_no(self.video_path)
            dev = self.device
            scene_div_points = self._timed("scene cuts", detector.get_scene_div_frame_no, self.video_path,
                                           device=int(dev.split(":")[1]) if isinstance(dev, str) and ":" in dev else 0, **self._clip_kw(clip))
        ranges = detector.split_range_by_scene(ranges, list(scene_div_points))
        if resident is not None:
            # the same walk over frame numbers as items() below, on the clip in HBM: batches are slices inpainted in place
            import torch

            clip, wf = resident
            n, index = len(clip), 0
            jobs = []                              # (slice of the clip, mask): independent batches, run by tools/batch_lanes.py

            def inpaint_all():
                nonlocal index
                while index < n:
                    index += 1
                    if index not in sub_list or not self.is_current_frame_no_start(index, ranges):
                        continue
                    start_frame_no = index
                    end_frame_no = self.find_frame_no_end(index, ranges)
                    if end_frame_no == -1:
                        continue
                    index = min(end_frame_no, n)
                    nos = list(range(start_frame_no - 1, index))                          # 0-based indices of the interval
                    mask = create_mask(self.mask_size, sub_list[start_frame_no])
                    for batch in ([nos] if len(nos) == 1 else batch_generator(nos, config.propainterMaxLoadNum.value)):
                        if len(batch) == 1:
                            if single_frame_inpaint is None:
                                self.passed_through_single_frames += 1
                                if self.passed_through_single_frames == 1:
                                    self.append_output("warning: no LaMa weights configured (LAMA_MODEL_PATH): isolated subtitle frames pass through")
                            else:                                                          # main.py:217-224: LamaInpaint.inpaint on the whole frame
                                one = single_frame_inpaint(clip.frames[batch[0]].cpu().numpy(), mask)
                                clip.frames[batch[0]].copy_(torch.from_numpy(np.ascontiguousarray(one)))
                        else:
                            jobs.append((clip.frames[batch[0]:batch[-1] + 1], mask))
                self._run_resident_jobs(jobs, propainter_inpaint, clip, store)

            from .tools.resident import StreamingStore

            store = StreamingStore(clip, self.video_writer, wf, lambda: self.update_progress(tbar, increment=1))
            try:
                self._timed("inpainting", inpaint_all)
            except BaseException:
                store.abort()
                raise
            self._timed("BGR->YUV + download + write (what is left after the last batch)", store.finish)
            return
        reader = open_video(self.video_path)

        def items():
            index = 0
            while True:
                ok, frame = reader.read()
                if not ok:
                    break
                index += 1
                if index not in sub_list:
                    yield ("pass", frame)
                    continue
                if not self.is_current_frame_no_start(index, ranges):
                    continue
                start_frame_no = index
                end_frame_no = self.find_frame_no_end(index, ranges)
                if end_frame_no == -1:
                    continue
                temp_frames = [frame]
                while index < end_frame_no:
                    ok, frame = reader.read()
                    if not ok:
                        break
                    index += 1
                    temp_frames.append(frame)
                mask = create_mask(self.mask_size, sub_list[start_frame_no])
                for batch in ([temp_frames] if len(temp_frames) == 1 else batch_generator(temp_frames, config.propainterMaxLoadNum.value)):
                    if len(batch) == 1:
                        if single_frame_inpaint is None:
                            self.passed_through_single_frames += 1
                            if self.passed_through_single_frames == 1:
                                self.append_output("warning: no LaMa weights configured (LAMA_MODEL_PATH): isolated subtitle frames pass through")
                        yield ("pass", single_frame_inpaint(batch[0], mask) if single_frame_inpaint is not None else batch[0])
                    else:
                        yield ("work", batch, mask)

        try:
            self._run_items(tbar, items(), propainter_inpaint)
        finally:
            reader.release()

    def video_inpaint(self, tbar, model, text_detector=None):
        """backend/main.py:260-333 -- detector pass, interval construction, then `model(batch, mask)` per batch.
        Frame numbers are 1-based here exactly as in the reference."""
        dist = self._distributed()
        if dist is not None and dist.get_rank() != 0:
            return self._run_items(tbar, (), model)
        detector = SubtitleDetect(self.video_path, self.sub_areas, text_detector=text_detector)
        resident = self._open_resident() if getattr(model, "accepts_device_frames", False) else None
        sub_list = self._timed("detector pass", detector.find_subtitle_frame_no, sub_remover=self,
                               **self._clip_kw(resident[0] if resident is not None else None))
        if len(sub_list) == 0:
            self._run_items(tbar, (), model)
            raise Exception(f"No subtitle detected in {self.video_path}")
        ranges = detector.find_continuous_ranges_with_same_mask(sub_list)
        ranges = expand_frame_ranges(ranges, config.subtitleTimelineBackwardFrameCount.value,
                                     config.subtitleTimelineForwardFrameCount.value)
        ranges = detector.filter_and_merge_intervals(ranges, config.sttnReferenceLength.value)
        start_end = {s: min(e, self.frame_count) for s, e in ranges}

        def interval_mask(first, last):
            coords = []
            for no in range(first, last):                          # NB: the reference's range excludes `last` (:310)
                for area in sub_list.get(no, []):
                    xmin, xmax, ymin, ymax = area
                    if (ymax - ymin) - (xmax - xmin) > config.subtitleYXAxisDifferencePixel.value:
                        continue                                   # taller than wide: treated as a false detection
                    if area not in coords:
                        coords.append(area)
            return create_mask(self.mask_size, coords)

        if resident is not None:
            # the same walk over frame numbers as items() below, on the clip in HBM: a batch is a slice, inpainted in place
            clip, wf = resident
            n = len(clip)

            def inpaint_all():
                idx, jobs = 0, []
                while idx < n:
                    idx += 1
                    if idx not in start_end:
                        continue
                    first, last = idx, start_end[idx]
                    idx = min(last, n)                             # frames first .. idx are read (:300-305)
                    mask = interval_mask(first, last)
                    for batch in batch_generator(list(range(first - 1, idx)), config.getSttnMaxLoadNum()):
                        if len(batch) >= 1:
                            jobs.append((clip.frames[batch[0]:batch[-1] + 1], mask))
                self._run_resident_jobs(jobs, model, clip, store)

            from .tools.resident import StreamingStore

            store = StreamingStore(clip, self.video_writer, wf, lambda: self.update_progress(tbar, increment=1))
            try:
                self._timed("inpainting", inpaint_all)
            except BaseException:
                store.abort()
                raise
            self._timed("BGR->YUV + download + write (what is left after the last batch)", store.finish)
            return
        reader = open_video(self.video_path)

        def items():
            idx = 0
            while True:
                ok, frame = reader.read()
                if not ok:
                    break
                idx += 1
                if idx not in start_end:
                    yield ("pass", frame)
                    continue
                first, last = idx, start_end[idx]
                frames = [frame]
                for _ in range(last - first):
                    ok, frame = reader.read()
                    if not ok:
                        break
                    idx += 1
                    frames.append(frame)
                mask = interval_mask(first, last)
                for batch in batch_generator(frames, config.getSttnMaxLoadNum()):
                    if len(batch) >= 1:
                        yield ("work", batch, mask)

        try:
            self._timed("read + inpainting + write (host frames)", self._run_items, tbar, items(), model)
        finally:
            reader.release()

    @property
    def lama_inpaint(self):
        """main.py:462-466 (cached_property): LamaInpaint over $LAMA_MODEL_PATH or backend/models/big-lama/big-lama.pt; an injected
        object (attribute `_lama_inpaint`) wins; None when no weights can be found."""
        if getattr(self, "_lama_inpaint", None) is None:
            path = os.environ.get("LAMA_MODEL_PATH", os.path.join(os.path.dirname(__file__), "models", "big-lama", "big-lama.pt"))
            if not os.path.exists(path) and not os.path.isfile(os.path.join(os.path.dirname(path), "fs_manifest.csv")):
                return None                                  # neither the checkpoint nor its split parts
            from .inpaint.lama_inpaint import LamaInpaint

            self._lama_inpaint = LamaInpaint(self.device, path)
        return self._lama_inpaint

    @lama_inpaint.setter
    def lama_inpaint(self, plugin):
        self._lama_inpaint = plugin

    def picture_mode(self):
        """main.py:353-371: a single image -- detect, mask, LamaInpaint.inpaint on the whole frame, write."""
        src = open_video(self.video_path)
        ok, frame = src.read()
        src.release()
        if not ok:
            raise Exception(f"cannot read {self.video_path}")
        detector = SubtitleDetect(self.video_path, self.sub_areas, text_detector=self._default_detector())
        sub_list = detector.detect_subtitle(frame)
        if len(sub_list):
            lama = self.lama_inpaint
            if lama is None:
                raise Exception("inpaint mode: lama needs its weights (LAMA_MODEL_PATH)")
            frame = lama.inpaint(frame, create_mask(frame.shape[0:2], sub_list))
        self.video_writer.write(frame)

    def merge_audio_to_video(self):
        """main.py:418-460: copy the input's audio track into the written file; needs ffmpeg, skipped (with a note) without."""
        import shutil
        import subprocess
        import tempfile

        ff = ffmpeg_path()
        out = self.video_out_path
        if ff is None or not self.is_path or os.path.splitext(out)[1].lower() in (".y4m", ".npy"):
            return False
        silent = out + ".video_only" + os.path.splitext(out)[1]
        os.replace(out, silent)
        with tempfile.NamedTemporaryFile(suffix=".aac", delete=False) as tmp:
            audio = tmp.name
        try:
            subprocess.check_output([ff, "-y", "-i", str(self.video_path), "-acodec", "copy", "-vn", "-loglevel", "error", audio],
                                    stdin=subprocess.DEVNULL, timeout=600)
            subprocess.check_output([ff, "-y", "-i", silent, "-i", audio, "-vcodec", "copy", "-acodec", "copy", "-loglevel", "error", out],
                                    stdin=subprocess.DEVNULL, timeout=600)
            os.remove(silent)
            return True
        except Exception as e:
            self.append_output(f"audio not merged ({e}); keeping the silent video")
            shutil.move(silent, out)
            return False
        finally:
            if os.path.exists(audio):
                os.remove(audio)

    def _default_detector(self):
        """the injected detector if any, else the MI355X TextDetection configured through the environment (tools/ocr_det.py)"""
        det = getattr(self, "text_detector", None)
        if det is None:
            from .tools import ocr_det

            det = ocr_det.from_env(0 if not isinstance(self.device, str) or ":" not in self.device else int(self.device.split(":")[1]))
        return det

    def run(self):
        start_time = time.time()
        if len(self.sub_areas) == 0:
            self.sub_areas.append((0, self.frame_height, 0, self.frame_width))
        mode = config.inpaintMode.value
        if self.is_picture:
            self.picture_mode()
        elif mode == InpaintMode.STTN_AUTO:
            self.sttn_auto_mode(None)
        elif mode == InpaintMode.STTN_DET:
            from .inpaint.sttn_det_inpaint import STTNDetInpaint

            det_path = os.environ.get("STTN_DET_MODEL_PATH", os.path.join(os.path.dirname(__file__), "models", "sttn-det", "sttn.pth"))
            self.video_inpaint(None, STTNDetInpaint(self.device, det_path), text_detector=self._default_detector())
        elif mode == InpaintMode.LAMA:
            lama = self.lama_inpaint
            if lama is None:
                raise Exception("inpaint mode: lama needs its weights (LAMA_MODEL_PATH or backend/models/big-lama/big-lama.pt)")
            self.video_inpaint(None, lama, text_detector=self._default_detector())
        elif mode == InpaintMode.PROPAINTER:
            self.propainter_mode(None, propainter_inpaint=getattr(self, "propainter_inpaint", None),
                                 text_detector=self._default_detector(), scene_div_points=getattr(self, "scene_div_points", None))
        elif mode == InpaintMode.OPENCV:
            # main.py:383-384: cv2.inpaint (Telea, radius 3) on the CPU, frame by frame -- not part of the MI355X path (SURVEY 2.1:
            # surface only).  With opencv-python installed it is the reference's own two lines; without it args_handler has
            # already refused the mode.
            from .inpaint.opencv_inpaint import OpenCVInpaint

            self.video_inpaint(None, OpenCVInpaint(), text_detector=self._default_detector())
        else:
            raise Exception(f"inpaint mode: {mode} not implemented")     # main.py:386
        if self._video_writer is not None:
            self._video_writer.release()                                 # main.py:389
            if self.is_path and self._distributed_rank() == 0:
                self.merge_audio_to_video()
        self.isFinished = True
        self.progress_total = 100
        self.append_output(f"Finished in {round(time.time() - start_time)} s")


def main(argv=None):
    args = parse_args(argv)
    config.inpaintMode.value = args.inpaint_mode
    sr = SubtitleRemover(args.input)
    sr.sub_areas = [tuple(c) for c in args.subtitle_area_coords]
    if args.output is not None:
        sr.video_out_path = os.path.abspath(args.output)
    sr.run()
    sr.append_output(f"written: {sr.video_out_path}")


if __name__ == "__main__":
    main(sys.argv[1:])
