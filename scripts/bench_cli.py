#!/usr/bin/env python
"""End-to-end rate of the CLI path on a raw container: *.y4m file in -> SubtitleRemover.run() (--inpaint-mode sttn-auto, one
subtitle box) -> *.y4m file out, i.e. file read + colour conversion + strip upload + inpainting + download + colour conversion +
file write, the way `python -m vsr_amd.backend.main -i IN.y4m -o OUT.y4m -c ...` runs it.

    python scripts/bench_cli.py [--res 1080p] [--frames 300] [--color device|host]

--color host keeps the numpy colour conversion of round 2's first version (VSR_IO_COLOR=host); device is the default
(csrc/io_kernels.hip).  One JSON line; the read-only and write-only rates of the container are measured beside the run.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--res", default="1080p", choices=["720p", "1080p", "4k"])
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--color", default="device", choices=["device", "host"])
    ap.add_argument("--chroma-out", default="420", choices=["420", "444"])
    ap.add_argument("--resident", default="1", choices=["0", "1"], help="0: host-frame chunk loop (VSR_IO_RESIDENT=0)")
    ap.add_argument("--profile", action="store_true", help="cProfile around run(): where the start-up goes (stderr)")
    args = ap.parse_args()
    os.environ["VSR_IO_COLOR"] = args.color
    os.environ["VSR_IO_RESIDENT"] = args.resident

    import numpy as np
    import torch

    import vsr_amd  # noqa: F401
    from bench import RES
    from vsr_amd import synth
    from vsr_amd.backend.config import config
    from vsr_amd.backend.main import SubtitleRemover
    from vsr_amd.backend.tools import video_io
    from vsr_amd.backend.tools.constant import InpaintMode

    H, W, box = RES[args.res]
    N = args.frames
    tmp = tempfile.mkdtemp(prefix="vsr_cli_")
    src, dst = os.path.join(tmp, "in.y4m"), os.path.join(tmp, "out.y4m")
    base = synth.make_clip(10, H, W, box, seed=1)
    t0 = time.perf_counter()
    w = video_io.Y4mWriter(src, 30.0, (W, H), chroma="420")
    for i in range(N):
        w.write(np.roll(base[i % 10], (3 * (i // 10), 5 * (i // 10)), axis=(0, 1)))
    w.release()
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    r = video_io.Y4mVideo(src)
    n = 0
    while r.read()[0]:
        n += 1
    r.release()
    t_read = time.perf_counter() - t0
    assert n == N

    config.inpaintMode.value = InpaintMode.STTN_AUTO

    class _Out(video_io.Y4mWriter):
        def __init__(self, path, fps, size):
            super().__init__(path, fps, size, chroma=args.chroma_out)

    if args.chroma_out != "444":
        import vsr_amd.backend.main as m

        m.open_writer = lambda path, fps, size, frames=None: _Out(path, fps, size)
    sr = SubtitleRemover(src, model_path={"netG": synth.make_state_dict(0, "auto")})
    sr.sub_areas = [box]
    sr.video_out_path = dst
    sr.append_output = lambda *a: None
    torch.cuda.synchronize()
    prof = None
    if args.profile:
        import cProfile

        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    sr.run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats

        prof.disable()
        pstats.Stats(prof, stream=sys.stderr).sort_stats("cumulative").print_stats(45)
    out_bytes = os.path.getsize(dst)
    rd = video_io.Y4mVideo(dst)
    assert rd.info()["len"] == N
    ok, f0 = rd.read()
    rd.release()
    ri = video_io.Y4mVideo(src)
    _, i0 = ri.read()
    ri.release()
    ymin, ymax, xmin, xmax = box
    changed = float((f0[ymin:ymax, xmin:xmax] != i0[ymin:ymax, xmin:xmax]).mean())
    print(json.dumps({
        "metric": f"CLI end to end, {args.res} y4m in -> sttn-auto -> y4m out", "value": round(N / dt, 2), "unit": "frames/s",
        "frames": N, "seconds": round(dt, 3), "colour_conversion": args.color,
        "frames_resident_in_hbm": bool(args.resident == "1" and args.color == "device"), "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
        "container_read_only_fps": round(N / t_read, 1), "container_write_only_fps_420": round(N / t_write, 1),
        "in_bytes": os.path.getsize(src), "out_bytes": out_bytes, "out_chroma": args.chroma_out,
        "box_pixels_changed_frame0": round(changed, 3), "host_cpus": os.cpu_count()}))
    for p in (src, dst):
        os.remove(p)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()
