"""Scene-cut pass on one MI355X: 1080p frames resident in HBM -> per-pair HSV difference sums.
Prints frames/s of the device part (HIP events on the launch stream) and of the whole pass from host frames."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_amd  # noqa: F401,E402
from vsr_amd.backend.tools import scene_detect  # noqa: E402
from vsr_amd.backend.tools.video_io import ArrayVideo  # noqa: E402


def main():
    H, W, B, reps = 1080, 1920, 64, 20
    rng = np.random.default_rng(0)
    clip = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    det = scene_detect.ContentDetector(device=0, batch_frames=B)
    d = torch.from_numpy(clip).to(det.device)
    det.device_batch(d, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        det.device_batch(d, True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    f = scene_detect.compute_downscale_factor(W)
    w, h = round(W / f), round(H / f)
    # bytes the pass has to move per frame: the 2x2 source taps of every output pixel at cache-line granularity is hardware detail;
    # algorithmic = taps read (4 x 3 B per output pixel) + small frame written/read by HSV + HSV written + read twice by the sums
    alg = (w * h * 3) * (4 + 1 + 1 + 1 + 2)
    t0 = time.perf_counter()
    div = scene_detect.get_scene_div_frame_no(ArrayVideo(clip, fps=25.0), detector=det)
    dt = time.perf_counter() - t0
    print(json.dumps({"workload": f"{B} frames {W}x{H} -> {w}x{h} HSV difference sums", "device_ms_per_batch": round(ms, 4),
                      "device_frames_per_s": round(B / ms * 1e3, 1), "algorithmic_bytes_per_frame": alg,
                      "algorithmic_GBps": round(alg * B / ms / 1e6, 2), "source_GBps_if_whole_frames_streamed": round(H * W * 3 * B / ms / 1e6, 1),
                      "host_to_cuts_frames_per_s": round(B / dt, 1), "cuts": div}))


if __name__ == "__main__":
    main()
