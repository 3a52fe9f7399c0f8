"""BASELINE-size oracle runs for tests/test_gpu_zbaseline.py (test infrastructure).

The CPU oracle needs 30 s .. 4 min for one BASELINE.json-size unit (a 50-frame stride-5 / ref-10 sttn-auto chunk, a 47-frame
sttn-det batch, a 20-frame 1920x360 propainter strip with 20 RAFT iterations).  So that the GPU suite does not wait for them
one after the other, every job runs in its own subprocess (`python -m tests._baseline_oracle JOB OUT.npy`), all started
when pytest has collected a test that needs them (tests/conftest.py) and joined by the test that compares against the result --
the GPU box has 256 host threads, the jobs share them.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

JOBS = {
    # BASELINE.json configs[1] / metric: sttn-auto, 50-frame chunks, stride 5, references every 10
    "auto_720p": dict(kind="auto", H=720, W=1280, box=(620, 700, 192, 1088), L=50, seed=4),
    "auto_1080p": dict(kind="auto", H=1080, W=1920, box=(950, 1070, 288, 1632), L=50, seed=3),
    # configs[4]: 4K strip (3840x720); the GPU side runs the fp16-operand mode against this fp32 oracle
    "auto_4k": dict(kind="auto", H=2160, W=3840, box=(1900, 2140, 576, 3264), L=50, seed=5),
    # configs[2]: sttn-det, batch_generator(1200, 50) -> 47-frame batches, strip 1920x533
    "det_1080p": dict(kind="det", H=1080, W=1920, box=(950, 1070, 288, 1632), L=47, seed=6),
    # configs[3]: propainter on the 1920x360 strip, 20 RAFT iterations
    "pp_1080p": dict(kind="pp", H=1080, W=1920, box=(950, 1070, 288, 1632), L=20, seed=7),
    # configs[3] at the batch sizes batch_generator(1200, 70) really hands the plugin: 17 x 68 frames + a 44-frame tail
    # (backend/tools/inpaint_tools.py:7-29, backend/main.py:229-245); the generator's windows / reference sets and RAFT's short-clip
    # loop (propainter_inpaint.py:221-247,317-343) depend on L.  The oracle needs ~1 h of CPU for these: their results are committed
    # as sub-sampled fixtures (tests/golden/pp_1080p_L68.npz, made by `python -m tests._baseline_oracle --fixture NAME`).
    "pp_1080p_L68": dict(kind="pp", H=1080, W=1920, box=(950, 1070, 288, 1632), L=68, seed=8),
    "pp_1080p_L44": dict(kind="pp", H=1080, W=1920, box=(950, 1070, 288, 1632), L=44, seed=9),
    # sttn-det on a portrait clip (sttn_det_inpaint.py:48-51: split_h = int(H*5/9)), a 25-frame tail batch
    "det_portrait": dict(kind="det", H=1920, W=1080, box=(1500, 1620, 108, 972), L=25, seed=10),
}

# jobs whose oracle run is too long for the GPU box's clock: the result comes from a committed, sub-sampled fixture
FIXTURE_JOBS = {"pp_1080p_L68": 3, "pp_1080p_L44": 4}             # name -> spatial stride of the sample
GOLDEN = os.path.join(ROOT, "tests", "golden")


def fast_clip(L, H, W, box, seed):
    """Seeded clip like vsr_amd.synth.make_clip (drifting smooth background, sensor noise, glyph blocks in the box) built from one
    background and one noise field, so that fifty 4K frames take seconds, not minutes.  uint8 BGR [L,H,W,3]."""
    rng = np.random.default_rng(seed)
    gh, gw = H // 40 + 3, W // 40 + 3
    base = rng.random((gh, gw, 3)).astype(np.float32)
    ys, xs = np.linspace(0, gh - 2, H + 64).astype(np.float32), np.linspace(0, gw - 2, W + 64).astype(np.float32)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    big = ((1 - fy) * (1 - fx) * base[y0][:, x0] + (1 - fy) * fx * base[y0][:, x0 + 1]
           + fy * (1 - fx) * base[y0 + 1][:, x0] + fy * fx * base[y0 + 1][:, x0 + 1])
    big = (big * 200 + 25).astype(np.int16)
    noise = rng.integers(-5, 6, size=(H + 64, W + 64, 3)).astype(np.int16)
    ymin, ymax, xmin, xmax = box
    frames = np.empty((L, H, W, 3), dtype=np.uint8)
    for i in range(L):
        dy, dx = (i * 2) % 64, (i * 3) % 64
        ny, nx = (i * 7) % 64, (i * 13) % 64
        img = np.clip(big[dy:dy + H, dx:dx + W] + noise[ny:ny + H, nx:nx + W], 0, 255).astype(np.uint8)
        grng = np.random.default_rng(seed * 1000 + i // 24)
        gpx = max((ymax - ymin) // 2, 4)
        gy = ymin + (ymax - ymin - gpx) // 2
        x = xmin + 8
        while x + gpx < xmax - 8:
            wpx = int(grng.integers(gpx // 2, gpx + 1))
            if grng.random() < 0.8:
                img[gy:gy + gpx, x:x + wpx] = 16
                img[gy + 2:gy + gpx - 2, x + 2:x + wpx - 2] = 250
            x += wpx + max(gpx // 4, 2)
        frames[i] = img
    return frames


def job_inputs(name):
    """(clip uint8 [L,H,W,3], mask uint8 {0,255} [H,W], job dict) -- the same arrays on the oracle and on the GPU side."""
    j = JOBS[name]
    return fast_clip(j["L"], j["H"], j["W"], j["box"], j["seed"]), job_mask(name), j


def strip_rows(name):
    """(ymin, ymax) of the rows a job's result holds (the one inpaint area of the job's mask)."""
    from oracle.sttn_auto import get_inpaint_area_by_mask

    mask, j = job_mask(name), JOBS[name]
    W, H = j["W"], j["H"]
    h = {"auto": int(W * 3 / 16), "det": int(H * 5 / 9) if H > W else int(W * 5 / 18), "pp": int(W * 3 / 16)}[j["kind"]]
    (a,) = get_inpaint_area_by_mask(W, H, h, mask[:, :, None], multiple=8 if j["kind"] == "pp" else 1)
    return a


def job_mask(name):
    from oracle.sttn_auto import create_mask

    j = JOBS[name]
    b = j["box"]
    return create_mask((j["H"], j["W"]), [(b[2], b[3], b[0], b[1])])


def run_job(name):
    """The oracle's output rows [ymin:ymax] of every frame, uint8 [L, ymax-ymin, W', 3]."""
    import torch

    from oracle import cv2_restate as cv2r
    from vsr_amd.synth import make_state_dict

    clip, mask, j = job_inputs(name)
    y0, y1, x0, x1 = strip_rows(name)
    if j["kind"] == "auto":
        from oracle.sttn_auto import STTNInpaintOracle, get_inpaint_area_by_mask

        o = STTNInpaintOracle(make_state_dict(0, "auto"), "auto")               # config defaults: stride 5, refs every 10
        mask01 = cv2r.threshold_binary(mask, 127, 1)[:, :, None]
        areas = get_inpaint_area_by_mask(j["W"], j["H"], int(j["W"] * 3 / 16), mask01)
        out = np.stack(o.chunk(list(clip), mask01, areas))
    elif j["kind"] == "det":
        from oracle.sttn_det import STTNDetOracle

        out = np.stack(STTNDetOracle(make_state_dict(1, "det"))(list(clip), mask))
    else:
        from oracle.propainter import ProPainterOracle
        from oracle.propainter_wrapper import PropainterOracle
        from oracle.raft import RaftOracle
        from oracle.rfc import RfcOracle
        from vsr_amd.synth import make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

        class ChunkedRaft(RaftOracle):              # pairs are independent (propainter_inpaint.py:219-247 chunks them too): bound the RAM
            def flows_bi(self, frames, iters=20):
                f, b = [], []
                for s in range(0, frames.shape[0] - 1, 3):
                    ff, bb = RaftOracle.flows_bi(self, frames[s:s + 4], iters)
                    f.append(ff)
                    b.append(bb)
                return torch.cat(f), torch.cat(b)

        ora = PropainterOracle(ChunkedRaft(make_raft_state_dict(0)), RfcOracle(make_rfc_state_dict(0)),
                               ProPainterOracle(make_propainter_state_dict(0)), raft_iter=20)
        out = np.stack(ora(list(clip), mask))
    assert np.array_equal(out[:, :y0], clip[:, :y0]) and np.array_equal(out[:, y1:], clip[:, y1:])
    return np.ascontiguousarray(out[:, y0:y1])


# ---- subprocess plumbing ------------------------------------------------------------------------------------------------
_running = {}


def launch(names=None):
    """start the jobs that are not running yet; the host threads are split between them"""
    names = [n for n in (names or JOBS) if n not in _running]
    if not names:
        return
    ncpu = os.cpu_count() or 8
    base = max(4, min(32, ncpu // max(1, len(names))))         # oneDNN on 30x160 / 60x108 maps stops scaling around 32 threads
    tmp = tempfile.mkdtemp(prefix="vsr_baseline_")
    for n in names:
        out = os.path.join(tmp, n + ".npy")
        # every job keeps the thread count the suite was measured with: on this host oneDNN gets SLOWER beyond ~32 threads on maps of
        # this size (bench.py's probe: 16 threads 0.46 s, 64 threads 1.19 s, 256 threads 75 s for the same three frames)
        threads = base
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), VSR_ORACLE_THREADS=str(threads),
                   HIP_VISIBLE_DEVICES="", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        log = open(os.path.join(tmp, n + ".log"), "w")
        p = subprocess.Popen([sys.executable, "-m", "tests._baseline_oracle", n, out], cwd=ROOT, env=env, stdout=log, stderr=subprocess.STDOUT)
        _running[n] = (p, out, log.name)


def result(name, timeout=1500):
    launch([name])
    p, out, log = _running[name]
    rc = p.wait(timeout=timeout)
    if rc != 0 or not os.path.exists(out):
        raise RuntimeError(f"oracle job {name} failed (rc {rc}):\n" + open(log).read()[-3000:])
    return np.load(out)


def fixture_sample(name, rows, clip_rows):
    """What a fixture keeps of the oracle's rows [L,h,W,3]: the pixels the oracle repainted (`changed`, [h,W] bool), the bounding
    box of them, every frame's box sub-sampled by the job's stride, and two frames' boxes in full."""
    st = FIXTURE_JOBS[name]
    changed = (rows != clip_rows).any(axis=(0, 3))
    ys, xs = np.where(changed.any(1))[0], np.where(changed.any(0))[0]
    y0, y1, x0, x1 = int(ys[0]), int(ys[-1]) + 1, int(xs[0]), int(xs[-1]) + 1
    full = [0, rows.shape[0] // 2]
    return dict(changed=np.packbits(changed), shape=np.array(changed.shape), box=np.array([y0, y1, x0, x1]), stride=np.array(st),
                sample=np.ascontiguousarray(rows[:, y0:y1:st, x0:x1:st]), full_frames=np.array(full),
                full=np.ascontiguousarray(rows[full, y0:y1, x0:x1]))


def fixture(name):
    """the committed sample of a FIXTURE_JOBS oracle run (dict of arrays, `changed` unpacked)"""
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    h, w = z["shape"]
    z["changed"] = np.unpackbits(z["changed"])[:h * w].reshape(h, w).astype(bool)
    return z


def main(argv):
    import time

    import torch

    if argv[0] == "--fixture":                      # python -m tests._baseline_oracle --fixture pp_1080p_L68   (about an hour of CPU)
        name = argv[1]
        torch.set_num_threads(int(os.environ.get("VSR_ORACLE_THREADS", "8")))
        import vsr_amd  # noqa: F401

        t0 = time.time()
        rows = run_job(name)
        clip, _, j = job_inputs(name)
        y0, y1, _, _ = strip_rows(name)
        np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **fixture_sample(name, rows, clip[:, y0:y1]))
        if len(argv) > 2:
            np.save(argv[2], rows)
        print(f"{name}: fixture from {rows.shape} in {time.time() - t0:.1f} s with {torch.get_num_threads()} threads")
        return
    name, out = argv
    torch.set_num_threads(int(os.environ.get("VSR_ORACLE_THREADS", "8")))
    import vsr_amd  # noqa: F401

    t0 = time.time()
    arr = run_job(name)
    np.save(out + ".tmp.npy", arr)
    os.replace(out + ".tmp.npy", out)
    print(f"{name}: {arr.shape} in {time.time() - t0:.1f} s with {torch.get_num_threads()} threads")


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    main(sys.argv[1:])
