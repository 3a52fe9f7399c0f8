#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configurations on ONE GPU (units resident in HBM, same timing rules as bench.py, which keeps
the headline config).  bench.py calls `run_all()` after its timed region and puts the result into the driver's JSON line as `configs`;
run on its own this prints one JSON line per configuration.

  config 2: 720p  sttn-auto, 50-frame chunks, fp32
  config 3: 1080p sttn-det, batch_generator sizes of a 1200-frame interval (25 x 47 + 25), fp32; the detector forward beside it
  config 4: 1080p propainter, the 68-frame batches batch_generator(1200, 70) makes (strip 1920x360, 20 RAFT iterations): exact fp32 and
            the reference's GPU arithmetic (RAFT fp32; flow completion + generator on fp16 operands, fp32 accumulation)
  config 5: 4K    sttn-auto, 50-frame chunks, fp16 operands (per GPU; the 8-GPU run is the driver's)

Every entry carries a `roofline` of its dominant gather-GEMM kernel symbol: algorithmic FLOPs / HIP-event time on the launch stream
(vsr_sttn_timing / vsr_flow_timing), against the dense MFMA peak of the arithmetic it runs in (MI355X_MICROARCH.md).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import vsr_amd  # noqa: E402,F401
from vsr_amd.backend.tools.inpaint_tools import batch_generator, create_mask, get_inpaint_area_by_mask, threshold_mask  # noqa: E402
from vsr_amd.engine import SttnEngine  # noqa: E402
from vsr_amd.synth import make_state_dict  # noqa: E402

PEAK_FP32 = 157.3          # TFLOP/s, v_mfma_f32_32x32x2_f32 dense (MI355X_MICROARCH.md)
PEAK_F16 = 2500.0          # TFLOP/s, dense f16 MFMA
RES = {"720p": (720, 1280, (620, 700, 192, 1088)), "1080p": (1080, 1920, (950, 1070, 288, 1632)),
       "4k": (2160, 3840, (1900, 2140, 576, 3264))}
TILE_DIMS = {0: (128, 128, 2, 2), 1: (256, 32, 4, 1), 2: (256, 64, 4, 1), 3: (128, 64, 2, 2)}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "config_traffic.json")     # written by scripts/pmc_configs.py from rocprofv3 --pmc passes
UNIT_ONLY = False           # scripts/pmc_configs.py: one warm unit, a marker kernel, ONE measured unit, a marker kernel -- nothing else
STAGE_MARKERS = False       # scripts/stage_stats.py (propainter legs): a marker kernel after every stage of one profiled batch


def pmc_marker():
    """a kernel whose name appears nowhere else in a run (at::native nextafter): scripts/pmc_configs.py counts the dispatches between
    two of them"""
    torch.cuda.synchronize()
    a = torch.zeros(64, device="cuda")
    torch.nextafter(a, a + 1)
    torch.cuda.synchronize()


def unit_only(step):
    step()
    pmc_marker()
    step()
    pmc_marker()
    return {"unit_only": True}


def attach_traffic(leg, out, units_per_s):
    """roofline.traffic (HBM bytes per launch of the dominant kernel symbol) and the whole unit's HBM bytes -> hbm_gbps, from the
    committed PMC summary of this leg (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950 correction applied there)"""
    try:
        t = json.load(open(TRAFFIC_FILE))["legs"].get(leg)
    except (OSError, ValueError, KeyError):
        t = None
    if not t:
        return out
    roofs = [out.get("roofline")] + [st.get("roofline") for st in (out.get("stages") or {}).values() if isinstance(st, dict)]
    for r in roofs:
        if not r or not r.get("kernel"):
            continue
        k = t["kernels"].get(r["kernel"])
        if k:
            r["traffic"] = k["hbm_bytes_per_launch"]
            r["traffic_launches_counted"] = k["launches"]
            if r.get("avg_launch_ms"):
                r["hbm_gbps_in_kernel"] = round(k["hbm_bytes_per_launch"] / r["avg_launch_ms"] / 1e6, 1)
    out["hbm_gb_per_unit"] = round(t["hbm_bytes_per_unit"] / 1e9, 3)
    if units_per_s:
        out["hbm_gbps"] = round(t["hbm_bytes_per_unit"] * units_per_s / 1e9, 1)
        out["hbm_frac_of_peak"] = round(t["hbm_bytes_per_unit"] * units_per_s / 8.0e12, 4)
    out["hbm_note"] = t.get("note")
    return out


def make_chunk_on_device(L, H, W, box, seed, device):
    """Seeded synthetic clip (vsr_amd.synth) -- 10 generated frames, extended to L by rolling (bench.py's)."""
    from vsr_amd import synth

    base = synth.make_clip(min(L, 10), H, W, box, seed=seed)
    d = torch.from_numpy(base).to(device)
    reps = [torch.roll(d, shifts=(3 * k, 5 * k), dims=(1, 2)) for k in range((L + base.shape[0] - 1) // base.shape[0])]
    return torch.cat(reps, 0)[:L].contiguous()


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def sttn_kernels_timed(eng, precision="f32"):
    """{kernel symbol: (ms, launches, flops)} from the engine's HIP-event records (vsr_sttn_timing_get)"""
    per_kernel = {}
    for cfg, (bm, bn, wm, wn) in TILE_DIMS.items():
        for bmode in (0, 1):
            for var, sym in ((1, "gather_gemm_f32"), (2, "gather_gemm_f32_v2"), (3, "gather_gemm_f32_v3"),
                             (4, "gather_gemm_f32_v4"), (5, "gather_gemm_f32_v5"), (6, "gather_gemm_f32_v5")):
                a, b, c = eng.timing_get(f"kernel:gg:{cfg}:{bmode}:v{var}")
                if var == 1 and bmode == 1:        # timing_get matches by prefix: "...:v1" also counts "...:v1x" (below)
                    ax, bx, cx = eng.timing_get(f"kernel:gg:{cfg}:1:v1x")
                    a, b, c = a - ax, b - bx, c - cx
                if b:
                    per_kernel[f"{sym}<{bm}, {bn}, {wm}, {wn}, {bmode}>"] = (a, b, c)
            a, b, c = eng.timing_get(f"kernel:gg:{cfg}:1:v1x")          # P.V of the fused attention (gather_gemm_pvx.h)
            if b:
                per_kernel[f"gather_gemm_f32_aexp<{bm}, {bn}, {wm}, {wn}>"] = (a, b, c)
    a, b, c = eng.timing_get("kernel:gg:5:0:v7")                    # split-format modes: the 256 x 256 tile (gather_gemm_v7.h)
    if b:
        per_kernel["gather_gemm_f16_v7<%d>" % (0 if precision == "f16" else 1)] = (a, b, c)
    return per_kernel


def roofline_of(per_kernel, peak, what):
    """the `roofline` object of the kernel symbol with the largest total time, plus the rate over every gather-GEMM launch"""
    if not per_kernel:
        return None
    dom = max(per_kernel, key=lambda k: per_kernel[k][0])
    ms, n, fl = per_kernel[dom]
    ach = fl / ms / 1e9 if ms > 0 else 0.0
    tms, tfl = sum(v[0] for v in per_kernel.values()), sum(v[2] for v in per_kernel.values())
    return {"bound": "mfma", "kernel": dom, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
            "launches": int(n), "avg_launch_ms": round(ms / n, 4) if n else None, "flops_per_launch": round(fl / n) if n else None,
            "share_of_gemm_time": round(ms / tms, 3) if tms > 0 else None,
            "every_gemm_launch": {"achieved": round(tfl / tms / 1e9, 2) if tms > 0 else None, "frac": round(tfl / tms / 1e9 / peak, 4) if tms > 0 else None},
            "measured_on": what}


def sttn_roofline(eng, step, precision, steps=2):
    """single-lane pass with HIP events around every launch (outside any timed region)"""
    eng.set_lanes(1)
    step()
    torch.cuda.synchronize()
    eng.timing_reset()
    eng.timing(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    eng.timing(False)
    r = roofline_of(sttn_kernels_timed(eng, precision), PEAK_FP32 if precision == "f32" else PEAK_F16,
                    f"{steps} single-lane units, HIP events around every launch, after the timed passes")
    eng.set_lanes(2)
    return r


def run_auto(name, res, precision, steps=4, warmup=1, L=50):
    H, W, box = RES[res]
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=0, precision=precision)
    mask01 = threshold_mask(create_mask((H, W), [(box[2], box[3], box[0], box[1])]))
    areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask01)
    dmask = torch.from_numpy(np.ascontiguousarray(mask01[:, :, 0])).cuda()
    src = make_chunk_on_device(L, H, W, box, seed=2, device=torch.device("cuda", 0))
    work = src.clone()

    def step():
        work.copy_(src)
        eng.auto_chunk(work, dmask, areas)

    if UNIT_ONLY:
        r = unit_only(step)
        eng.close()
        return r
    dt = timed(step, steps, warmup)
    fps = steps * L / dt
    fl = eng.chunk_flops(L, dmask, areas)
    peak = PEAK_FP32 if precision == "f32" else PEAK_F16
    out = {"config": name, "mode": "sttn-auto", "res": res, "dtype": {"f32": "f32", "f16": "f16 operands, f32 accumulate"}[precision], "chunk_frames": L,
           "value": round(fps, 2), "unit": "frames/s", "ms_per_chunk": round(dt / steps * 1e3, 2), "gflop_per_frame": round(fl / L / 1e9, 1),
           "model_tflops": round(fl / L * fps / 1e12, 2), "model_frac_of_peak": round(fl / L * fps / 1e12 / peak, 4),
           "fp32_fallback_chunks": eng.fallbacks(), "roofline": sttn_roofline(eng, step, precision)}
    eng.close()
    return attach_traffic(name.split(":")[0], out, fps / L)


def run_det(name, res, precision, total=1200):
    H, W, box = RES[res]
    eng = SttnEngine(make_state_dict(0, "det"), "det", device=0, precision=precision)
    mask = create_mask((H, W), [(box[2], box[3], box[0], box[1])])
    areas = get_inpaint_area_by_mask(W, H, int(W * 5 / 18), mask[:, :, None] if mask.ndim == 2 else mask)
    dmask = torch.from_numpy(np.ascontiguousarray(mask if mask.ndim == 2 else mask[:, :, 0])).cuda()
    sizes = [len(b) for b in batch_generator(list(range(total)), 50)]          # 25 x 47 + 25 for 1200 frames
    Lmax = max(sizes)
    src = make_chunk_on_device(Lmax, H, W, box, seed=3, device=torch.device("cuda", 0))
    work = src.clone()
    sample = sorted(set(sizes))                                               # time each distinct batch size, weight by count

    def make(L):
        def step():
            work[:L].copy_(src[:L])
            eng.det_batch(work[:L], dmask, areas)
        return step

    if UNIT_ONLY:
        r = unit_only(make(Lmax))
        eng.close()
        return r
    per = {}
    for L in sample:
        dt = timed(make(L), 3 if L == Lmax else 2, 1)
        per[L] = dt / (3 if L == Lmax else 2)
    wall = sum(per[L] for L in sizes)
    fps = total / wall
    flops = sum(eng.chunk_flops(L, dmask, areas) for L in sizes)          # what is contracted (last block / decoder rows trimmed to what is read)
    out = {"config": name, "mode": "sttn-det", "res": res, "dtype": "f32", "batches": f"{sizes.count(Lmax)}x{Lmax}+{sizes[-1]}",
           "value": round(fps, 2), "unit": "frames/s", "ms_per_batch": {str(L): round(per[L] * 1e3, 2) for L in sample},
           "model_tflops": round(flops / wall / 1e12, 2), "model_frac_of_peak": round(flops / wall / 1e12 / PEAK_FP32, 4),
           "gflop_per_frame": round(flops / total / 1e9, 1), "roofline": sttn_roofline(eng, make(Lmax), precision),
           "note": "inpainting only, the known box injected on every frame; the detector forward is the `detector` entry"}
    eng.close()
    return attach_traffic("3", out, 1.0 / per[Lmax])


def run_detector(name, nb=8, reps=5):
    """config 3's other half: the PP-OCRv5 server detector's forward at the 1080p net input (960x544), `nb` frames per forward, from the
    compiled NHWC plan's launch list (subtitle_detect.py:41-82's TextDetection.predict without the DB post-process)."""
    from vsr_amd.backend.tools import ocr_det
    from vsr_amd.backend.tools.paddle_graph import load_graph
    from vsr_amd.synth import make_det_weights

    g = load_graph(os.path.join(ROOT, "tests", "golden", "ppocr_det_graph.json"))
    det = ocr_det.TextDetection(g, make_det_weights(g), device=0)
    det.use_tape = True
    img = np.random.default_rng(3).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
    imgs = [img] * nb
    if UNIT_ONLY:
        det.probability_maps(imgs)
        r = unit_only(lambda: det.probability_maps(imgs))
        det.runner.close()
        return r
    dt = timed(lambda: det.probability_maps(imgs), reps, 3)
    ms = dt / reps / nb * 1e3
    gflop = det.gflop_per_frame(544, 960)             # from the walk of the program itself (270.8 for the server program)
    tf = gflop / ms
    out = {"config": name, "mode": "text detector forward (server program)", "res": "1080p -> 960x544 net input", "dtype": "f32",
           "value": round(1e3 / ms, 1), "unit": "frames/s", "ms_per_frame": round(ms, 3), "frames_per_forward": nb, "gflop_per_frame": round(gflop, 1),
           "roofline": {"bound": "mfma", "kernel": "whole forward (NHWC-resident plan: 114 gather-GEMM steps, 27 depthwise, 2 per-pixel dots, stem; 150 launches)",
                        "achieved": round(tf, 2), "peak": PEAK_FP32, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32, 4), "traffic": None,
                        "measured_on": f"{reps} forwards of {nb} frames, wall clock around the recorded launch list"}}
    det.runner.close()
    return attach_traffic("3d", out, 1e3 / ms / nb)


def flow_kernel_symbol(name):
    """'gg:<cfg>:<bmode>:v<variant>' -> the kernel symbol the flow engines launch for it"""
    _, cfg, bmode, var = name.split(":")
    if cfg == "flash":                     # the generator's fused window attention (pp_attn_kernels.hip)
        return "k_pp_flash_attn_f16" if var == "v7" else "k_pp_flash_attn_f32"
    if cfg == "narrow":                    # GEMMs of <= 4 output columns on the dot-product kernel (gather_gemm_narrow.h)
        return "gather_gemm_f32_narrow"
    bm, bn, wm, wn = TILE_DIMS[int(cfg)]
    v = int(var[1:])
    if v == 3:
        return f"gather_gemm_f32_v3<{bm}, {bn}, {wm}, {wn}, {bmode}>"
    return f"gather_gemm_f32_v4<{bm}, {bn}, {wm}, {wn}, {bmode}, {'true' if v == 7 else 'false'}>"


def run_propainter(name, precision="f32", L=68, reps=1):
    """one PropainterInpaint.inpaint call per rep on an HBM-resident uint8 strip batch [L,360,1920,3] (what the resident loop hands the
    plugin for a 1080p clip); then one profiled call: per-stage seconds / FLOPs and per-kernel HIP-event times."""
    from vsr_amd import engine as E
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.synth import make_clip, make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

    H, W = 360, 1920
    box = (H // 2, H - H // 6, W // 6, W - W // 6)
    base = make_clip(10, H, W, box, seed=4)
    d = torch.from_numpy(base).cuda()
    frames = torch.cat([torch.roll(d, shifts=(2 * k, 3 * k), dims=(1, 2)) for k in range((L + 9) // 10)], 0)[:L].contiguous()
    mask = np.zeros((H, W), np.uint8)
    mask[box[0]:box[1], box[2]:box[3]] = 255
    plug = PropainterInpaint("cuda:0", {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0), "propainter": make_propainter_state_dict(0)},
                             precision=precision)
    if UNIT_ONLY:
        r = unit_only(lambda: plug.inpaint(frames, mask))
        plug.close()
        return r
    if STAGE_MARKERS:
        # scripts/stage_stats.py: one warm call, then ONE profiled call (every stage on one stream, device-synchronised between stages)
        # in which a marker kernel closes every stage -- the kernel trace is cut at the markers into one summary per engine
        plug.profile = {}
        plug.inpaint(frames, mask)
        plug.profile = {}
        ops = {"raft": torch.nextafter, "flow_completion": torch.hypot, "other": torch.copysign, "generator": torch.logaddexp}
        z = torch.ones(64, device="cuda")

        def marker(stage):
            ops[stage](z, z + 1)

        torch.cuda.synchronize()
        torch.fmod(z, z + 1)                     # "begin": everything before it is warm-up
        plug.stage_marker = marker
        plug.inpaint(frames, mask)
        torch.cuda.synchronize()
        plug.stage_marker = None
        plug.close()
        return {"stage_markers": True}
    dt = timed(lambda: plug.inpaint(frames, mask), reps, 1) / reps
    # profiled call (device-synchronised around every stage + events around every launch: slower than the timed one)
    plug.profile = {}
    plug.inpaint(frames, mask)               # (a profiled call runs everything on one stream: the plans / workspaces that needs are built here)
    plug.profile = {}
    E.flow_timing_reset()
    E.flow_timing(True)
    plug.inpaint(frames, mask)
    torch.cuda.synchronize()
    E.flow_timing(False)
    prof, plug.profile = plug.profile, None
    modes = dict(zip(("raft", "rfc", "pp"), PropainterInpaint.PRECISIONS[precision]))
    stages = {}
    total_fl = 0.0
    for stage, eng in (("raft", "raft"), ("flow_completion", "rfc"), ("generator", "pp")):
        s, fl = prof.get(stage, [0.0, 0.0])
        total_fl += fl
        peak = PEAK_FP32 if modes[eng] == "f32" else PEAK_F16
        kern = {flow_kernel_symbol(k): v for k, v in E.flow_timing_by_kernel(eng).items() if k != "op"}
        opms = E.flow_timing_get(f"{eng}:op:")[0]
        stages[stage] = {"s": round(s, 4), "tflop": round(fl / 1e12, 3), "tflops": round(fl / s / 1e12, 2) if s > 0 else None,
                         "arithmetic": {"f32": "f32", "split": "f32 (fp16 hi/lo operand pairs)", "f16": "f16 operands, f32 accumulate"}[modes[eng]],
                         "frac_of_peak": round(fl / s / 1e12 / peak, 4) if s > 0 else None, "non_gemm_kernel_ms": round(opms, 2),
                         "roofline": roofline_of(kern, peak, "one profiled call, HIP events around every launch of the engine's plans")}
    stages["other"] = {"s": round(prof.get("other", [0.0, 0.0])[0], 4), "what": "image propagation, normalise / compose kernels, mask upload"}
    fb = [e.fallbacks() for e in (plug.fix_raft, plug.fix_flow_complete, plug.model)]
    psnr = None
    if precision != "f32":                   # the same batch in the exact mode of the same engines: PSNR of the frames over the repainted pixels
        got = plug.inpaint(frames, mask)
        for e in (plug.fix_raft, plug.fix_flow_complete, plug.model):
            e.set_precision("f32")
        ref = plug.inpaint(frames, mask)
        ch = (ref != frames).any(dim=3)
        mse = float(((got[ch].float() - ref[ch].float()) ** 2).mean().item()) if bool(ch.any()) else 0.0
        psnr = "inf" if mse == 0 else round(float(20 * np.log10(255.0 / np.sqrt(mse))), 2)
    plug.close()
    dom = max(("raft", "flow_completion", "generator"), key=lambda k: stages[k]["s"])
    out = {"config": name, "mode": "propainter", "res": "1080p (strip 1920x360)", "batch_frames": L, "raft_iters": 20,
           "dtype": {"f32": "f32", "f16": "f16 operands + f32 accumulate for flow completion and generator, RAFT f32 (the reference's GPU arithmetic)",
                     "f16-raft-split": "f16 operands + f32 accumulate; RAFT on fp16 hi/lo operand pairs", "split": "f32 (fp16 hi/lo operand pairs)"}[precision],
           "value": round(L / dt, 2), "unit": "frames/s", "s_per_batch": round(dt, 3), "tflop_per_frame": round(total_fl / L / 1e12, 3),
           "model_tflops": round(total_fl / dt / 1e12, 2), "range_guard_fallbacks": fb, "psnr_db_vs_exact_mode": psnr, "stages": stages, "roofline": stages[dom]["roofline"],
           "roofline_stage": dom}
    return attach_traffic({"f32": "4", "f16": "4h", "f16-raft-split": "4s", "split": "4x"}[precision], out, 1.0 / dt)


def run_e2e_det(name, frames=1200):
    """BASELINE config 3 AS STATED: a 1080p clip of `frames` frames through SubtitleRemover.run() in --inpaint-mode sttn-det, file to file
    (y4m in -> detector pass over the sampled frames -> inpainting -> y4m out), in a process of its own (scripts/bench_e2e.py; the clip
    repeats its first 50 frames: every stage's cost is content-independent).  The detector's probability map is injected at the graph
    output (synthetic weights find no text); the forward, the DB post-process and everything else are the product's."""
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "scripts", "bench_e2e.py"), "--frames", str(frames), "--always-on", "--mode", "sttn-det", "--cycle", "50"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"config": name, "error": (r.stderr or r.stdout)[-300:]}
    d = json.loads(lines[-1])
    return {"config": name, "mode": "sttn-det, file to file, detector pass included", "res": "1080p", "dtype": "f32", "value": d["value"], "unit": "frames/s",
            "frames": d["frames"], "wall_s": d["wall_s"], "phases_s": d["phases_s"], "detector": d["detector"], "frames_written": d["frames_written"],
            "clip": d["clip"], "roofline": None,
            "note": "whole-run wall clock incl. reading and writing the y4m files; the inpainting alone is the `3` entry, the detector forward alone `3d`"}


LEGS = {
    "1": lambda: run_auto("1: 1080p sttn-auto fp32 (the headline's unit; bench.py times it)", "1080p", "f32"),
    "2": lambda: run_auto("2: 720p sttn-auto fp32", "720p", "f32"),
    "3": lambda: run_det("3: 1080p sttn-det fp32, 47-frame batches", "1080p", "f32"),
    "3d": lambda: run_detector("3: text detector forward"),
    "3e": lambda: run_e2e_det("3: 1080p x 1200 frames sttn-det as stated (file to file, detector pass + inpainting)"),
    "4": lambda: run_propainter("4: 1080p propainter fp32, 68-frame batch", "f32"),
    "4h": lambda: run_propainter("4: 1080p propainter, reference GPU arithmetic (f16 operands; RAFT f32)", "f16"),
    "4s": lambda: run_propainter("4: 1080p propainter, f16 operands; RAFT on hi/lo pairs", "f16-raft-split"),
    "5": lambda: run_auto("5: 4K sttn-auto fp16 operands (one GPU of the 8)", "4k", "f16"),
    "5x": lambda: run_auto("5: 4K sttn-auto fp32 (exact mode)", "4k", "f32", steps=2),
}
DEFAULT = ["2", "3", "3d", "3e", "4", "4h", "5"]


def run_all(which=None):
    """{leg: result dict}; a leg that raises is reported as {"error": ...} -- the configs are informational beside the headline"""
    out = {}
    for k in which or DEFAULT:
        t0 = time.perf_counter()
        try:
            out[k] = LEGS[k]()
        except Exception as e:      # noqa: BLE001
            out[k] = {"error": repr(e)[:300]}
        out[k]["leg_seconds"] = round(time.perf_counter() - t0, 1)
        torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    argv = sys.argv[1:]
    if argv and argv[0] == "--unit":        # scripts/pmc_configs.py
        UNIT_ONLY = True
        argv = argv[1:]
    if argv and argv[0] == "--stages":      # scripts/stage_stats.py
        STAGE_MARKERS = True
        argv = argv[1:]
    for k, v in run_all(argv or None).items():
        print(json.dumps(v), flush=True)
