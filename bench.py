#!/usr/bin/env python
"""Headline benchmark: inpainted frames/sec @1080p, STTN (sttn-auto), 5-frame window stride.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one chunk: ``clip_gap`` (50) synthetic 1080p frames
already resident in HBM go through vsr_sttn_auto_chunk (crop strip -> cv2-style resize to 640x120
-> encoder -> 10 sliding windows x 8 transformer blocks -> decoder -> tanh/u8/overlap average ->
resize back -> mask blend, in place).  Chunks are independent (sttn_auto_inpaint.py:242-328), so
with N GPUs the chunks are dealt round-robin and no data-path collective exists ("weak" scaling:
per-GPU work is fixed).  N > 1 times the north-star data path: all N chunks of a step are resident in
rank 0's HBM, the strip rows of chunk k travel to rank k over xGMI (RCCL send / recv, grouped, pipelined
under the compute: backend/tools/chunk_parallel.py), are inpainted there and return to rank 0's HBM;
`value` = N x 50 x K frames / max-over-ranks time.  The replica rate (every rank on its own resident
chunk, nothing exchanged) is reported beside it as `replicas`.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the dominant kernel symbol (the gather-GEMM instantiation with the largest total time:
                  gather_gemm_f32_v3<128, 64, 2, 2, 0>, the 3x3 / 1x1 convs and QK^T as implicit GEMMs, 83 % of
                  the GPU time), algorithmic FLOPs / HIP-event time on the launch stream against the 157.3 TFLOP/s
                  fp32 MFMA peak of MI355X_MICROARCH.md; `every_gemm_launch` beside it is the same ratio over
                  ALL gather-GEMM launches of a chunk (every symbol).
  configs_multi -- N > 1 only: BASELINE.json's multi-GPU configurations through their own sharding (scripts/bench_multi.py): "5" = 4K
                  fp16-operand chunks through chunk_parallel, "4h" / "4" = 68-frame propainter batches through batch_parallel
                  (reference GPU arithmetic / exact fp32); each with value, replicas, efficiency, hbm_gbps; a watchdog per phase.
  configs      -- BASELINE.json's other configurations on this GPU, each with its own `roofline` (scripts/bench_configs.py:
                  2 = 720p sttn-auto, 3 = 1080p sttn-det 47-frame batches + the text detector's forward, 4 = 1080p
                  propainter 68-frame batches in exact fp32 and in the reference's GPU arithmetic, 5 = 4K sttn-auto on
                  fp16 operands); measured after the timed region, N = 1 only.
  full_work    -- `value` is measured on a plan that leaves out what the reference computes and nothing reads (DESIGN
                  4.3c; the frames are the same bit for bit; `gflop_per_frame` vs `gflop_per_frame_reference`); this is
                  the same step with all of it, from a child process (the library reads the switches once per process).
  cpu_baseline -- the CPU oracle (torch fp32 restatement of the reference modules, "port") timed on
                  this box's host cores on ONE full 50-frame chunk of the same clip, end to end (crop,
                  cv2-style resize, network, resize back, blend) with the network-only time beside it.
                  kind is "port", not "reference": /root/reference does not exist on the GPU box and its
                  wrappers need cv2; the port's network was checked identical (max |d| = 0.0) to the reference's
                  own module on the build container (tests/golden/sttn_auto_net.npz, oracle/make_golden.py).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL / device-memory IPC on this pool needs dmabuf handles (set before HIP starts)

import numpy as np
import torch

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_F16_MFMA_TFLOPS = 2500.0          # ... dense f16 / bf16 MFMA (the --precision modes only)
RES = {"720p": (720, 1280, (620, 700, 192, 1088)), "1080p": (1080, 1920, (950, 1070, 288, 1632)),
       "4k": (2160, 3840, (1900, 2140, 576, 3264))}


def make_chunk_on_device(L, H, W, box, seed, device):
    """Seeded synthetic clip (vsr_amd.synth) -- 10 generated frames, extended to L by rolling."""
    from vsr_amd import synth

    base = synth.make_clip(min(L, 10), H, W, box, seed=seed)
    d = torch.from_numpy(base).to(device)
    reps = [torch.roll(d, shifts=(3 * k, 5 * k), dims=(1, 2)) for k in range((L + base.shape[0] - 1) // base.shape[0])]
    return torch.cat(reps, 0)[:L].contiguous()


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def effective_cpus():
    """(cpus this process may actually use at once, what limits them): os.cpu_count() counts the machine's hardware threads, but a
    container may run under an affinity mask or a CFS quota (cgroup cpu.max) far below that -- then more threads or processes only
    get throttled"""
    ncpu = os.cpu_count() or 1
    info = {"cpu_count": ncpu}
    eff = float(ncpu)
    try:
        aff = len(os.sched_getaffinity(0))
        info["affinity"] = aff
        eff = min(eff, aff)
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    eff = min(eff, int(txt[0]) / int(txt[1]))
            else:
                q = int(txt[0])
                info["cgroup_cfs_quota_us"] = q
                if q > 0:
                    eff = min(eff, q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()))
            break
        except (OSError, ValueError, IndexError):
            continue
    try:
        info["loadavg"] = os.getloadavg()[0]
    except OSError:
        pass
    info["effective_cpus"] = round(eff, 1)
    return eff, info


def cpu_baseline(sd, clip, mask01, areas):
    """The oracle's chunk body (STTNAutoInpaint.__call__ :242-317 restated) on one full chunk of host frames, timed on the host
    cores: end to end, and the network part (STTNInpaint.inpaint) on its own."""
    from oracle.sttn_auto import STTNInpaintOracle

    # pick the thread count that suits this host (oneDNN collapses when 256 threads fight over a 30x160 feature map):
    # one 3-frame STTNInpaint.inpaint (a single window through all 8 blocks + decoder) per candidate, keep the fastest
    ncpu = os.cpu_count() or 1
    probe = STTNInpaintOracle(sd, "auto")
    pf = list(np.random.default_rng(0).integers(0, 256, size=(3, 120, 640, 3), dtype=np.uint8))
    best, threads, tried = None, 1, {}
    for cand in sorted({c for c in (16, 32, 64) if c <= ncpu} or {ncpu}):      # all 256 hardware threads took 75 s for the probe alone (profiles/r02_bench.log)
        torch.set_num_threads(cand)
        probe.inpaint(pf[:1])
        t0 = time.perf_counter()
        probe.inpaint(pf)
        dt = time.perf_counter() - t0
        tried[cand] = round(dt, 3)
        if best is None or dt < best:
            best, threads = dt, cand
    torch.set_num_threads(threads)
    o = STTNInpaintOracle(sd, "auto")
    net = {"s": 0.0}
    inner = o.inpaint

    def timed_inpaint(frames):
        t = time.perf_counter()
        r = inner(frames)
        net["s"] += time.perf_counter() - t
        return r

    o.inpaint = timed_inpaint
    t0 = time.perf_counter()
    ref = o.chunk(list(clip), mask01, areas)
    dt = time.perf_counter() - t0
    return np.stack(ref), dt, net["s"], threads, tried


_PARALLEL_WORKER = r"""
import importlib.util, os, sys, time
idx, threads, L, root = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
try:                                        # one group of host threads per process
    ncpu = os.cpu_count() or 1
    os.sched_setaffinity(0, {c % ncpu for c in range(idx * threads, (idx + 1) * threads)})
except (AttributeError, OSError):
    pass
import numpy as np
import torch
torch.set_num_threads(threads)
sys.path.insert(0, root)
spec = importlib.util.spec_from_file_location("vsr_synth", os.path.join(root, "video-subtitle-remover_amd", "synth.py"))
synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)      # the synthetic checkpoint only: no HIP library in this process
from oracle.sttn_auto import STTNInpaintOracle
o = STTNInpaintOracle(synth.make_state_dict(0, "auto"), "auto")
frames = list(np.random.default_rng(100 + idx).integers(0, 256, size=(L, 120, 640, 3), dtype=np.uint8))
o.inpaint(frames[:2])
print("READY", flush=True)
sys.stdin.readline()
t0 = time.perf_counter()
o.inpaint(frames)
print("DONE %.3f" % (time.perf_counter() - t0), flush=True)
"""


def cpu_baseline_parallel(L, threads, budget_s=170.0, flops_sample=None, flops_per_frame=None):
    """The honest 'host cores of the same box' figure: chunks share no state (sttn_auto_inpaint.py:242-328), so the CPU
    path scales by running one chunk per group of cores.  cpu_count // threads plain Python processes (no GPU library loaded, pinned
    to their own `threads` hardware threads), each the network part of one full chunk, started together; aggregate frames/s =
    processes x L / wall.  None when the box is too small; never fails the bench line (a timeout is reported as such)."""
    import subprocess

    ncpu = os.cpu_count() or 1
    eff, limits = effective_cpus()
    nproc = max(1, int(eff) // max(threads, 1))
    if nproc < 2:
        return {"skipped": f"this process may use {eff:.1f} CPUs at once; the single-process figure already runs on {threads} threads", "limits": limits}
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="",
               CUDA_VISIBLE_DEVICES="")
    procs = [subprocess.Popen([sys.executable, "-c", _PARALLEL_WORKER, str(i), str(threads), str(L), ROOT], stdin=subprocess.PIPE,
                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for i in range(nproc)]
    deadline = time.perf_counter() + budget_s

    def read_line(p):
        import select

        while time.perf_counter() < deadline:
            r, _, _ = select.select([p.stdout], [], [], 1.0)
            if r:
                return p.stdout.readline().strip()
            if p.poll() is not None:
                return p.stdout.readline().strip() or f"EXIT {p.returncode}"
        return "TIMEOUT"

    try:
        ready = [read_line(p) for p in procs]
        if any(r != "READY" for r in ready):
            return {"error": f"workers not ready: {sorted(set(ready))}"}
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go\n")
            p.stdin.flush()
        done = [read_line(p) for p in procs]
        wall = time.perf_counter() - t0
        if not all(d.startswith("DONE") for d in done):
            return {"error": f"workers did not finish within {budget_s:.0f} s: {sorted(set(d.split()[0] for d in done))}"}
        per = [float(d.split()[1]) for d in done]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = {"value": round(nproc * L / wall, 3), "unit": "frames/s", "processes": nproc, "threads_per_process": threads,
           "cores": nproc * threads, "limits": limits, "wall_s": round(wall, 1), "slowest_process_s": round(max(per), 1), "fastest_process_s": round(min(per), 1),
           "sample": f"{nproc} independent {L}-frame chunks at once, one per process, STTNInpaint.inpaint (the network: 95 % of a chunk's "
                     f"CPU time) on {threads} torch threads each -- {nproc * threads} of the box's {ncpu} hardware threads ({eff:.0f} usable by this container)"}
    if flops_sample and flops_per_frame:
        # short chunks have short windows (attention cost grows with T^2): the rate that compares with `value` is FLOP-normalised
        res["tflops"] = round(nproc * flops_sample / wall / 1e12, 3)
        res["full_chunk_equivalent"] = {"value": round(nproc * flops_sample / wall / flops_per_frame, 3), "unit": "frames/s",
                                        "note": f"aggregate model FLOP/s of the sample / the {flops_per_frame / 1e9:.1f} GFLOP per frame of full 50-frame chunks"}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--res", default="1080p", choices=sorted(RES))
    ap.add_argument("--chunk", type=int, default=50, help="frames per chunk (config.sttnMaxLoadNum)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-parallel", action="store_true", help="skip cpu_baseline.parallel (one oracle chunk per 16 host threads, all at once)")
    ap.add_argument("--lanes", type=int, default=int(os.environ.get("VSR_STTN_LANES", "2")), choices=[1, 2, 3, 4],
                    help="streams a chunk's sliding windows are issued on (vsr_sttn_set_lanes; 2 = default of the library)")
    ap.add_argument("--no-selftest", action="store_true", help="N > 1: skip the check of the gathered chunks against each rank's replica result")
    ap.add_argument("--e2e-chunks", type=int, default=4, help="chunks of the PCIe-inclusive plugin leg (0 = skip)")
    ap.add_argument("--no-split-half", action="store_true", help="skip the informational split-half (f16 MFMA) leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` object (BASELINE configs 2-5, scripts/bench_configs.py)")
    ap.add_argument("--configs", default=None, help="comma-separated legs of scripts/bench_configs.py (default: 2,3,3d,3e,4,4h,5)")
    ap.add_argument("--no-multi-configs", action="store_true", help="N > 1: skip the N-rank legs of BASELINE configs 5 and 4 (scripts/bench_multi.py)")
    ap.add_argument("--multi-configs", default=None, help="N > 1: comma-separated legs of scripts/bench_multi.py (default: 5,4h,4)")
    ap.add_argument("--no-full-work", action="store_true",
                    help="skip the `full_work` leg: the same step with every row the reference's modules compute (a child process with "
                         "VSR_TRIM_LAST_BLOCK=0 VSR_DECODE_ROWS=0 -- the library reads these once per process)")
    ap.add_argument("--precision", default=None, choices=["f32", "split", "split-format", "f16"],
                    help="arithmetic of the contractions in the timed region (default: exact fp32; BASELINE.json's config 5 "
                         "is --res 4k --precision f16)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become `python -m torch.distributed.run --nproc-per-node N bench.py ...` (one
        # process per GPU; rank 0 still prints the ONE JSON line).  HSA_ENABLE_IPC_MODE_LEGACY=0 is already in the environment (above).
        import socket

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the MI355X path has no CPU fallback")
    # VSR_BENCH_DRYRUN_1GPU=1: exercise the N > 1 code path on a single-GPU box (every rank on cuda:0, gloo)
    dry = os.environ.get("VSR_BENCH_DRYRUN_1GPU") == "1"
    if dry:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_

        dist = dist_
        if dry:
            dist.init_process_group(backend="gloo")
        else:
            import datetime

            # a stuck exchange ends the run with an error after five minutes instead of holding the node
            dist.init_process_group(backend="nccl", device_id=device, timeout=datetime.timedelta(seconds=300))

    import vsr_amd  # noqa: F401
    from vsr_amd.backend.tools.inpaint_tools import create_mask, get_inpaint_area_by_mask, threshold_mask
    from vsr_amd.engine import SttnEngine
    from vsr_amd.synth import make_state_dict      # stand-in checkpoint (the real .pth files are missing blobs)

    H, W, box = RES[args.res]
    L = args.chunk
    sd = make_state_dict(0, "auto")
    eng = SttnEngine(sd, "auto", device=local_rank, precision=args.precision)
    eng.set_lanes(args.lanes)
    base_precision = args.precision or {"1": "split", "s": "split", "2": "split-format", "3": "f16"}.get(
        os.environ.get("VSR_PRECISION", "0")[:1], "f32")
    mask = create_mask((H, W), [(box[2], box[3], box[0], box[1])])
    mask01 = threshold_mask(mask)
    areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask01)
    dmask = torch.from_numpy(np.ascontiguousarray(mask01[:, :, 0])).to(device)
    src = make_chunk_on_device(L, H, W, box, seed=1 + rank, device=device)
    work = src.clone()

    def step():
        work.copy_(src)                          # device-to-device restore of the in-place chunk
        eng.auto_chunk(work, dmask, areas)

    def timed(fn):
        """barrier + synchronize on both sides of fn(); max over ranks"""
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    replicas = None
    selftest_failed = False
    if world == 1:
        for _ in range(args.warmup):
            step()
        torch.cuda.synchronize()
        # the timed region carries no events: with two lanes (vsr_sttn_set_lanes, the default) launches of the two streams overlap,
        # so per-launch durations are taken in a single-lane pass afterwards (roofline_leg) and the per-op breakdown after that
        elapsed = timed(lambda: [step() for _ in range(args.steps)])
    else:
        # N > 1: the north-star data path.  A step = one round of N chunks: all resident in rank 0's HBM, the rows between the
        # first and the last strip of chunk k go to rank k (RCCL send / recv over xGMI), are inpainted there in place and come
        # back into rank 0's HBM.  backend/tools/chunk_parallel.py pipelines scatter / compute / gather over the rounds.
        from vsr_amd.backend.tools import chunk_parallel as cp

        y_lo, y_hi = min(a[0] for a in areas), max(a[1] for a in areas)
        local_areas = [(a[0] - y_lo, a[1] - y_lo, a[2], a[3]) for a in areas]
        dmask_rows = dmask[y_lo:y_hi].contiguous()
        if rank == 0:
            srcs = [src] + [make_chunk_on_device(L, H, W, box, seed=1 + k, device=device) for k in range(1, world)]
            dsts = [torch.empty((L, y_hi - y_lo, W, 3), dtype=torch.uint8, device=device) for _ in range(world)]

        corrupt = os.environ.get("VSR_BENCH_SELFTEST_CORRUPT") == "1"      # test hook: flip one bit of the last rank's gathered rows

        def store_rows(i, rows):
            dsts[i % world].copy_(rows)
            if corrupt and i % world == world - 1:
                dsts[i % world][0, 0, 0, 0] ^= 1

        def run_rounds(n_rounds):
            ranges = [(i * L, (i + 1) * L) for i in range(n_rounds * world)]
            cp.run_chunk_parallel(ranges, (y_hi - y_lo, W, 3),
                                  lambda i, out: out.copy_(srcs[i % world][:, y_lo:y_hi]),
                                  lambda i, rows: eng.auto_chunk(rows, dmask_rows, local_areas),
                                  store_rows,
                                  dist=dist, device=device, io="device")

        # (1) every rank on its own resident chunk, nothing exchanged: the upper bound of the path below -- and what the line falls back
        # to when the exchange does not come through (it runs over RCCL between DEVICES for the first time on the driver's node)
        for _ in range(max(1, args.warmup)):
            step()
        dt_rep = timed(lambda: [step() for _ in range(args.steps)])
        replicas = {"value": round(args.steps * L * world / dt_rep, 3), "unit": "frames/s", "ms_per_step": round(dt_rep / args.steps * 1e3, 3),
                    "note": "every rank on its own HBM-resident chunk, nothing exchanged (upper bound of the scatter / gather path)"}

        # (2) the scatter / gather path, under a watchdog: a stuck or failing exchange costs this leg, not the line -- rank 0 then prints
        # the line with `value` = the replicas' rate (units sharded across ranks, no data-path exchange) and says so in `scatter_gather`
        def fallback_line(reason):
            fb = {"metric": "inpainted frames/sec @1080p (STTN, 5-frame window)" if args.res == "1080p"
                  else f"inpainted frames/sec @{args.res} (STTN, 5-frame window)",
                  "value": replicas["value"], "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                  "ms_per_step": replicas["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": base_precision if base_precision != "f16" else "f16 (fp32 accumulate)", "data": "synthetic",
                  "config": {"workload": f"{args.res} synthetic clip, --inpaint-mode sttn-auto, {L}-frame chunks resident in HBM, neighbor stride 5 / "
                                         "refs every 10 (BASELINE.json metric)", "frame_size": [W, H], "strip": [W, int(W * 3 / 16)], "chunk_frames": L,
                             "parallelism": f"chunk-parallel x{world}: every rank on its own resident chunks, NO exchange (the scatter / gather leg failed)"},
                  "replicas": replicas, "scatter_gather": {"error": reason[:400]}}
            print(json.dumps(fb), flush=True)

        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        from bench_multi import Watchdog

        def on_timeout(leg, phase):
            if rank == 0:
                fallback_line(f"watchdog: phase '{phase}' of the scatter / gather leg did not finish within {budget:.0f} s")

        budget = float(os.environ.get("VSR_BENCH_HEADLINE_TIMEOUT", "240"))
        dog = Watchdog(budget, on_timeout)
        try:
            dog.phase("headline", "warm-up rounds")
            if os.environ.get("VSR_BENCH_SCATTER_FAIL") == "1":              # test hook (tests/test_bench_multi.py)
                raise RuntimeError("VSR_BENCH_SCATTER_FAIL=1")
            run_rounds(max(1, args.warmup))
            dog.phase("headline", "timed rounds")
            elapsed = timed(lambda: run_rounds(args.steps))
            dog.cancel()
        except Exception as e:                   # noqa: BLE001 -- the ranks may have left the exchange at different points: no further collective
            dog.cancel()
            if rank == 0:
                fallback_line(f"{type(e).__name__}: {e}")
            sys.stdout.flush()
            os._exit(0)
        if rank == 0:
            checksum = int(sum(int(d[::7, ::5, ::11].sum().item()) for d in dsts))
            replicas["scatter_gather_result_checksum"] = checksum
        if not args.no_selftest:
            # the chunk rank k inpainted from rows that travelled over RCCL and back must equal what rank k computes from its own
            # resident copy of the same clip (same seed): every rank hashes its replica's strip rows, rank 0 hashes what it gathered
            def digest(t):
                v = t.to(torch.int64)
                w = torch.arange(1, v.numel() + 1, dtype=torch.int64, device=t.device).reshape(v.shape) % 1000003
                return [int(v.sum().item()), int((v * w).sum().item() % (1 << 61))]

            own = src.clone()
            eng.auto_chunk(own, dmask, areas)
            torch.cuda.synchronize()
            mine = torch.tensor(digest(own[:, y_lo:y_hi]), dtype=torch.int64, device="cpu" if dry else device)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(allr, mine)
            if rank == 0:
                gathered = [digest(d) for d in dsts]
                per_rank = [{"rank": k, "replica": [int(x) for x in allr[k].tolist()], "gathered": gathered[k]} for k in range(world)]
                ok = all(p["replica"] == p["gathered"] for p in per_rank)
                replicas["selftest"] = {"ok": ok, "ranks": world, "backend": dist.get_backend(),
                                        "what": "strip rows of chunk k after scatter -> inpaint on rank k -> gather, against rank k's own "
                                                "resident run of the same clip (sum and position-weighted sum of all bytes)",
                                        "mismatching_ranks": [p["rank"] for p in per_rank if p["replica"] != p["gathered"]]}
                if not ok:
                    replicas["selftest"]["per_rank"] = per_rank

    total_frames = args.steps * L * world
    fps = total_frames / elapsed
    flops_chunk = eng.chunk_flops(L, dmask, areas)      # what the HIP path contracts: the last block of a window on its neighbour frames
                                                        # only, the decoder on the rows the mask's strip rows are resized from only
    flops_per_frame = flops_chunk / L
    flops_chunk_ref = eng.flops(L, reference=True)      # what the reference's modules (and the CPU oracle) compute: SURVEY 8(d)

    out = {
        "metric": "inpainted frames/sec @1080p (STTN, 5-frame window)" if args.res == "1080p"
        else f"inpainted frames/sec @{args.res} (STTN, 5-frame window)",
        "value": round(fps, 3), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": {"f32": "f32", "split": "f32 (operands as fp16 hi/lo pairs)", "split-format": "f32 (operands as fp16 hi/lo pairs)",
                  "f16": "f16 (fp32 accumulate)"}[base_precision],
        "data": "synthetic",
        "config": {"workload": f"{args.res} synthetic clip, --inpaint-mode sttn-auto, {L}-frame chunks resident in HBM, "
                               f"neighbor stride 5 / refs every 10 (BASELINE.json metric; model cost is resolution-independent)",
                   "frame_size": [W, H], "strip": [W, int(W * 3 / 16)], "chunk_frames": L,
                   "parallelism": f"chunk-parallel x{world}" + ("" if world == 1 else
                                  " (all chunks resident on rank 0, strip rows scattered / gathered point-to-point over RCCL, pipelined)"),
                   "weights": "synthetic variance-preserving (no checkpoint in the reference mount)"},
        "model_tflops": round(flops_per_frame * fps / 1e12, 3),
        "gflop_per_frame": round(flops_per_frame / 1e9, 2),
        "gflop_per_frame_reference": round(flops_chunk_ref / L / 1e9, 2),      # incl. what the reference computes and nothing reads: the last-block
                                                                               # rows of the reference frames, the decoder rows outside the mask
        "model_frac_of_peak": round(flops_per_frame * fps / world / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),   # whole step, per GPU, vs the fp32-MFMA spec
    }

    # ---- roofline leg: K steps of the same resident chunk on ONE lane, HIP events on the launch stream around every launch of the
    # dominant kernel symbol (one record pair per launch, ~410 of the ~1300 launches of a chunk, read back once per chunk).  With
    # two lanes a launch's event-to-event time contains the other lane's work, and rocprofv3's per-kernel durations do too
    # (profiles/: the kernel-trace summary that agrees with this leg is the one of `bench.py --lanes 1`).
    eng.set_lanes(1)
    step()
    torch.cuda.synchronize()
    eng.timing(2)
    eng.timing_reset()
    torch.cuda.synchronize()
    t_leg = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t_leg = time.perf_counter() - t_leg
    eng.timing(False)
    out["lanes"] = args.lanes
    out["single_lane"] = {"value": round(args.steps * L / t_leg, 3), "unit": "frames/s (this rank, resident chunk)",
                          "ms_per_step": round(t_leg / args.steps * 1e3, 3),
                          "note": "the same step with every op on one stream (vsr_sttn_set_lanes(1)), dominant-kernel launches bracketed by "
                                  "HIP events: the pass `roofline` is measured on"}
    if replicas is not None:
        out["replicas"] = replicas
    if world > 1:          # the CPU baseline and the informational legs belong to the N = 1 line only
        args.no_cpu_baseline, args.no_split_half, args.e2e_chunks, args.no_full_work, args.no_configs = True, True, 0, True, True
    if rank == 0:
        # ---- roofline of the dominant kernel from the HIP events of the timed region
        # dominant kernel symbol = the gather-GEMM instantiation with the largest total time; every
        # launch of that symbol is counted (same population as rocprofv3 --stats's per-kernel row)
        dims = {0: (128, 128, 2, 2), 1: (256, 32, 4, 1), 2: (256, 64, 4, 1), 3: (128, 64, 2, 2)}

        def kernels_timed():
            per_kernel = {}
            for cfg, (bm, bn, wm, wn) in dims.items():
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
                per_kernel["gather_gemm_f16_v7<%d>" % (0 if base_precision == "f16" else 1)] = (a, b, c)
            return per_kernel

        per_kernel = kernels_timed()
        dom = max(per_kernel, key=lambda k: per_kernel[k][0])
        ms, n, fl = per_kernel[dom]
        traffic, unit_traffic = None, None
        # PMC passes are separate rocprofv3 runs: scripts/pmc_configs.py (leg "1" = this chunk; profiles/config_traffic.json), before
        # round 6 scripts/summarize_profile.py (profiles/dominant_kernel_pmc.json)
        try:
            leg1 = json.load(open(os.path.join(ROOT, "profiles", "config_traffic.json")))["legs"]["1"]
            traffic = leg1["kernels"][dom]["hbm_bytes_per_launch"]
            unit_traffic = leg1["hbm_bytes_per_unit"]
        except Exception:
            prof = os.path.join(ROOT, "profiles", "dominant_kernel_pmc.json")
            if os.path.exists(prof):
                try:
                    pj = json.load(open(prof))
                    traffic = pj.get("hbm_bytes_per_launch") if pj.get("kernel", "").endswith(dom) else None
                except Exception:
                    traffic = None
        ach = fl / ms / 1e9 if ms > 0 else 0.0
        # --precision split / split-format / f16 (never the default line): the products run on the f16 matrix cores; FLOPs stay the
        # algorithmic 2 per product (three MFMAs each in the split modes), the peak is the dense f16 one
        peak = PEAK_FP32_MFMA_TFLOPS if base_precision == "f32" else PEAK_F16_MFMA_TFLOPS
        out["roofline"] = {"bound": "mfma", "kernel": dom + " (3x3 / 1x1 convs as implicit GEMM: bias + LeakyReLU + residual fused)",
                           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic,
                           "launches": int(n), "avg_launch_ms": round(ms / n, 4) if n else None,
                           "flops_per_launch": round(fl / n) if n else None,
                           "measured_on": f"{args.steps} single-lane steps after the timed region (`single_lane`): with {args.lanes} lanes the "
                                          "launches of the two streams overlap, so a launch's own duration is taken with one"}
        if unit_traffic and args.res == "1080p" and base_precision == "f32":
            # the whole chunk's HBM bytes (every kernel, PMC) at the measured chunk rate
            out["hbm_gb_per_chunk"] = round(unit_traffic / 1e9, 3)
            out["hbm_gbps"] = round(unit_traffic * fps / L / world / 1e9, 1)
        if world == 1:
            # per-op and per-kernel breakdown: events around every launch of two extra chunks, outside the timed region
            eng.timing_reset()
            eng.timing(1)
            extra = 2
            for _ in range(extra):
                step()
            torch.cuda.synchronize()
            eng.timing(False)
            breakdown = {}
            for tag in ("enc", "attn.qkv", "attn.qk", "attn.softmax", "attn.pv", "attn.pv.reduce", "attn.out", "ffn", "dec"):
                a, b, c = eng.timing_get(tag)
                breakdown[tag] = {"ms": round(a, 3), "launches": b, "tflops": round(c / a / 1e9, 2) if a > 0 and c > 0 else None}
            out["op_breakdown"] = breakdown
            out["kernel_breakdown"] = {k: {"ms": round(v[0], 3), "launches": v[1], "avg_launch_ms": round(v[0] / v[1], 4),
                                           "tflops": round(v[2] / v[0] / 1e9, 2)} for k, v in kernels_timed().items()}
            kb_ms = sum(v[0] for v in kernels_timed().values())
            kb_fl = sum(v[2] for v in kernels_timed().values())
            if kb_ms > 0:
                out["roofline"]["every_gemm_launch"] = {
                    "achieved": round(kb_fl / kb_ms / 1e9, 2), "frac": round(kb_fl / kb_ms / 1e9 / peak, 4),
                    "note": "all gather-GEMM launches of a chunk, every kernel symbol (kernel_breakdown), single lane, events around every launch"}
            out["breakdown_note"] = (f"HIP events around every launch of {extra} extra chunks after the timed region (they cost 2.5 % of a "
                                     f"chunk, so the timed region brackets the dominant kernel's launches only)")

        eng.set_lanes(args.lanes)
        refa = None
        if not args.no_cpu_baseline:
            # one full chunk of the timed clip through the oracle on the host cores, and the same chunk through the HIP path
            host_clip1 = src.cpu().numpy()
            ref, dt, dt_net, threads, tried = cpu_baseline(sd, host_clip1, mask01, areas)
            got = src.clone()
            eng.auto_chunk(got, dmask, areas)
            torch.cuda.synchronize()
            got = got.cpu().numpy()
            m = mask01[:, :, 0].astype(bool)
            mse = float(np.mean((got[:, m].astype(np.float64) - ref[:, m].astype(np.float64)) ** 2))
            psnr = float("inf") if mse == 0 else 20 * np.log10(255.0 / np.sqrt(mse))
            out["cpu_baseline"] = {
                "value": round(L / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
                "kind_note": "torch-CPU restatement of the reference modules (identical to them: tests/golden/sttn_auto_net.npz); /root/reference "
                             "is not on the GPU box and its wrappers need cv2, so the reference itself cannot be timed here",
                "model_only": {"value": round(L / dt_net, 4), "unit": "frames/s", "tflops": round(flops_chunk_ref / dt_net / 1e12, 3)},
                "host": {"cpu_count": os.cpu_count(), "limits": effective_cpus()[1], "cpu_model": cpu_model_name(), "torch_threads": threads,
                         "probe_seconds_by_threads": tried},
                "sample": f"oracle chunk body (torch-CPU fp32 restatement of the reference modules + restated cv2 resize / blend) on ONE full "
                          f"{L}-frame {args.res} chunk of the timed clip, end to end in {dt:.1f} s of which STTNInpaint.inpaint "
                          f"(the network, {flops_chunk_ref / 1e12:.2f} TFLOP) {dt_net:.1f} s; thread count picked by a 3-frame probe of the same oracle"}
            if not args.no_cpu_parallel:
                Lp = min(L, 10)       # a bounded sample: 10-frame chunks (2 windows each); 16 x 20 frames at once did not finish in 150 s on the box
                par = cpu_baseline_parallel(Lp, threads, budget_s=150.0, flops_sample=eng.flops(Lp, reference=True), flops_per_frame=flops_chunk_ref / L)
                if par is not None:
                    out["cpu_baseline"]["parallel"] = par
            out["psnr_db_vs_oracle"] = round(psnr, 2) if np.isfinite(psnr) else "inf"
            out["psnr_note"] = (f"masked strip pixels of the full {L}-frame chunk, HIP path vs CPU oracle; max |d| "
                                f"{int(np.abs(got[:, m].astype(np.int16) - ref[:, m].astype(np.int16)).max())}, "
                                f"pixels outside the mask bit-identical: {bool(np.array_equal(got[:, ~m], host_clip1[:, ~m]))}")
            refa = ref[:, m].astype(np.float64)
        if args.e2e_chunks > 0:
            # PCIe-inclusive rate of the plugin's host loop (frames start and end in host memory: pinned staging,
            # H2D / compute / D2H pipelined over three streams).  Reported beside `value`, never as `value`.
            from vsr_amd.backend.inpaint.sttn_auto_inpaint import STTNAutoInpaint
            from vsr_amd.backend.tools.video_io import ArrayVideo, CountingWriter

            host_clip = np.concatenate([src.cpu().numpy()] * args.e2e_chunks)[: args.e2e_chunks * L]
            sink = CountingWriter()

            class _Host:
                ab_sections = None
                gui_mode = False
                video_writer = sink

                def update_progress(self, tbar, increment):
                    pass

            plug = STTNAutoInpaint(f"cuda:{local_rank}", {"netG": sd}, ArrayVideo(host_clip), clip_gap=L)
            plug(input_mask=mask, input_sub_remover=_Host(), tbar=None)          # warm-up pass (plans, pinned buffers)
            sink.count = 0
            t2 = time.perf_counter()
            plug(input_mask=mask, input_sub_remover=_Host(), tbar=None)
            dt2 = time.perf_counter() - t2
            out["pcie_inclusive"] = {"value": round(sink.count / dt2, 3), "unit": "frames/s", "frames": sink.count,
                                     "note": "STTNAutoInpaint.__call__ over an in-memory 1080p clip: host frames -> pinned -> HBM -> "
                                             "inpaint -> pinned -> writer.write(frame); decode/encode excluded"}
            plug.sttn_inpaint.engine.close()
        if not args.no_split_half:
            # informational: the same workload with split-half operands on the f16 matrix cores (fp32 data,
            # fp32 accumulation, 22-bit operands, device-side range guard with fp32 fallback).  NOT `value`.
            for mode, key in (("split", "split_half_mode"), ("split-format", "split_format_mode"), ("f16", "fp16_mode")):
                eng.set_precision(mode)
                for _ in range(max(1, args.warmup)):
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t1
                sp = {"value": round(args.steps * L / dt, 3), "unit": "frames/s (this rank)", "ms_per_step": round(dt / args.steps * 1e3, 3),
                      "fp32_fallback_chunks": eng.fallbacks(),
                      "arithmetic": ("fp16 operands (hi halves of the split-format tensors), fp32 accumulate, 1x v_mfma_f32_32x32x16_f16 "
                                     "per product; bias / activation / residual / softmax in fp32") if mode == "f16" else
                                    ("fp32 data + fp32 accumulate; operands as fp16 hi/lo pairs, a*b = a_lo*b_hi + a_hi*b_lo + a_hi*b_hi "
                                     "(3x v_mfma_f32_32x32x16_f16)" + ("; tensors kept in split format by their producers, "
                                     "operands by LDS-DMA" if mode == "split-format" else "; split inside the GEMM"))}
                if refa is not None:
                    got2 = src.clone()
                    eng.auto_chunk(got2, dmask, areas)
                    torch.cuda.synchronize()
                    mse2 = float(np.mean((got2.cpu().numpy()[:, m].astype(np.float64) - refa) ** 2))
                    sp["psnr_db_vs_oracle"] = "inf" if mse2 == 0 else round(20 * np.log10(255.0 / np.sqrt(mse2)), 2)
                out[key] = sp
            eng.set_precision(base_precision)
        if not args.no_full_work and os.environ.get("VSR_TRIM_LAST_BLOCK", "1") != "0":
            # `value` is measured on a plan that leaves out what the reference computes and nothing reads (the last block's
            # reference-frame rows, the decoder rows outside the mask; same frames bit for bit).  The same step with ALL of the
            # reference's work, for comparison: a child process, because the library reads the two switches once per process.
            import subprocess

            env = dict(os.environ, VSR_TRIM_LAST_BLOCK="0", VSR_DECODE_ROWS="0")
            cmd = [sys.executable, os.path.abspath(__file__), "--steps", str(args.steps), "--warmup", str(args.warmup), "--res", args.res,
                   "--chunk", str(L), "--lanes", str(args.lanes), "--no-cpu-baseline", "--no-split-half", "--e2e-chunks", "0", "--no-full-work", "--no-configs"]
            if args.precision:
                cmd += ["--precision", args.precision]
            try:
                r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                fw = json.loads(line)
                out["full_work"] = {"value": fw["value"], "unit": "frames/s", "ms_per_step": fw["ms_per_step"],
                                    "gflop_per_frame": fw["gflop_per_frame"], "model_tflops": fw["model_tflops"],
                                    "note": "VSR_TRIM_LAST_BLOCK=0 VSR_DECODE_ROWS=0 in a child process: every row of the last block and "
                                            "of the decoder, i.e. the reference's FLOP count; the frames are the same bit for bit "
                                            "(tests/test_gpu_sttn.py::test_decoder_rows_give_the_same_frames)"}
            except Exception as e:                 # noqa: BLE001 -- informational leg, never fatal
                out["full_work"] = {"error": repr(e)[:200]}
        if not args.no_configs:
            # BASELINE.json's other configurations, each with its own roofline: outside the timed region, informational beside `value`
            try:
                sys.path.insert(0, os.path.join(ROOT, "scripts"))
                import bench_configs

                out["configs"] = bench_configs.run_all(args.configs.split(",") if args.configs else None)
            except Exception as e:                 # noqa: BLE001 -- never fatal for the headline line
                out["configs"] = {"error": repr(e)[:300]}
    eng.close()
    if world > 1 and not args.no_multi_configs:
        # BASELINE.json's own multi-GPU configurations after the headline: config 5 (4K, fp16 operands, chunk-parallel) and config 4
        # (propainter batches, batch-parallel; reference arithmetic and exact), every rank takes part.  A leg that hangs is cut by
        # a watchdog which prints this line with what there is (scripts/bench_multi.py).
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import bench_multi

        res = bench_multi.run_multi(dist, rank, world, device, dry, out if rank == 0 else {},
                                    legs=args.multi_configs.split(",") if args.multi_configs else None, steps=2, warmup=1)
        if rank == 0:
            out["configs_multi"] = res
    if rank == 0:
        print(json.dumps(out), flush=True)
        if world > 1 and replicas is not None and not replicas.get("selftest", {"ok": True})["ok"]:
            print("SELFTEST FAILED: gathered chunks differ from the ranks' replica results", file=sys.stderr, flush=True)
            selftest_failed = True

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if selftest_failed:
        raise SystemExit(3)


if __name__ == "__main__":
    main()
