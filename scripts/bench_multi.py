#!/usr/bin/env python3
"""The N-rank legs of BASELINE.json's multi-GPU configurations, run by bench.py after its headline when WORLD_SIZE > 1:

  config 5 ("5"):  4K sttn-auto on fp16 operands, one 50-frame chunk per rank and round through backend/tools/chunk_parallel.py
                   (all chunks resident on rank 0, strip rows scattered / gathered point-to-point, pipelined) -- the reference's
                   independent chunks, backend/inpaint/sttn_auto_inpaint.py:242-328.
  config 4 ("4", "4h"):  1080p propainter, one 68-frame batch (what batch_generator(1200, 70) makes) per rank through
                   backend/tools/batch_parallel.py, in exact fp32 and in the reference's GPU arithmetic -- the reference's independent
                   batches, backend/main.py:229-245.

Each leg reports `value` (whole-job frames/s through the exchange), `replicas` (every rank on its own HBM-resident unit, nothing
exchanged: N x the one-GPU rate when the ranks have a GPU each), `efficiency` = value / replicas.value -- what the exchange costs --
and `per_rank_replica_fps`.  (fps(N) / (N fps(1)) proper needs the N = 1 run of the same leg: the driver computes it from its own
per-N lines; `configs` of the N = 1 line carries fps(1).)  `hbm_gbps` comes from the committed PMC summary (profiles/config_traffic.json).

A WATCHDOG guards every phase of a leg (VSR_BENCH_LEG_TIMEOUT seconds, default 120): when a phase does not finish -- a stuck RCCL
group is the case this exists for -- rank 0 prints the bench line with what it has (the headline and the finished legs, the stuck
leg marked) and every rank leaves with os._exit: the lease loses one leg, not the run.
"""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))


class Watchdog:
    """phase(name) re-arms the timer; on expiry `on_timeout(leg, phase)` runs on the timer thread and the process exits"""

    def __init__(self, seconds, on_timeout):
        self.seconds, self.on_timeout = seconds, on_timeout
        self.timer, self.leg, self.name = None, None, None

    def phase(self, leg, name):
        self.cancel()
        self.leg, self.name = leg, name
        self.timer = threading.Timer(self.seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def cancel(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self):
        try:
            self.on_timeout(self.leg, self.name)
        finally:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)


def _timed(dist, dry, device, fn):
    """bench.py's bracket: barrier + synchronize on both sides, max over ranks"""
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dry else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def leg_sttn_chunks(dist, rank, world, device, dry, wd, leg, res, precision, steps, warmup, L=50):
    from bench_configs import RES, attach_traffic, make_chunk_on_device
    from vsr_amd.backend.tools import chunk_parallel as cp
    from vsr_amd.backend.tools.inpaint_tools import create_mask, get_inpaint_area_by_mask, threshold_mask
    from vsr_amd.engine import SttnEngine
    from vsr_amd.synth import make_state_dict

    wd.phase(leg, "setup")
    H, W, box = RES[res]
    eng = SttnEngine(make_state_dict(0, "auto"), "auto", device=device.index, precision=precision)
    mask01 = threshold_mask(create_mask((H, W), [(box[2], box[3], box[0], box[1])]))
    areas = get_inpaint_area_by_mask(W, H, int(W * 3 / 16), mask01)
    dmask = torch.from_numpy(np.ascontiguousarray(mask01[:, :, 0])).to(device)
    y_lo, y_hi = min(a[0] for a in areas), max(a[1] for a in areas)
    local_areas = [(a[0] - y_lo, a[1] - y_lo, a[2], a[3]) for a in areas]
    dmask_rows = dmask[y_lo:y_hi].contiguous()
    src = make_chunk_on_device(L, H, W, box, seed=11 + rank, device=device)
    work = src.clone()
    if rank == 0:
        # (the strips only: N whole 4K chunks would be N x 1.2 GB for rows that never travel)
        srcs = [src[:, y_lo:y_hi].contiguous()] + [make_chunk_on_device(L, H, W, box, seed=11 + k, device=device)[:, y_lo:y_hi].contiguous()
                                                  for k in range(1, world)]
        dsts = [torch.empty((L, y_hi - y_lo, W, 3), dtype=torch.uint8, device=device) for _ in range(world)]

    def run_rounds(n):
        ranges = [(i * L, (i + 1) * L) for i in range(n * world)]
        cp.run_chunk_parallel(ranges, (y_hi - y_lo, W, 3), lambda i, out: out.copy_(srcs[i % world]),
                              lambda i, rows: eng.auto_chunk(rows, dmask_rows, local_areas), lambda i, rows: dsts[i % world].copy_(rows),
                              dist=dist, device=device, io="device")

    def step():
        work.copy_(src)
        eng.auto_chunk(work, dmask, areas)

    wd.phase(leg, "warmup")
    run_rounds(max(1, warmup))
    wd.phase(leg, "timed exchange")
    dt = _timed(dist, dry, device, lambda: run_rounds(steps))
    wd.phase(leg, "replicas")
    step()
    dt_rep = _timed(dist, dry, device, lambda: [step() for _ in range(steps)])
    wd.phase(leg, "selftest")
    # what rank k inpainted from rows that travelled must equal what it computes from its own resident copy of the same clip
    own = src.clone()
    eng.auto_chunk(own, dmask, areas)
    torch.cuda.synchronize()
    mine = torch.tensor([int(own[:, y_lo:y_hi].to(torch.int64).sum().item())], dtype=torch.int64, device="cpu" if dry else device)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    out = None
    if rank == 0:
        ok = all(int(allr[k].item()) == int(dsts[k].to(torch.int64).sum().item()) for k in range(world))
        fps, rep = steps * L * world / dt, steps * L * world / dt_rep
        out = {"config": f"{leg}: {res} sttn-auto, {precision} operands, {L}-frame chunks, chunk-parallel x{world}", "n_ranks": world,
               "value": round(fps, 2), "unit": "frames/s", "ms_per_round": round(dt / steps * 1e3, 2),
               "replicas": {"value": round(rep, 2), "unit": "frames/s", "note": "every rank on its own resident chunk, nothing exchanged"},
               "per_rank_replica_fps": round(rep / world, 2), "efficiency": round(fps / rep, 4),
               "efficiency_note": "value / replicas.value: the cost of the scatter / gather path at this N (fps(N) / (N fps(1)) needs the N = 1 line)",
               "selftest_ok": ok, "backend": dist.get_backend(), "fp32_fallback_chunks": eng.fallbacks()}
        attach_traffic(leg, out, fps / L)
    eng.close()
    return out


def leg_propainter_batches(dist, rank, world, device, dry, wd, leg, precision, steps, L=None):
    from bench_configs import attach_traffic
    from vsr_amd.backend.inpaint.propainter_inpaint import PropainterInpaint
    from vsr_amd.backend.tools import batch_parallel as bp
    from vsr_amd.synth import make_clip, make_propainter_state_dict, make_raft_state_dict, make_rfc_state_dict

    wd.phase(leg, "setup")
    L = L or int(os.environ.get("VSR_BENCH_MULTI_PP_FRAMES", "68"))      # (a dry run of two ranks on ONE GPU has no room for two 68-frame workspaces)
    H, W = 360, 1920
    box = (H // 2, H - H // 6, W // 6, W - W // 6)
    base = make_clip(10, H, W, box, seed=4 + rank)
    host = np.concatenate([np.roll(base, (2 * k, 3 * k), axis=(1, 2)) for k in range((L + 9) // 10)], 0)[:L]
    frames_dev = torch.from_numpy(np.ascontiguousarray(host)).to(device)
    mask = np.zeros((H, W), np.uint8)
    mask[box[0]:box[1], box[2]:box[3]] = 255
    plug = PropainterInpaint(f"cuda:{device.index}", {"raft": make_raft_state_dict(0), "rfc": make_rfc_state_dict(0),
                                                       "propainter": make_propainter_state_dict(0)}, precision=precision)
    p2p_dev = "cpu" if dry else device          # gloo (dry run) moves host memory only
    written = [0]

    def process(frames, m):
        return plug.inpaint(frames, m)

    def items(n_rounds):
        for _ in range(n_rounds * world):
            yield (bp.WORK, list(host), mask)

    def run(n_rounds):
        written[0] = 0
        bp.run_batch_parallel(items(n_rounds) if rank == 0 else None, process, lambda f: written.__setitem__(0, written[0] + 1), dist=dist,
                              device=p2p_dev, prefetch_frames=0)

    wd.phase(leg, "warmup (plans, workspaces)")
    plug.inpaint(frames_dev, mask)
    torch.cuda.synchronize()
    wd.phase(leg, "warmup exchange")
    run(1)
    wd.phase(leg, "timed exchange")
    dt = _timed(dist, dry, device, lambda: run(steps))
    nwritten = written[0]
    wd.phase(leg, "replicas")
    dt_rep = _timed(dist, dry, device, lambda: [plug.inpaint(frames_dev, mask) for _ in range(steps)])
    out = None
    if rank == 0:
        fps, rep = steps * L * world / dt, steps * L * world / dt_rep
        out = {"config": f"{leg}: 1080p propainter ({precision}), one {L}-frame batch per rank and round, batch-parallel x{world}", "n_ranks": world,
               "value": round(fps, 2), "unit": "frames/s", "s_per_round": round(dt / steps, 3), "frames_written": nwritten,
               "replicas": {"value": round(rep, 2), "unit": "frames/s", "note": "every rank on its own HBM-resident batch, nothing exchanged"},
               "per_rank_replica_fps": round(rep / world, 2), "efficiency": round(fps / rep, 4),
               "efficiency_note": "value / replicas.value; the exchange of batch_parallel goes through host frames on rank 0 (the reader's and the "
                                  "sink's side of backend/main.py:229-245), the replicas start from HBM",
               "backend": dist.get_backend(), "range_guard_fallbacks": [e.fallbacks() for e in (plug.fix_raft, plug.fix_flow_complete, plug.model)]}
        if L == 68:                              # the PMC summary's unit is a 68-frame batch (a dry run with a shorter batch has no traffic figure)
            attach_traffic(leg, out, fps / L)
    plug.close()
    return out


def run_multi(dist, rank, world, device, dry, headline, legs=None, steps=2, warmup=1):
    """returns {leg: result} on rank 0 (None elsewhere).  `headline`: rank 0's bench line so far -- printed by the watchdog if a leg hangs"""
    results = {}
    budget = float(os.environ.get("VSR_BENCH_LEG_TIMEOUT", "120"))

    def on_timeout(leg, phase):
        if rank == 0:
            results[leg] = {"error": f"watchdog: phase '{phase}' of leg {leg} did not finish within {budget:.0f} s; the remaining legs were not run"}
            line = dict(headline)
            line["configs_multi"] = results
            print(json.dumps(line), flush=True)
        print(f"[bench_multi] rank {rank}: leg {leg} phase '{phase}' timed out after {budget:.0f} s; leaving", file=sys.stderr, flush=True)

    wd = Watchdog(budget, on_timeout)
    for leg in (legs or ["5", "4h", "4"]):
        t0 = time.perf_counter()
        try:
            if leg == "5":
                r = leg_sttn_chunks(dist, rank, world, device, dry, wd, "5", "4k", "f16", steps, warmup)
            elif leg == "2":
                r = leg_sttn_chunks(dist, rank, world, device, dry, wd, "2", "720p", "f32", steps, warmup)
            elif leg in ("4", "4h"):
                r = leg_propainter_batches(dist, rank, world, device, dry, wd, leg, "f32" if leg == "4" else "f16", max(1, steps // 2))
            else:
                raise ValueError(f"unknown multi-GPU leg {leg!r}")
        except Exception as e:          # noqa: BLE001 -- a failing leg is reported, the line still goes out
            r = {"error": repr(e)[:300]}
            # the ranks may have left the leg at different points: re-align (the watchdog bounds this barrier as well)
            wd.phase(leg, "barrier after a failed leg")
            try:
                dist.barrier()
            except Exception:           # noqa: BLE001
                pass
        wd.cancel()
        if rank == 0:
            r = r or {}
            r["leg_seconds"] = round(time.perf_counter() - t0, 1)
            results[leg] = r
        torch.cuda.empty_cache()
    return results if rank == 0 else None
